"""`.nnet` text reader and the FeedFwdNet container (oracle side).

Follows the file layout consumed by the reference's parsers
(exts/nnet_parser.jl:23-131, exts/NNet/utils/readNNet.py:3-78) and the struct
in src/MyNeuralNetwork/MyNeuralNetwork.jl:12-27:
  xdims = layer sizes (length K+1), Ms[k] = [W_k b_k] of size xdims[k+1] x (xdims[k]+1),
  zdims = [xdims[0:K]; 1].
Normalisation lines (mins/maxes/means/ranges) are read and ignored, exactly as
loadFromNnet does (src/MyNeuralNetwork/network_files.jl:17-22).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


@dataclass
class FeedFwdNet:
    xdims: List[int]          # length K+1
    Ms: List[np.ndarray]      # K matrices [W_k b_k], float64

    @property
    def K(self) -> int:
        return len(self.Ms)

    @property
    def zdims(self) -> List[int]:
        return list(self.xdims[:-1]) + [1]

    @property
    def Zdim(self) -> int:
        return int(sum(self.xdims[:-1]) + 1)

    def W(self, k: int) -> np.ndarray:
        return self.Ms[k][:, :-1]

    def b(self, k: int) -> np.ndarray:
        return self.Ms[k][:, -1]

    def check(self) -> None:
        assert len(self.xdims) >= 3
        assert len(self.xdims) == self.K + 1
        for k in range(self.K):
            assert self.Ms[k].shape == (self.xdims[k + 1], self.xdims[k] + 1)


def eval_net(net: FeedFwdNet, x: np.ndarray) -> np.ndarray:
    """ReLU forward pass, batched over columns (MyNeuralNetwork.jl:40-46)."""
    xk = np.asarray(x, dtype=np.float64)
    single = xk.ndim == 1
    if single:
        xk = xk[:, None]
    for k in range(net.K - 1):
        xk = np.maximum(net.W(k) @ xk + net.b(k)[:, None], 0.0)
    xk = net.W(net.K - 1) @ xk + net.b(net.K - 1)[:, None]
    return xk[:, 0] if single else xk


def read_nnet(path: str) -> FeedFwdNet:
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f if not ln.startswith("//")]
    it = iter(lines)
    rec = next(it).split(",")
    num_layers = int(rec[0])
    sizes = [int(v) for v in next(it).split(",")[: num_layers + 1]]
    next(it)                      # obsolete flag line
    for _ in range(4):            # mins, maxes, means, ranges: unused by the path
        next(it)
    Ms = []
    for k in range(num_layers):
        nin, nout = sizes[k], sizes[k + 1]
        W = np.zeros((nout, nin))
        for i in range(nout):
            vals = [v for v in next(it).split(",") if v != ""]
            W[i, :] = [float(v) for v in vals[:nin]]
        b = np.zeros(nout)
        for i in range(nout):
            b[i] = float(next(it).split(",")[0])
        Ms.append(np.hstack([W, b[:, None]]))
    net = FeedFwdNet(xdims=sizes, Ms=Ms)
    net.check()
    return net


def save_npz(net: FeedFwdNet, path: str) -> None:
    """Binary fixture format used under tests/golden/ (weights as float64)."""
    arrs = {"xdims": np.asarray(net.xdims, dtype=np.int32)}
    for k, M in enumerate(net.Ms):
        arrs[f"M{k}"] = M
    np.savez_compressed(path, **arrs)


def load_npz(path: str) -> FeedFwdNet:
    d = np.load(path)
    xdims = [int(v) for v in d["xdims"]]
    Ms = [np.array(d[f"M{k}"], dtype=np.float64) for k in range(len(xdims) - 1)]
    net = FeedFwdNet(xdims=xdims, Ms=Ms)
    net.check()
    return net


def random_net(xdims: List[int], seed: int = 1234) -> FeedFwdNet:
    """Synthetic net with the reference's distribution: every W/b entry i.i.d.
    N(0, sigma^2), sigma = 2/sqrt(W ln W) with W the hidden width
    (scripts/make_networks.jl:22-28,43-46)."""
    rng = np.random.default_rng(seed)
    width = max(xdims[1:-1])
    sigma = 2.0 / np.sqrt(width * np.log(width))
    Ms = [rng.normal(0.0, sigma, size=(xdims[k + 1], xdims[k] + 1)) for k in range(len(xdims) - 1)]
    net = FeedFwdNet(xdims=list(xdims), Ms=Ms)
    net.check()
    return net
