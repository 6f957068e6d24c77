# Reference-side binding for libnnsdp_hip.so (include/nnsdp.h).  Drop this file into
# src/Methods/ of AntonXue/nn-sdp and add `include("admm_sdp.jl"); export AdmmSdpOptions` after
# the chordal_sdp.jl include in src/Methods/Methods.jl:141-143.  NOT executed in this repository
# (no Julia in the build image); the ctypes mirror in nn-sdp_amd/nnsdp_amd binds the same symbols
# and is what the tests exercise.
#
# Replaces, for opts::AdmmSdpOptions only, the generic runQuery (src/Methods/Methods.jl:91-131):
# no JuMP model is built, no MOSEK is called.

const LIBNNSDP = get(ENV, "NNSDP_LIB", "libnnsdp_hip.so")

# extension: cliques {x_k, x_{k+1}, affine index}; exact whenever the output QC has no x_1 -- x_K coupling (S12 = 0)
struct PathDecomp <: DecompMode end
struct AutoDecomp <: DecompMode end      # PathDecomp when the query allows it, DoubleDecomp otherwise (nnsdp.h: NNSDP_DECOMP_AUTO)

@with_kw struct AdmmSdpOptions <: QueryOptions
  decomp_mode::DecompMode = SingleDecomp()   # SingleDecomp / DoubleDecomp / DoubleRelaxDecomp (chordal_sdp.jl:4-8) / PathDecomp
  dense::Bool = false                        # true: one dense cone (DeepSdpOptions behaviour)
  max_iters::Int = 20000
  eps_rel::Float64 = 1e-6
  max_time::Float64 = 0.0
  sigma::Float64 = 0.1
  alpha::Float64 = 1.6
  adapt_every::Int = 50
  check_every::Int = 50
  normalize::Bool = true
  warm_start::Bool = true
  proj_tol::Float64 = 0.0
  polish::Bool = true
  cert_tol::Float64 = 0.0
  verbose::Bool = false
  device::Int = -1
  interval_guard::Float64 = 5e-5
  minv_mode::Int = 0
  proj_refine::Int = 1
end

# field order and types must match include/nnsdp.h
struct CProblem
  K::Int32; xdims::Ptr{Int32}; M::Ptr{Float64}
  x1min::Ptr{Float64}; x1max::Ptr{Float64}
  acymin::Ptr{Float64}; acymax::Ptr{Float64}; smin::Ptr{Float64}; smax::Ptr{Float64}
  beta::Int32; query_kind::Int32; out_kind::Int32
  normal::Ptr{Float64}; yc::Ptr{Float64}; invP::Ptr{Float64}; S::Ptr{Float64}
  activ::Int32
end
struct COptions
  decomp_mode::Int32; max_iters::Int32; eps_rel::Float64; max_time::Float64; sigma::Float64; alpha::Float64
  adapt_every::Int32; check_every::Int32; normalize::Int32; warm_start::Int32; proj_tol::Float64
  polish::Int32; cert_tol::Float64; verbose::Int32; device::Int32; interval_guard::Float64; minv_mode::Int32; proj_refine::Int32
end
mutable struct CResult
  gamma_in::Ptr{Float64}; gamma_out::Ptr{Float64}; gamma_ac1::Ptr{Float64}; gamma_ac2::Ptr{Float64}; Z::Ptr{Float64}
  objective::Float64; status::Int32; iters::Int32; pres::Float64; dres::Float64; lambda_max::Float64
  t_setup::Float64; t_solve::Float64; t_total::Float64; t_eig::Float64
  n_cliques::Int32; max_clique::Int32; eig_flops_per_iter::Int64; eig_bytes_per_iter::Int64; avg_sweeps::Float64
  objective_admm::Float64; polish_shift::Float64
  refine_blocks::NTuple{5,Int64}
end

function runQuery(query::Query, opts::AdmmSdpOptions)
  ffnet = query.ffnet
  qb = only(filter(q -> q isa QcActivBounded, query.qc_activs))
  qs = only(filter(q -> q isa QcActivSector, query.qc_activs))
  xdims = Int32.(ffnet.xdims)
  M = vcat([vec(Matrix{Float64}(Mk)) for Mk in ffnet.Ms]...)           # column-major, back to back
  x1min, x1max = Float64.(query.qc_input.x1min), Float64.(query.qc_input.x1max)
  acymin, acymax = Float64.(qb.acymin), Float64.(qb.acymax)
  smin, smax = Float64.(Vector(qs.smin)), Float64.(Vector(qs.smax))
  normal = Float64[]; yc = Float64[]; invP = Float64[]; S = Float64[]
  if query isa ReachQuery
    qkind = Int32(1); qo = query.qc_reach
    if qo isa QcReachHplane;        okind = Int32(1); normal = Float64.(qo.normal)
    elseif qo isa QcReachCircle;    okind = Int32(2); yc = Float64.(qo.yc)
    elseif qo isa QcReachEllipsoid; okind = Int32(3); yc = Float64.(qo.yc); invP = vec(Matrix{Float64}(qo.invP))
    else error("unrecognized qc: $(qo)") end
  elseif query isa SafetyQuery
    qkind = Int32(0); okind = Int32(0); S = vec(Matrix{Float64}(query.qc_safety.S))
  else
    error("unrecognized query: $(query)")
  end
  Zdim = sum(ffnet.zdims)
  gin = zeros(query.qc_input.vardim); gout = zeros(1); gac1 = zeros(qb.vardim); gac2 = zeros(qs.vardim)
  Z = zeros(Zdim, Zdim)
  # DoubleRelaxDecomp is treated exactly like DoubleDecomp by the reference (chordal_sdp.jl:25)
  mode = opts.dense ? Int32(0) : opts.decomp_mode isa SingleDecomp ? Int32(1) : opts.decomp_mode isa PathDecomp ? Int32(3) : opts.decomp_mode isa AutoDecomp ? Int32(4) : Int32(2)
  # obj_func (Methods.jl:41) is affine in γout[1] at every call site (x -> x[1], NnSdp.jl:46,66,87); any increasing affine
  # a*ρ + b has the same minimiser, so it is evaluated on the solution instead of being handed to the solver
  if query isa ReachQuery
    f0, f1 = query.obj_func([0.0]), query.obj_func([1.0])
    f1 > f0 || error("obj_func must be increasing in γout[1]")
  end
  copts = COptions(mode, opts.max_iters, opts.eps_rel, opts.max_time, opts.sigma, opts.alpha, opts.adapt_every,
                   opts.check_every, opts.normalize, opts.warm_start, opts.proj_tol, opts.polish, opts.cert_tol, opts.verbose, opts.device, opts.interval_guard, opts.minv_mode, opts.proj_refine)
  res = CResult(pointer(gin), pointer(gout), pointer(gac1), pointer(gac2), pointer(Z),
                0.0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0.0, 0.0, 0.0, (0, 0, 0, 0, 0))
  GC.@preserve xdims M x1min x1max acymin acymax smin smax normal yc invP S gin gout gac1 gac2 Z begin
    p(v) = isempty(v) ? Ptr{Float64}(C_NULL) : pointer(v)
    prob = CProblem(ffnet.K, pointer(xdims), pointer(M), pointer(x1min), pointer(x1max), pointer(acymin), pointer(acymax),
                    pointer(smin), pointer(smax), Int32(qs.β), qkind, okind, p(normal), p(yc), p(invP), p(S),
                    ffnet.activ isa TanhActiv ? Int32(1) : Int32(0))
    rc = ccall((:nnsdp_solve, LIBNNSDP), Cint, (Ref{CProblem}, Ref{COptions}, Ref{CResult}), prob, copts, res)
    rc == 0 || error("nnsdp_solve failed ($rc): " * unsafe_string(ccall((:nnsdp_last_error, LIBNNSDP), Cstring, ())))
  end
  values = Dict{Symbol,Any}(:γin => gin, :γac1 => gac1, :γac2 => gac2, :Z => Z)
  if query isa ReachQuery; values[:γout] = gout end
  status = unsafe_string(ccall((:nnsdp_status_string, LIBNNSDP), Cstring, (Int32,), res.status))
  if opts.verbose
    println("setup: $(round(res.t_setup, digits=3)) \tsolve: $(round(res.t_solve, digits=3)) \ttotal: $(round(res.t_total, digits=3)) \t" *
            "obj: $(round(res.objective, digits=5)) ($(status)) \tλmax: $(round(res.lambda_max, digits=7))")
  end
  objval = query isa ReachQuery ? query.obj_func(gout) : res.objective
  return QuerySolution(model = nothing, objective_value = objval, values = values, summary = res,
                       termination_status = status, total_time = res.t_total, setup_time = res.t_setup, solve_time = res.t_solve)
end

# Utils.sampleTrajs (src/Utils/qc.jl:40-47) with the N forward passes on the GPU (nnsdp_eval_network, fp64): same return value,
# a Vector of output vectors; Utils.approxEllipsoid (qc.jl:50-67) works on it unchanged.
function sampleTrajsGpu(ffnet::FeedFwdNet, x1min::VecReal, x1max::VecReal, N=Int(1e5))
  @assert length(x1min) == length(x1max) == ffnet.xdims[1]
  xdims = Int32.(ffnet.xdims)
  M = vcat([vec(Matrix{Float64}(Mk)) for Mk in ffnet.Ms]...)
  X = x1min .+ rand(ffnet.xdims[1], N) .* (x1max - x1min)            # xdims[1] x N, column-major as the library expects
  Y = zeros(ffnet.xdims[end], N)
  activ = ffnet.activ isa TanhActiv ? Int32(1) : Int32(0)
  GC.@preserve xdims M X Y begin
    rc = ccall((:nnsdp_eval_network, LIBNNSDP), Cint,
               (Int32, Ptr{Int32}, Ptr{Float64}, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
               Int32(ffnet.K), pointer(xdims), pointer(M), activ, Int64(N), pointer(X), pointer(Y), C_NULL)
    rc == 0 || error("nnsdp_eval_network failed ($rc): " * unsafe_string(ccall((:nnsdp_last_error, LIBNNSDP), Cstring, ())))
  end
  return [Y[:, j] for j in 1:N]
end

# Interval pre-processing without the PyCall / ONNX / auto_LiRPA bridge (replaces Intervals.intervalsAutoLirpaSliced,
# src/Intervals/intervals_auto_lirpa.jl:12-64, and the sector test of Qc.makeSectorMinMax, src/Qc/activ_sector.jl:63-72).
# Drop-in for Qc.makeQcActivs (src/Qc/activ.jl:45-72): same return value.
function makeQcActivsNative(ffnet::FeedFwdNet; x1min::VecReal, x1max::VecReal, β::Int)
  xdims = Int32.(ffnet.xdims)
  M = vcat([vec(Matrix{Float64}(Mk)) for Mk in ffnet.Ms]...)       # column-major [W_k b_k], back to back
  acdim = sum(ffnet.xdims[2:end-1])
  acymin = zeros(acdim); acymax = zeros(acdim); smin = zeros(acdim); smax = zeros(acdim)
  lo = Vector{Float64}(x1min); hi = Vector{Float64}(x1max)
  GC.@preserve xdims M lo hi acymin acymax smin smax begin
    rc = ccall((:nnsdp_make_intervals, LIBNNSDP), Cint,
               (Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
               Int32(ffnet.K), pointer(xdims), pointer(M), pointer(lo), pointer(hi), pointer(acymin), pointer(acymax),
               C_NULL, C_NULL, pointer(smin), pointer(smax), C_NULL, C_NULL)
    rc == 0 || error("nnsdp_make_intervals failed ($rc): " * unsafe_string(ccall((:nnsdp_last_error, LIBNNSDP), Cstring, ())))
  end
  qc_bounded = QcActivBounded(acydim=acdim, acymin=acymin, acymax=acymax)
  qc_sector = QcActivSector(activ=ffnet.activ, acxdim=acdim, β=β, base_smin=0.0, base_smax=1.0, smin=smin, smax=smax)
  return [qc_bounded, qc_sector]
end
