#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the bench workload,
# aggregated into gpurun_out/profiles_new/ (copy into profiles/ afterwards).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 200 --warmup 50 --no-cpu-baseline --cert-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_LDS -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_LDS.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_SQ.err || true
cd $R
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, glob, json, sys, shutil
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for path in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            if "nnsdp" not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(f"{out}/{tag}_pmc_counters_W40-D20.json", "w"), indent=1)
for p in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    shutil.copy(p, f"{out}/{tag}_bench_W40-D20_kernel_stats.csv")
k = [v for kk, v in res.items() if "k_proj_jacobi" in kk]
if k:
    k = k[0]
    f, w = k.get("FETCH_SIZE", {}).get("mean_per_launch"), k.get("WRITE_SIZE", {}).get("mean_per_launch")
    print("k_proj_jacobi per launch: FETCH_SIZE KB", f, "WRITE_SIZE KB", w, "LDS conflict share",
          k.get("SQ_LDS_BANK_CONFLICT", {}).get("mean_per_launch", 0) / max(k.get("SQ_LDS_IDX_ACTIVE", {}).get("mean_per_launch", 1), 1))
PY
rm -rf $OUT/stats $OUT/pmc_*   # raw traces are large; only the aggregates travel back
head -4 $OUT/${TAG}_bench_W40-D20_kernel_stats.csv | cut -c1-160
