#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py workload (run on the GPU box): tools/stats_one.sh <workload> <beta> <tag>
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-W40-D40}; B=${2:-0}; TAG=${3:-r02}
OUT=$R/gpurun_out/stats_${W}_b${B}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $R/bench.py --workload $W --beta $B --steps 200 --warmup 50 --burn-in 1000 --batch 0 --no-cpu-baseline --cert-seconds 0 > $OUT/bench.json 2> $OUT/err.log
cp $OUT/raw/*/*kernel_stats.csv $OUT/${TAG}_${W}_b${B}_kernel_stats.csv
head -14 $OUT/${TAG}_${W}_b${B}_kernel_stats.csv | cut -c1-150
