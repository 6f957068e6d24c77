#!/bin/bash
# (diagnostic) does rocprofv3 --pmc work at all on this box?  a plain HIP binary, then python + the library eagerly, then bench.py eagerly
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
cd /tmp
run() { # name, command...
  n=$1; shift
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$n -- "$@" > $O/$n.out 2> $O/$n.err
  rc=$?
  echo "$n: rc=$rc counter files: $(find $O/$n -name '*counter_collection.csv' | wc -l)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  find $O/$n -name '*kernel_trace.csv' -size +4M -delete
}
export NNSDP_NO_GRAPH=1
run w1 python3 $R/tools/pmc_window.py
run w2 python3 $R/tools/pmc_window.py
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/w3 -- python3 $R/tools/pmc_window.py > $O/w3.out 2> $O/w3.err; echo "w3 rc=$?"
find $O -name '*kernel_trace.csv' -size +4M -delete
for f in $O/*.err; do echo "== $f"; grep -m3 -E "SIGSEGV|Aborted|rror" $f | cut -c1-200; done
find $O -name '*counter_collection.csv' -size +8M -delete
exit 0
