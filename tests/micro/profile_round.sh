#!/bin/bash
# Collects the rocprofv3 summaries quoted in profiles/ (run on the GPU box from the repo root):
#   kernel trace + stats of config 3 and of bench.py --inflight 1, then the PMC passes of config 3 (counters in their own
#   runs, --kernel-trace only, as the pool requires; WRITE_SIZE and FETCH_SIZE do not fit one pass).
set -e
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/tests/micro/config3.py 100000 > $O/c3.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --inflight 1 --steps 10 --warmup 3 > $O/bench_inflight1.log 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  echo "pass $c"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tests/micro/config3.py 100000 > $O/pmc_$n.log 2>&1
done
cd $R
for d in $O/c3 $O/bench; do f=$(find $d -name '*kernel_stats.csv' | head -1); echo "== $f"; head -12 $f; done
python3 tests/micro/pmc_sum.py $O k_fill_strip
