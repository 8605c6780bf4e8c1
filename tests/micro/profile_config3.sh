#!/bin/bash
# rocprofv3 summaries of BASELINE config 3 (tests/micro/config3.py: one 100 kb x 100 kb standard-mode LOCAL pair on the strip
# pipeline) quoted in profiles/: kernel trace + stats, then the PMC passes (each counter set in its own run, --kernel-trace
# only, as the pool requires).  Run on the GPU box from the repo root.
set -e
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/prof3
rm -rf $O; mkdir -p $O
cd /tmp
echo "pass kernel-trace --stats"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/tests/micro/config3.py 100000 > $O/c3.log 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  echo "pass $c"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tests/micro/config3.py 100000 > $O/pmc_$n.log 2>&1
done
cd $R
f=$(find $O/c3 -name '*kernel_stats.csv' | head -1); echo "== $f"; head -8 $f | cut -c1-200; cp $f $O/kernel_stats_config3.csv
grep -h "fill" $O/c3.log | head -2
python3 tests/micro/pmc_sum.py $O k_fill_strip
