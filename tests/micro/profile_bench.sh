#!/bin/bash
# rocprofv3 summaries of bench.py (BASELINE config 2) quoted in profiles/: kernel trace + stats with one and with two
# batches in flight, then the PMC passes (each counter set in its own run, --kernel-trace only).  Run on the GPU box from the
# repo root; every pass prints a line first (a silent call is taken for hung).  Writes gpurun_out/profb/pmc_kernel.json in
# the format bench.py reads (copy it to profiles/ to have roofline.traffic / valu_* reported).
set -e
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/profb
rm -rf $O; mkdir -p $O
cd /tmp
# (--no-extras: only the timed loop and the serial leg run, one batch in flight, so EVERY launch of the fill kernel in this
#  trace runs alone -- its --stats average is the figure roofline.kernel_ms reports; the e2e leg keeps three batches in flight)
echo "stats, --inflight 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inflight1 -- python3 $R/bench.py --inflight 1 --steps 10 --warmup 3 --min-seconds 0 --no-extras > $O/bench_inflight1.log 2>&1
echo "stats, default"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python3 $R/bench.py > $O/bench_default.log 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  echo "pass $c"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --min-seconds 0 > $O/pmc_$n.log 2>&1
done
cd $R
for d in $O/inflight1 $O/default; do f=$(find $d -name '*kernel_stats.csv' | head -1); echo "== $f"; head -6 $f | cut -c1-220; cp $f $O/kernel_stats_$(basename $d).csv; done
python3 tests/micro/pmc_sum.py $O "k_fill16<8, false, 3, true>" --json $O/pmc_kernel.json --library-name "k_fill16<8, false> x4 matrix" --pairs 10000
grep -h '^{"metric"' $O/bench_inflight1.log > $O/bench_inflight1_under_rocprof.json
grep -h '^{"metric"' $O/bench_default.log > $O/bench_default_under_rocprof.json
