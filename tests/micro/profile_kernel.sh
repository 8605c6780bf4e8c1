#!/bin/bash
# rocprofv3 PMC passes of ONE forced kernel on config 2's shape: tests/micro/profile_kernel.sh <f64|i32|generic|matrix> "<kernel substring>"
# (each counter set in its own run, --kernel-trace only, as the pool requires).  Prints the per-launch sums; run from the repo root.
set -e
W=$1; SUB=$2
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/profk_$W
rm -rf $O; mkdir -p $O
cd /tmp
echo "stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tests/micro/run_batch.py $W 5 > $O/stats.log 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  echo "pass $c"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tests/micro/run_batch.py $W 3 > $O/pmc_$n.log 2>&1
done
cd $R
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); head -4 $f | cut -c1-200
find $O/stats -name '*kernel_trace.csv' | head -1 | xargs -I{} python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open('{}')) if '$SUB' in r['Kernel_Name']]
if rows: print('kernel resources: VGPR', rows[0]['VGPR_Count'], 'AGPR', rows[0]['Accum_VGPR_Count'], 'SGPR', rows[0]['SGPR_Count'], 'scratch', rows[0]['Scratch_Size'], 'LDS', rows[0]['LDS_Block_Size'])"
python3 tests/micro/pmc_sum.py $O "$SUB"
