#!/bin/bash
# A/B of two builds of the library over the lane-packed / wide-lane shapes: tests/micro/ab_occupancy.sh pwlib.so pwlib_b.so
A=${1:-pwlib.so}; B=${2:-pwlib_b.so}
for so in $A $B $A $B; do
  echo "== $so config 4"
  PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 300 python3 tests/micro/overlap_all_bench.py 50000 16 -1 2>/dev/null | grep -i "align\|band\|wall\|kernel" | cut -c1-200 || exit 1
done
for so in $A $B; do
  echo "== $so band sweep"
  PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 300 python3 tests/micro/band_sweep.py 2>/dev/null | cut -c1-200 || exit 1
  echo "== $so shape sweep (small)"
  PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 400 python3 tests/micro/shape_sweep_small.py 2>/dev/null | cut -c1-200 || exit 1
  echo "== $so shape sweep"
  PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 400 python3 tests/micro/shape_sweep.py 2>/dev/null | cut -c1-200 || exit 1
done
