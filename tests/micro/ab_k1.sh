#!/bin/bash
# A/B of two builds of the library on the config-2 batch: tests/micro/ab_k1.sh pwlib.so pwlib_b.so [rounds]
A=${1:-pwlib.so}; B=${2:-pwlib_b.so}; R=${3:-2}
for i in $(seq 1 $R); do
  for so in $A $B; do
    echo "== $so"
    PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 300 python tests/micro/k1_variants.py 8 2>/dev/null || exit 1
  done
done
