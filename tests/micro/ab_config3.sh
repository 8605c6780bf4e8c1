#!/bin/bash
# A/B of two builds of the library on the same box: tests/micro/ab_config3.sh pwlib.so pwlib_b.so [rounds]
# (build the second one with PW_OBJ_DIR=_build_b PW_SO_NAME=pwlib_b.so PW_EXTRA_CXXFLAGS=... python -m biseqt_amd.csrc.build)
A=${1:-pwlib.so}; B=${2:-pwlib_b.so}; R=${3:-3}
for i in $(seq 1 $R); do
  for so in $A $B; do
    echo "== $so"
    PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 200 python tests/micro/config3.py 100000 2>/dev/null | grep "fill" || exit 1
    PWLIB_SO=$PWD/biseqt_amd/pwlib/$so timeout -k 10 200 python tests/micro/strip_probe.py 2>/dev/null | sed -n '1p;4p;6p' || exit 1
  done
done
