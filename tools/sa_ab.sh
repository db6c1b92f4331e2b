#!/bin/bash
# usage: bash tools/sa_ab.sh [rounds]  -> tools/sa_bench.py alternating between the release library and every rdst_amd/lib_sa*.so
# variant on ONE box (box-to-box spread is larger than most of the effects being compared)
N=${1:-3}
for i in $(seq $N); do
  echo "== release"; python tools/sa_bench.py 30 2>&1 | grep "C=" | awk '{print $1, $2, $4}' | tr '\n' ' '; echo
  for l in rdst_amd/lib_sa*.so; do echo "== $l"; RDST_HIP_LIB=$PWD/$l python tools/sa_bench.py 30 2>&1 | grep "C=" | awk '{print $1, $2, $4}' | tr '\n' ' '; echo; done
done
