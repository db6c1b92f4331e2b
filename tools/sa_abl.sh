#!/bin/bash
# usage: bash tools/sa_abl.sh  -> tools/sa_bench.py with the release library and with every rdst_amd/lib_sa*.so ablation variant
echo "== release"; python tools/sa_bench.py 20 2>&1 | grep "C="
for l in rdst_amd/lib_sa*.so; do echo "== $l"; RDST_HIP_LIB=$PWD/$l python tools/sa_bench.py 20 2>&1 | grep "C=" | grep "shift=0"; done
