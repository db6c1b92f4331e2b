#!/bin/bash
# build the library (release), run the CPU-side export check, then run a command on the GPU box: tools/go.sh <timeout> '<cmd>'
set -e
cd /root/repo
python -m rdst_amd.build 2>&1 | grep -E "error|librdst|failed" | head -8
python -m pytest tests/test_host_logic.py -x -q 2>&1 | tail -1
/usr/local/graft/bin/gpurun --timeout $1 -- "$2"
