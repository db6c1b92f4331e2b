#!/bin/bash
# usage: tools/isa.sh <file.hip> <kernel-name-regex>  -> /tmp/k.s (that kernel's ISA) + resource summary
f=$1; k=$2
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I/root/repo/rdst_amd/csrc /root/repo/rdst_amd/csrc/$f -o all.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|VGPRs Spill|Scratch" | grep -A5 -E "error|$k"
awk "/^[A-Za-z0-9_]*${k}[A-Za-z0-9_]*:/,/s_endpgm/" all.s > k.s
echo "lines $(wc -l < k.s) mfma $(grep -c v_mfma k.s) accread $(grep -c v_accvgpr_read k.s) accwrite $(grep -c v_accvgpr_write k.s) ds_read $(grep -c ds_read k.s) saveexec $(grep -c s_and_saveexec k.s) scratch $(grep -c scratch_ k.s)"
