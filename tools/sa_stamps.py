"""Per-phase cycle stamps of K8 (diagnostic library rdst_amd/lib_sa_stamps.so built with -DSA_STAMPS): one launch per width.
usage: RDST_HIP_LIB=$PWD/rdst_amd/lib_sa_stamps.so SA_STAMPS_N=1 python tools/sa_stamps.py C"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from rdst_amd import _lib
from test_swinattn_gpu import _params, _fused, DEV
from util import rand
lib = _lib.load()
C = int(sys.argv[1]); B, H, W = 32, 64, 64
P = {k: v.to(DEV).contiguous() for k, v in _params(C, 3).items()}
x = rand((B * H * W, C), 7).to(DEV).bfloat16()
_fused(lib, _lib, x, C, P, B, H, W, C, 4, (C // 6) ** -0.5)
torch.cuda.synchronize()
