"""Which ops of an RDST-E1 forward find their packed weight image in the PackPlan, and which pack for themselves.
Run on the GPU box: python tools/pack_audit.py"""
import collections, sys, torch
sys.path.insert(0, ".")
import bench
from rdst_amd import ops, _lib

dev = torch.device("cuda:0")
net = bench.build_net(dev, torch.bfloat16)
x = torch.randn(4, 1, 64, 64, device=dev)
orig = ops._packed_workspace
log = collections.Counter()


def audit(kind, w, lw, lb, b, N, K, s, nbytes, device):
    r = orig(kind, w, lw, lb, b, N, K, s, nbytes, device)
    log[(kind, N, K, round(float(s), 6), "plan" if r[2] == _lib.PREPACKED else "self")] += 1
    return r


ops._packed_workspace = audit
for it in range(3):
    log.clear()
    with torch.no_grad():
        net(x)
    torch.cuda.synchronize()
    plan = ops.pack_plan_of(net)
    print(f"forward {it}: plan specs {None if plan is None else len(plan.specs)} misses {None if plan is None else plan.misses}")
    for k, v in sorted(log.items()):
        print("   ", k, v)
