"""Which parameter gradients of one training step are NOT written straight into the flat bucket (FlatGradBucket.gather copies them:
one __amd_rocclr_copyBuffer each)?   python tools/grad_copies.py [bf16|fp32x3]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rdst_amd import dp
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
net = bench.build_net(dev, torch.bfloat16 if mode == "bf16" else mode)
bucket = dp.FlatGradBucket(net.parameters())
x = torch.rand(4, 1, 64, 64, device=dev); tgt = torch.rand(4, 1, 256, 256, device=dev)
bucket.detach_grads()
torch.nn.functional.l1_loss(net(x).float(), tgt).backward()
base, off = bucket.flat.data_ptr(), 0
names = {p: n for n, p in net.named_parameters()}
kinds = collections.Counter()
for p in bucket.params:
    g = p.grad
    if g is None:
        kinds["(no gradient) " + names[p].split(".")[-2] + "." + names[p].split(".")[-1]] += 1
    elif g.data_ptr() != base + 4 * off:
        kinds[".".join(names[p].split(".")[-3:])] += 1
    off += p.numel()
for k, v in kinds.most_common():
    print(v, k)
print("total copied / zeroed:", sum(kinds.values()), "of", len(bucket.params))
