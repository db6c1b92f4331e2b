"""Which parameter gradients are NOT written in place into the flat bucket (debug): python tools/count_copies.py"""
import collections, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rdst_amd import dp, optim

dev = torch.device("cuda:0")
net = bench.build_net(dev, torch.bfloat16)
bucket = dp.FlatGradBucket(net.parameters())
names = {id(p): n for n, p in net.named_parameters()}
x = torch.rand(32, 1, 64, 64, device=dev); tgt = torch.rand(32, 1, 256, 256, device=dev)
for it in range(2):
    bucket.detach_grads()
    loss = F.l1_loss(net(x), tgt)
    loss.backward()
    base = bucket.flat.data_ptr(); off = 0; bad = collections.Counter(); nb = 0
    for p in bucket.params:
        g = p.grad
        if g is None or g.data_ptr() != base + 4 * off:
            nb += 1
            n = names[id(p)]
            bad[".".join(w for w in n.split(".") if not w.isdigit())] += 1
        off += p.numel()
    print("not in place:", nb, "of", len(bucket.params))
    for k, v in bad.most_common(20):
        print(f"  {v:4d} {k}")
    bucket.gather()
