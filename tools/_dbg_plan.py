import sys; sys.path.insert(0,'/root/repo')
import torch, bench, collections
import torch.nn.functional as F
from rdst_amd import ops, dp
dev=torch.device('cuda:0')
net=bench.build_net(dev, torch.bfloat16)
bucket=dp.FlatGradBucket(net.parameters())
x=torch.rand(2,1,64,64,device=dev); t=torch.rand(2,1,256,256,device=dev)
bucket.detach_grads()
F.l1_loss(net(x),t).backward()
base=bucket.flat.data_ptr(); off=0
names={id(p):n for n,p in net.named_parameters()}
cnt=collections.Counter()
for p in bucket.params:
    g=p.grad
    if g is None: cnt['none:'+names[id(p)].split('.')[-2]+'.'+names[id(p)].split('.')[-1]]+=1
    elif g.data_ptr()!=base+4*off: cnt[names[id(p)].split('.')[-2]+'.'+names[id(p)].split('.')[-1]]+=1
    off+=p.numel()
print(cnt)
