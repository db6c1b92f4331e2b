import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load(); DEV="cuda:0"
torch.manual_seed(0)
M,K,N=int(sys.argv[1]),60,180
x=torch.randn(M,K); gy=torch.randn(M,N); w=torch.randn(N,K)*K**-0.5; lw=1+0.1*torch.randn(K); lb=0.1*torch.randn(K)
xg,gyg=x.to(DEV).bfloat16(),gy.to(DEV).bfloat16()
xf=xg.float(); stats=torch.stack([xf.mean(-1),(xf.var(-1,unbiased=False)+1e-5).rsqrt()],dim=1).contiguous()
P=[t.to(DEV).contiguous() for t in (w,lw,lb)]
dx=torch.empty_like(xg); dW=torch.empty_like(P[0]); db=torch.empty(N,device=DEV); dlw=torch.empty(K,device=DEV); dlb=torch.empty(K,device=DEV)
nws=lib.rdst_ln_linear_bwd_workspace(M,K,N); wsp=torch.empty(nws,dtype=torch.uint8,device=DEV)
st=torch.cuda.current_stream().cuda_stream
_lib.check(lib.rdst_ln_linear_bwd(xg.data_ptr(),K,P[1].data_ptr(),P[2].data_ptr(),stats.data_ptr(),0,P[0].data_ptr(),gyg.data_ptr(),N,dx.data_ptr(),K,None,K,dW.data_ptr(),db.data_ptr(),dlw.data_ptr(),dlb.data_ptr(),wsp.data_ptr(),nws,M,K,N,1.0,_lib.BF16,st),"x")
torch.cuda.synchronize()
xh=(xf-stats[:,:1])*stats[:,1:]
xh=xh.bfloat16().float()
G=gyg.float().T@xh          # dY^T xhat
dbr=gyg.float().sum(0); dWref=G*P[1][None,:]+dbr[:,None]*P[2][None,:]
err=(dW-dWref)
print("M",M,"rel",(err.norm()/dWref.norm()).item(),"db rel",((db-gyg.float().sum(0)).norm()/gyg.float().sum(0).norm()).item())
# which tile is wrong? recompute leaving tiles out
nt=(M+31)//32
for t in range(nt):
    Gt=gyg[t*32:(t+1)*32].float().T@xh[t*32:(t+1)*32]*P[1][None,:]
    # project error on tile contribution
    c=(err*Gt).sum()/(Gt*Gt).sum()
    if abs(c.item())>0.2: print(" tile",t,"coef",round(c.item(),3))
