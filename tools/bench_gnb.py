import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pti_ldm_vae_amd import ops
dev=torch.device("cuda:0")
for (c,s) in ((32,256),(64,128),(128,64),(128,32)):
    x=torch.randn(32,s,s,c,device=dev).half(); dy=torch.randn(32,s,s,c,device=dev).bfloat16(); dres=torch.randn_like(dy); dx=torch.empty_like(dy)
    st=ops.gn_stats(x,16); g=torch.ones(c,device=dev); b=torch.zeros(c,device=dev); sums=torch.randn(32,c,2,device=dev); dg=torch.zeros(c,device=dev); db=torch.zeros(c,device=dev)
    f=lambda: ops.gn_bwd_apply(x,dy,dx,st,g,b,sums,dg,db,groups=16,dres=dres)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/20*1e3
    print(f"c={c} s={s}: {t:7.1f} us {x.numel()*8/t/1e6:6.2f} TB/s")
