import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle.autoencoderkl import CONFIG_A, synthetic_images
from pti_ldm_vae_amd.models import VAEModel
dev=torch.device("cuda:0")
torch.manual_seed(0)
model = VAEModel.from_config(CONFIG_A).to(dev).eval()
def rel(a,b): return ((a-b).norm()/b.norm()).item()
for B,S in ((32,256),(4,256),(4,64)):
    x = synthetic_images(B, 1, S, seed=11).to(dev)
    eps = torch.randn(B, 4, S//8, S//8, generator=torch.Generator().manual_seed(12)).to(dev)
    outs=[]
    with torch.no_grad():
        for _ in range(3):
            mu, sig = model.autoencoder.encode(x); rec = model.autoencoder.decode(mu + eps*sig)
            outs.append((rec.clone(), mu.clone()))
    print(B,S,"run-to-run recon relL2", rel(outs[1][0],outs[0][0]), rel(outs[2][0],outs[0][0]), "mu", rel(outs[1][1],outs[0][1]))
