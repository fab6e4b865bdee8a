import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
CONFIG_A = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64, 128, 128],
                num_res_blocks=2, norm_num_groups=16, norm_eps=1e-6, attention_levels=[False] * 4,
                with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)   # BASELINE.json configs[1]


def synthetic_images(batch, channels, size, seed=42):
    """z-scored elliptical foreground on an exact-zero background (the bench's synthetic input)."""
    x = torch.randn(batch, channels, size, size, generator=torch.Generator().manual_seed(seed))
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    return x * ((xx / 0.80) ** 2 + (yy / 0.64) ** 2 <= 1.0).float()


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

# ---- training step: same weights, same batch, same noise -> same bits --------------------------------------------
from pti_ldm_vae_amd.trainer import VAETrainer  # noqa: E402
x = synthetic_images(32, 1, 256, seed=21).to(dev)
eps = torch.randn(3, 32, 4, 32, 32, generator=torch.Generator().manual_seed(22)).to(dev)
state = {k: v.clone() for k, v in model.state_dict().items()}
runs = []
for _ in range(3):
    m = VAEModel.from_config(CONFIG_A).to(dev)
    m.load_state_dict(state)
    tr = VAETrainer(m, lr=1e-4)
    ls = [tr.step(x, eps[i])["loss"] for i in range(3)]
    torch.cuda.synchronize()
    runs.append((m.autoencoder.param_arena.clone(), [v.item() for v in ls]))
for a, ls in runs[1:]:
    print("train 3 steps @32x256^2: params bitwise equal", torch.equal(a, runs[0][0]), "max|diff|",
          (a - runs[0][0]).abs().max().item(), "losses equal", ls == runs[0][1], ls)
