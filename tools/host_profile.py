"""Where the host time of one native training step goes (cProfile over 20 steps with the in-flight cap lifted)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from pti_ldm_vae_amd.models import VAEModel
from pti_ldm_vae_amd.trainer import VAETrainer
from pti_ldm_vae_amd.utils import read_config
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = read_config(os.path.join(root, "config", "vae_dente_no_adv.json"))
torch.manual_seed(42)
model = VAEModel.from_config(cfg["autoencoder_def"]).to(dev)
tr = VAETrainer(model, lr=2.5e-5)
x = synthetic_batch(32, 1, 256, dev, 42)
for _ in range(5):
    tr.step(x)
torch.cuda.synchronize()
tr.max_steps_in_flight = 1 << 20
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(x)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
