"""Feasibility probe: capture one native training step (fixed eps, Adam outside) into a HIP graph and replay it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from pti_ldm_vae_amd.models import VAEModel
from pti_ldm_vae_amd.trainer import VAETrainer
from pti_ldm_vae_amd.utils import read_config
dev = torch.device("cuda:0")
cfg = read_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "vae_dente_no_adv.json"))
torch.manual_seed(42)
model = VAEModel.from_config(cfg["autoencoder_def"]).to(dev)
tr = VAETrainer(model, lr=2.5e-5)
x = synthetic_batch(int(os.environ.get("BATCH", "32")), 1, 256, dev, 42)
eps = torch.randn(x.shape[0], 4, 32, 32, device=dev)
for _ in range(3):
    tr.step(x, eps)
torch.cuda.synchronize()
tr.max_steps_in_flight = 1 << 20
tr._step_done.clear()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g, stream=s):
        out = tr.step(x, eps)
    print("captured ok")
except Exception as ex:
    print("capture failed:", repr(ex)[:2000])
    sys.exit(0)
tr._step_done.clear()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph replay: host {1e3 * (t1 - t0) / 20:.3f} ms/step, wall {1e3 * (t2 - t0) / 20:.3f} ms/step, loss {out['loss'].item():.5f}")
t0 = time.perf_counter()
for _ in range(20):
    tr.step(x, eps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host {1e3 * (t1 - t0) / 20:.3f} ms/step, wall {1e3 * (t2 - t0) / 20:.3f} ms/step")
