"""Probe: does running the step's main chain on a HIGH-priority stream (weight gradients stay on the normal-priority
side stream) shorten the step?  The batched weight-gradient kernel holds a whole CU's LDS per workgroup, so main-chain
kernels queue behind it for CUs; priority could let them jump the queue."""
import os, sys, time
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


def run(tag, stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(8):
            tr.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            tr.step(x)
        torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)


print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
hi = torch.cuda.Stream(device=dev, priority=-1)
for r in range(2):
    run("default stream      ", None)
    run("high-priority stream", hi)
