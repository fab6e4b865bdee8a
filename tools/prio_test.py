"""Run ON THE GPU BOX: does HIP stream priority move the training step?  The step's main stream (forward, data
gradients, GroupNorm backward) is the critical path; the weight gradients run on a side stream and compete for the
same CUs.  Times the bench workload (config A, batch 32, 256x256x1) with the step issued (a) on torch's default stream
(priority 0, as the side stream) and (b) on a HIGH-priority stream, interleaved, a few rounds each.
Usage: python tools/prio_test.py [rounds] [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    from pti_ldm_vae_amd.utils import read_config
    cfg = read_config(os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = VAEModel.from_config(cfg["autoencoder_def"]).to(dev)
    tr = VAETrainer(model, lr=1e-4, recon_loss=cfg["autoencoder_train"]["recon_loss"], kl_weight=cfg["autoencoder_train"]["kl_weight"])
    images = torch.randn(32, 1, 256, 256, device=dev)
    lo, hi = torch.cuda.Stream.priority_range()
    print(f"priority range (least, greatest) = ({lo}, {hi})")
    hp = torch.cuda.Stream(device=dev, priority=hi)
    for _ in range(6):
        tr.step(images)
    torch.cuda.synchronize()

    def timed(stream):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if stream is None:
            for _ in range(steps):
                tr.step(images)
        else:
            with torch.cuda.stream(stream):
                for _ in range(steps):
                    tr.step(images)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    with torch.cuda.stream(hp):
        for _ in range(3):
            tr.step(images)
    torch.cuda.synchronize()
    for r in range(rounds):
        a = timed(None)
        b = timed(hp)
        print(f"round {r}: default stream {a:.3f} ms/step   high-priority main stream {b:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
