"""Worker of tests/test_gpu_data_parallel.py: ONE rank of a world-size-2 data-parallel job whose ranks share cuda:0
(gloo process group; the one-GPU box has no second card, and RCCL needs one device per rank).  Launched by
``python -m torch.distributed.run`` from a process that has not touched the GPU (tests/conftest.py).

Each rank runs the NATIVE trainer (``VAETrainer.step``: HIP forward / backward, weight gradients on the side stream,
``grad_ready_cb`` -> ``FlatGradAllReducer.ready`` enqueued behind that stream, ``finish()`` before Adam) on ITS half of a
fixed batch and rank 0 writes what the test checks: the all-reduced gradient arena, the parameters after the step,
the list of buckets the reducer launched.  Then the drop-in path: the same model wrapped in
``DistributedDataParallel(..., find_unused_parameters=True)`` exactly as the reference wraps it (train_vae.py:282),
one forward / backward through autograd; rank 0 writes the DDP-averaged gradients."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64, 128], num_res_blocks=1,
           norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False, False], with_encoder_nonlocal_attn=True,
           with_decoder_nonlocal_attn=True)
BATCH, SIZE, LR, SEED = 4, 64, 1e-4, 1234


def fixed_inputs():
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(BATCH, 1, SIZE, SIZE, generator=g)
    eps = torch.randn(BATCH, 4, SIZE // 4, SIZE // 4, generator=g)
    return x, eps


def build_model(dev):
    from pti_ldm_vae_amd.models import VAEModel
    torch.manual_seed(SEED)
    return VAEModel.from_config(CFG).to(dev)


def build_disc(dev):
    from pti_ldm_vae_amd.models import PatchDiscriminator
    torch.manual_seed(SEED + 1)
    return PatchDiscriminator().to(dev)


def main(outdir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", init_method="env://")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from pti_ldm_vae_amd.models import compute_kl_loss
    from pti_ldm_vae_amd.trainer import VAETrainer
    x, eps = fixed_inputs()
    lo, hi = rank * BATCH // world, (rank + 1) * BATCH // world
    xs, es = x[lo:hi].to(dev), eps[lo:hi].to(dev)

    # ---- native trainer, side stream ON (the default), tiny buckets so that several go out during backward ----
    model = build_model(dev)
    if rank == 1:                      # the constructor's broadcast must bring every rank to rank 0's weights
        with torch.no_grad():
            model.autoencoder.param_arena.add_(0.5)
    tr = VAETrainer(model, lr=LR, world_size=world, bucket_bytes=256 << 10)
    assert tr.eng.wgrad_stream is not None, "the side stream must be on for this test"
    out = tr.step(xs, es)
    torch.cuda.synchronize()
    ae = model.autoencoder
    launched = sorted(tr.reducer.launched)
    res = {"rank": rank, "loss": out["loss"].item(), "launched": launched, "arena_len": ae.grad_arena.numel()}
    if rank == 0:
        torch.save({"grad_sum": ae.grad_arena.detach().cpu(), "params": ae.param_arena.detach().cpu()},
                   os.path.join(outdir, "native_rank0.pt"))
    # every rank must hold the same reduced gradients and the same parameters afterwards
    chk = torch.stack([ae.grad_arena.double().sum(), ae.param_arena.double().sum()]).cpu()
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    res["ranks_agree"] = bool(torch.equal(both[0], both[1]))

    # ---- HIP-graph step mode under data parallelism: two captured halves, collectives eager in between ----
    try:
        pair = []
        for graph in (False, True):
            mg = build_model(dev)
            trg = VAETrainer(mg, lr=LR, world_size=world, bucket_bytes=256 << 10)
            trg.step_graph = graph
            losses = [trg.step(xs, es)["loss"].item() for _ in range(3)]      # steps 0, 1 eager; step 2 captured + replayed
            torch.cuda.synchronize()
            pair.append((mg.autoencoder.grad_arena.detach().clone(), mg.autoencoder.param_arena.detach().clone(), losses,
                         len(trg._graphs), sorted(trg.reducer.launched)))
        (g_e, p_e, l_e, n_e, _), (g_g, p_g, l_g, n_g, spans) = pair
        chk = torch.stack([g_g.double().sum(), p_g.double().sum()]).cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        res["graph_dp"] = {"ok": True, "graphs": [n_e, n_g], "loss_eager": l_e, "loss_graph": l_g,
                           "grad_rel": ((g_g - g_e).norm() / g_e.norm()).item(), "param_maxdiff": (p_g - p_e).abs().max().item(),
                           "ranks_agree": bool(torch.equal(both[0], both[1])), "launched": spans, "arena_len": g_g.numel()}
    except Exception as ex:
        import traceback
        res["graph_dp"] = {"ok": False, "error": repr(ex), "trace": traceback.format_exc()}

    # ---- native trainer with the adversarial branch: the discriminator has its own arena, exchange and Adam ----
    try:
        from pti_ldm_vae_amd.models import PatchDiscriminator
        m3 = build_model(dev)
        disc = build_disc(dev)
        if rank == 1:
            with torch.no_grad():
                disc.param_arena.add_(0.25)      # the constructor must broadcast rank 0's discriminator too
        tr3 = VAETrainer(m3, lr=LR, world_size=world, discriminator=disc, adv_weight=0.1, adv_no_activation_leastsq=True)
        out3 = tr3.step(xs, es, adversarial=True)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"d_grad_sum": disc.grad_arena.detach().cpu(), "d_params": disc.param_arena.detach().cpu(),
                        "g_grad_sum": m3.autoencoder.grad_arena.detach().cpu()}, os.path.join(outdir, "adv_rank0.pt"))
        chk = torch.stack([disc.grad_arena.double().sum(), disc.param_arena.double().sum()]).cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        res["adv"] = {"ok": True, "ranks_agree": bool(torch.equal(both[0], both[1])), "adv_gen": out3["adv_gen"].item(),
                      "adv_disc": out3["adv_disc"].item()}
    except Exception as ex:
        res["adv"] = {"ok": False, "error": repr(ex)}

    # ---- drop-in path under DistributedDataParallel(find_unused_parameters=True), as train_vae.py:282 wraps it ----
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        m2 = build_model(dev)
        ddp = DDP(m2, device_ids=[0], find_unused_parameters=True)
        rec, mu, sig = ddp(xs)            # sampled forward (device RNG): only gradient CONSISTENCY is checked here
        loss = torch.nn.functional.l1_loss(rec, xs) + 1e-3 * compute_kl_loss(mu, sig)
        loss.backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.detach().flatten() for p in m2.parameters()])
        s = torch.stack([g.double().sum(), g.double().abs().sum()]).cpu()
        both = [torch.zeros_like(s) for _ in range(world)]
        dist.all_gather(both, s)
        res["ddp"] = {"ok": True, "finite": bool(torch.isfinite(g).all()), "ranks_agree": bool(torch.equal(both[0], both[1])),
                      "grad_abs_sum": float(s[1])}
    except Exception as ex:   # reported, the test decides
        res["ddp"] = {"ok": False, "error": repr(ex)}
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
