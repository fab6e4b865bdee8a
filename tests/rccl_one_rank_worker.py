"""Worker of tests/test_gpu_data_parallel.py::test_rccl_backend_exchange_single_rank: ONE process on cuda:0 whose process
group is the REAL backend of the multi-GPU run -- ``nccl`` (= RCCL on ROCm) -- with world size 1 (the one-GPU box cannot
hold two RCCL ranks: RCCL wants one device per rank).  The native trainer is stepped twice on the same input:

  * plain: world 1, no exchange (graph-replayed, weight gradients batched per launch);
  * callback: the engine's "gradients ready" callbacks installed, but no collective behind them -- the reference for
    bit-equality (the callback path batches weight gradients per block, so it differs from the plain step by fp32
    summation order, ~5e-8 relative);
  * forced: the same trainer class told that the world has two ranks (``tr.world = tr.reducer.world = 2``), so the engine
    installs the "gradients ready" callbacks and every bucket really goes through ``dist.all_reduce(..., async_op=True)``
    on RCCL's own stream behind the weight-gradient side stream, and ``finish()`` joins them before Adam.  With one rank
    the SUM is the identity, so the gradient arena must come out BIT-IDENTICAL to the callback step's.

What this covers that the two-rank gloo job cannot: RCCL initialises on the box (dmabuf IPC mode), accepts slices of the
flat arena as in-place buffers from a non-default stream, and its work handles chain back into the compute stream.
Launched by tests/conftest.py before the pytest process touches the GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(outdir):
    import dp_gpu_worker as W
    from pti_ldm_vae_amd.trainer import VAETrainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    x, eps = W.fixed_inputs()
    x, eps = x.to(dev), eps.to(dev)
    res = {"backend": dist.get_backend()}

    m0 = W.build_model(dev)
    tr0 = VAETrainer(m0, lr=W.LR, world_size=1)
    out0 = tr0.step(x, eps)
    torch.cuda.synchronize()
    g0 = m0.autoencoder.grad_arena.detach().clone()

    # the same callback-driven step WITHOUT any collective (the reducer still believes in one rank): the reference
    # for bit-equality, since the callback path may batch weight gradients differently from the plain step
    mc = W.build_model(dev)
    trc = VAETrainer(mc, lr=W.LR, world_size=1, bucket_bytes=256 << 10)
    trc.world = 2
    trc.step(x, eps)
    torch.cuda.synchronize()
    gc = mc.autoencoder.grad_arena.detach().clone()

    m1 = W.build_model(dev)
    tr1 = VAETrainer(m1, lr=W.LR, world_size=1, bucket_bytes=256 << 10)
    tr1.world = tr1.reducer.world = 2          # force the exchange path; one real rank => SUM is the identity
    assert tr1.eng.wgrad_stream is not None
    out1 = tr1.step(x, eps)
    torch.cuda.synchronize()
    g1 = m1.autoencoder.grad_arena.detach()
    res.update(loss_plain=out0["loss"].item(), loss_forced=out1["loss"].item(), grads_bit_equal=bool(torch.equal(gc, g1)),
               plain_vs_callback_rel=((g0 - gc).norm() / g0.norm()).item(),
               callback_vs_rccl_rel=((gc - g1).norm() / gc.norm()).item(),
               n_diff=int((gc != g1).sum().item()),
               launched=sorted(tr1.reducer.launched), arena_len=g1.numel(),
               params_finite=bool(torch.isfinite(m1.autoencoder.param_arena).all()))
    # a second step through the same process group (work handles of step 1 are gone, buckets are re-launched)
    out2 = tr1.step(x, eps)
    torch.cuda.synchronize()
    res.update(loss_step2=out2["loss"].item(), launched2=sorted(tr1.reducer.launched))
    with open(os.path.join(outdir, "rccl.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
