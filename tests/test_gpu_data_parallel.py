"""GPU test of the data-parallel engine path (reference vae_scripts/train_vae.py:282,301,
src/pti_ldm_vae/utils/distributed.py:8-29): TWO processes, each stepping the native trainer on half a batch with the
weight-gradient side stream ON and the bucketed all-reduce launched from the engine's "gradients ready" callbacks.

The one-GPU box has a single card, so both ranks use cuda:0 and exchange through gloo (RCCL wants one device per rank);
what is exercised is everything above the collective itself: events between the main stream, the side stream and the
exchange, bucket coverage, ``finish()`` before Adam, lr x world, 1/world in Adam, the constructor's parameter broadcast.

The job is launched by tests/conftest.py at session start -- BEFORE this pytest process touches the GPU (a process that
has initialised the GPU must not start other programs on this pool) -- and runs beside the other tests; this test joins
it and compares with a single-process full-batch step computed here.  Tolerances: the two-half and the full-batch
gradient differ only by fp32 summation order and bf16 rounding of slightly different partial sums: rel L2 <= 1e-3
(VERDICT r1 item 7); parameters after Adam (update = lr * g / (|g| + 1e-8), so elements with a gradient at the noise
floor may move by up to 2 lr the other way): max-abs <= 2.5 * lr * world."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_rank_native_step_equals_single_rank_full_batch(dev, dp_job):
    import dp_gpu_worker as W
    from pti_ldm_vae_amd.trainer import VAETrainer
    outdir = dp_job()
    r0, r1 = (json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in (0, 1))
    dp = torch.load(os.path.join(outdir, "native_rank0.pt"), weights_only=True)
    # single process, full batch, lr x world (the reference scales lr by the world size, train_vae.py:301)
    model = W.build_model(dev)
    x, eps = W.fixed_inputs()
    tr = VAETrainer(model, lr=W.LR * 2)
    out = tr.step(x.to(dev), eps.to(dev))
    torch.cuda.synchronize()
    ae = model.autoencoder
    g_full, p_full = ae.grad_arena.detach().cpu(), ae.param_arena.detach().cpu()
    g_dp = dp["grad_sum"] / 2                       # the arena holds the SUM over ranks; Adam applies 1/world
    rel = ((g_dp - g_full).norm() / g_full.norm()).item()
    dmax = (dp["params"] - p_full).abs().max().item()
    print(f"[dp2] loss rank0 {r0['loss']:.6f} rank1 {r1['loss']:.6f} full {out['loss'].item():.6f}; grad relL2 {rel:.2e}; "
          f"param max|diff| {dmax:.2e} (lr*world {2 * W.LR:.1e}); buckets {len(r0['launched'])}")
    assert 0.5 * (r0["loss"] + r1["loss"]) == pytest.approx(out["loss"].item(), rel=1e-4)
    assert rel <= 1e-3
    assert dmax <= 2.5 * 2 * W.LR
    assert r0["ranks_agree"] and r1["ranks_agree"]
    # every arena element went out exactly once: the launched buckets tile [0, arena) with no gap and no overlap
    for r in (r0, r1):
        spans = [tuple(s) for s in r["launched"]]
        assert spans[0][0] == 0 and spans[-1][1] == r["arena_len"], spans
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), spans
        assert len(spans) >= 4                      # really bucketed (256 KiB buckets over a ~5 MB arena)
    assert r0["launched"] == r1["launched"]         # same collective sequence on both ranks


def test_dropin_model_under_torch_ddp_wrapper(dev, dp_job):
    """INTEGRATION.md's claim: the drop-in ``VAEModel`` can be wrapped in ``DistributedDataParallel(...,
    find_unused_parameters=True)`` the way the reference does (train_vae.py:282).  The engine's autograd Functions
    hand views of the flat gradient arena as gradients; DDP's reducer must average them across ranks."""
    outdir = dp_job()
    r0, r1 = (json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in (0, 1))
    print("[ddp wrapper]", r0["ddp"], r1["ddp"])
    for r in (r0, r1):
        assert r["ddp"]["ok"], r["ddp"]
        assert r["ddp"]["finite"] and r["ddp"]["ranks_agree"] and r["ddp"]["grad_abs_sum"] > 0


def test_two_rank_adversarial_step_equals_single_rank_full_batch(dev, dp_job):
    """The adversarial branch under data parallelism: the discriminator's gradient arena is exchanged (one bucket) and
    averaged like the generator's, its parameters are broadcast at construction, and two half-batch ranks equal one
    full-batch rank (InstanceNorm is per sample, the losses are batch means)."""
    import dp_gpu_worker as W
    from pti_ldm_vae_amd.trainer import VAETrainer
    outdir = dp_job()
    r0, r1 = (json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in (0, 1))
    assert r0["adv"]["ok"] and r1["adv"]["ok"], (r0["adv"], r1["adv"])
    assert r0["adv"]["ranks_agree"] and r1["adv"]["ranks_agree"]
    dp = torch.load(os.path.join(outdir, "adv_rank0.pt"), weights_only=True)
    model, disc = W.build_model(dev), W.build_disc(dev)
    x, eps = W.fixed_inputs()
    tr = VAETrainer(model, lr=W.LR * 2, discriminator=disc, adv_weight=0.1, adv_no_activation_leastsq=True)
    out = tr.step(x.to(dev), eps.to(dev), adversarial=True)
    torch.cuda.synchronize()
    gd_full, gg_full = disc.grad_arena.detach().cpu(), model.autoencoder.grad_arena.detach().cpu()
    rel_d = ((dp["d_grad_sum"] / 2 - gd_full).norm() / gd_full.norm()).item()
    rel_g = ((dp["g_grad_sum"] / 2 - gg_full).norm() / gg_full.norm()).item()
    dmax = (dp["d_params"] - disc.param_arena.detach().cpu()).abs().max().item()
    print(f"[dp2 adv] D grad relL2 {rel_d:.2e}, G grad relL2 {rel_g:.2e}, D param max|diff| {dmax:.2e}; adv_disc "
          f"{0.5 * (r0['adv']['adv_disc'] + r1['adv']['adv_disc']):.5f} vs {out['adv_disc'].item():.5f}")
    assert rel_d <= 2e-3 and rel_g <= 2e-3
    assert dmax <= 2.5 * 2 * W.LR
    assert 0.5 * (r0["adv"]["adv_disc"] + r1["adv"]["adv_disc"]) == pytest.approx(out["adv_disc"].item(), rel=1e-3)


def test_rccl_backend_exchange_single_rank(dev, rccl_job):
    """The exchange path on the REAL backend of the multi-GPU run (nccl = RCCL), one rank (tests/rccl_one_rank_worker.py):
    the bucketed all-reduces are launched from the engine's callbacks behind the side stream and joined before Adam; with
    one rank SUM is the identity, so the gradient arena must be bit-identical to the same callback-driven step without a
    collective behind it (and within fp32 summation order of the plain, batched step)."""
    outdir = rccl_job()
    r = json.load(open(os.path.join(outdir, "rccl.json")))
    print("[rccl x1]", {k: v for k, v in r.items() if not k.startswith("launched")}, "buckets", len(r["launched"]))
    assert r["backend"] == "nccl"
    assert r["grads_bit_equal"] and r["n_diff"] == 0 and r["params_finite"]
    assert r["plain_vs_callback_rel"] <= 1e-6
    assert r["loss_plain"] == r["loss_forced"]
    spans = [tuple(s) for s in r["launched"]]
    assert spans[0][0] == 0 and spans[-1][1] == r["arena_len"] and len(spans) >= 4
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert r["launched2"] == r["launched"]
    assert r["loss_step2"] < r["loss_plain"] * 1.5      # the second step ran on updated parameters and stayed sane


def test_graph_step_mode_under_data_parallelism(dev, dp_job):
    """HIP-graph step mode for world > 1 (``VAETrainer._step_graphed_dp``): [forward + loss + decoder
    backward] and [encoder backward] are two captured graphs, the bucketed all-reduce of each region is launched eagerly
    after the graph that completes it (the decoder region's exchange runs under the second graph), Adam stays eager.
    Two ranks on the one GPU over gloo: the third step of a graph-mode trainer against the third step of an eager trainer
    started from the same weights -- same loss, all-reduced gradients equal up to the fp32 summation order of the batched
    weight-gradient launches (rel-L2 <= 1e-5), parameters within one Adam step of each other on noise-floor elements,
    every arena element exchanged exactly once, both ranks bit-identical."""
    import dp_gpu_worker as W
    outdir = dp_job()
    for r in (0, 1):
        g = json.load(open(os.path.join(outdir, f"rank{r}.json")))["graph_dp"]
        assert g["ok"], g
        assert g["graphs"] == [0, 1], g["graphs"]                   # eager trainer: none; graph trainer: one (two-graph) entry
        assert g["loss_graph"][:2] == g["loss_eager"][:2]           # the two eager warm-up steps are the same code path
        assert g["loss_graph"][2] == pytest.approx(g["loss_eager"][2], rel=1e-5)
        print(f"[dp2 graph] rank {r}: grad relL2 {g['grad_rel']:.2e}, param max|diff| {g['param_maxdiff']:.2e}, buckets {len(g['launched'])}")
        assert g["grad_rel"] <= 1e-5
        assert g["param_maxdiff"] <= 2.5 * 2 * W.LR
        assert g["ranks_agree"]
        spans = [tuple(s) for s in g["launched"]]
        assert spans[0][0] == 0 and spans[-1][1] == g["arena_len"] and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), spans
        assert len(spans) >= 4


def test_bench_two_gpu_line_from_the_drivers_launch_command(dev, dp_job):
    """``bench.py --gpus 2`` launched the way the driver launches it (torch.distributed.run, one process per rank; here in
    the one-GPU rehearsal form: gloo, both ranks on cuda:0 -- tests/dp_jobs_launcher.py): rank 0 prints exactly ONE JSON line
    with ``n_gpus: 2``, ``parallelism: dp2``, the global batch of both ranks, weak scaling, under 4 KB."""
    outdir = dp_job()
    assert open(os.path.join(outdir, "bench_dp2.rc")).read().strip() == "0", open(os.path.join(outdir, "bench_dp2.err")).read()[-3000:]
    lines = [l for l in open(os.path.join(outdir, "bench_dp2.out")).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 3 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "cpu_baseline" not in d                               # rank 0 at N = 1 only
    # (which family tops the instrumented steps is not asserted: at batch 2 with both ranks sharing one GPU it varies)
    # nor is its fraction (two processes time-slicing one GPU: launches of tens of milliseconds, frac rounds to 0)
    assert d["roofline"] and isinstance(d["roofline"]["kernel"], str) and d["roofline"]["frac"] >= 0
