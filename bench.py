#!/usr/bin/env python3
"""bench.py — VAE training throughput of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step of the reference's train loop body (vae_scripts/train_vae.py:380-445:
zero_grad, forward encode+sample+decode, L1 recon + 1e-3*KL, backward, gradient all-reduce, Adam) on one
batch of synthetic 256x256x1 images per GPU (config/vae_dente_no_adv.json: channels [32,64,128,128],
16 groups, latent 4; batch 32 per GPU, weak scaling).  Perceptual (LPIPS) and adversarial terms are
omitted: unavailable offline / inactive before epoch 6 — stated in ``config``.
Inputs are resident in HBM before the timed region.  For N>1 launch with
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...``.

Prints ONE JSON line (rank 0), kept under 4 KB (the driver keeps ~8 KB of stdout): metric/value/unit/..., plus
  "roofline":     the dominant kernel FUNCTION by time (template instantiations folded: conv_mfma2_kernel), timed with
                  events on the launch stream during extra instrumented steps right after the timed region (so the
                  headline number is not perturbed); achieved = algorithmic FLOP (or bytes) of its launches / their
                  summed duration.  The conv family is priced against the dense MFMA peak (north_star's target is
                  stated against it); its HBM-side fraction and the time-weighted mixed fraction ride along, and the
                  next kernels by time are listed in "secondary".
  "cpu_baseline": the CPU fp32 oracle's training step (oracle/, kind "port") on the host cores.
The per-shape table and the per-instantiation table go to a FILE (--detail-out, default gpurun_out/r03_per_shape.json;
copied to profiles/ when it is to be judged), not into the line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec, same guide (~6.3 TB/s is what a streaming kernel reaches)
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"   # tools/pmc_traffic.py output of the two --pmc passes of THIS command
TRAIN_GFLOP_PER_IMG_A = 148.11   # BASELINE.md §3 (config A, 256x256, 1 channel; fwd 49.37 x 3)


def synthetic_batch(batch, channels, size, device, seed):
    """z-scored elliptical foreground (~40 % of pixels), exact-zero background."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(batch, channels, size, size, generator=g, device=device)
    lin = torch.linspace(-1, 1, size, device=device)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    return (x * ((xx / 0.80) ** 2 + (yy / 0.64) ** 2 <= 1.0).float()).contiguous()


def cpu_baseline(cfg_def, size, batch, steps):
    """Oracle (CPU fp32 restatement) training step, timed on this box's host cores."""
    from oracle.autoencoderkl import build_oracle, synthetic_images
    from oracle.losses import train_step_losses
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    model = build_oracle(cfg_def, seed=42)
    opt = torch.optim.Adam(model.parameters(), lr=2.5e-5)
    x = synthetic_images(batch, cfg_def["in_channels"], size, seed=42)
    lat = size // (2 ** (len(cfg_def["channels"]) - 1))
    times = []
    for i in range(steps + 1):
        eps = torch.randn(batch, cfg_def["latent_channels"], lat, lat)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, *_ = train_step_losses(model, x, eps)
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    # the reference's own thread setting (torch.set_num_threads(4), train_vae.py:94): 1 warm-up + 5 timed steps
    torch.set_num_threads(min(4, cores))
    t4 = []
    for i in range(6):
        eps = torch.randn(batch, cfg_def["latent_channels"], lat, lat)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, *_ = train_step_losses(model, x, eps)
        loss.backward()
        opt.step()
        if i > 0:
            t4.append(time.perf_counter() - t0)
    torch.set_num_threads(cores)
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "value_at_reference_threads": round(batch * len(t4) / sum(t4), 4), "reference_threads": min(4, cores),
            "sample": f"oracle fp32 train step (fwd+L1+KL+bwd+Adam), config A {size}x{size}, batch {batch}, "
                      f"1 warm-up + {steps} timed steps, torch threads={cores}; plus 1 warm-up + 5 timed steps at the "
                      f"reference's torch.set_num_threads(4)"}


PMC_WORKLOAD = None   # "<config basename>:b<batch>:<size>" of this run; set by main()


def source_hash():
    """sha256 over the kernel sources: ties a committed counter file to the library it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pti_ldm_vae_amd", "csrc", "*.h*")) +
                    glob.glob(os.path.join(ROOT, "pti_ldm_vae_amd", "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _pmc():
    """The committed counter passes, but only when they were taken on THIS workload with THIS library source (else
    None: a per-launch average of another batch size / model / kernel version would be paired with the wrong launches)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
    except (OSError, ValueError):
        return None
    if pmc.get("workload") != PMC_WORKLOAD or pmc.get("source_hash") != source_hash():
        return None
    return pmc


CONV3X3 = "conv_mfma2_kernel"   # the MFMA conv family; launches of its experimental sibling conv_ws128_kernel (PTI_CONV_WS=1) fold into it, see "parts"


def family(name):
    """Kernel FUNCTION of an instantiation name: template arguments stripped; the two kernels that together implement the
    stride-1 MFMA convolution (conv_mfma.hip / conv_ws.hip, picked per layer shape) are one family."""
    i = name.find("<")
    f = (name if i < 0 else name[:i]).strip()
    return CONV3X3 if f in ("conv_mfma2_kernel", "conv_ws128_kernel") else f


def roofline_from_profile(rec, steps):
    """Aggregate ops.KERNEL_PROFILE records (kernel name as the HIP runtime reports it, algorithmic flops, bytes, start,
    end event, shape) of ``steps`` instrumented steps into (compact roofline object keyed on the dominant kernel
    FUNCTION, detail dict with the per-instantiation and per-shape tables)."""
    pmc = _pmc()
    pk = pmc["kernels"] if pmc else {}
    balance = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    fam, inst, shp = {}, {}, {}
    for name, flops, nbytes, e0, e1, shape in rec:
        dt = e0.elapsed_time(e1) * 1e-3
        for table, key in ((fam, family(name)), (inst, name), (shp, (shape[:7], name))):
            a = table.setdefault(key, [0.0, 0.0, 0.0, 0])
            a[0] += dt
            a[1] += flops
            a[2] += nbytes
            a[3] += 1
    if not fam:
        return None, {}
    # time-weighted mixed fraction of a family: every (shape, instantiation) against its own bound
    def mixed(f):
        num = den = 0.0
        for (shape, kname), (t, fl, nb, c) in shp.items():
            if family(kname) != f:
                continue
            fr = (nb / t / 1e9 / PEAK_HBM_GBS) if fl / nb < balance else (fl / t / 1e12 / PEAK_BF16_TFLOPS)
            num += fr * t
            den += t
        return num / den if den else None

    def traffic(f):
        """counter bytes per launch of the family: launch-weighted mean over its instantiations (None unless the
        committed counter file matches this workload AND this library source)."""
        tot = cnt = 0.0
        for k, v in pk.items():
            if family(k) == f:
                tot += (v["fetch_bytes"] + v["write_bytes"]) * v["launches"]
                cnt += v["launches"]
        return round(tot / cnt) if cnt else None

    def roof(f, force_mfma=False):
        tsec, flops, nbytes, cnt = fam[f]
        intensity = flops / nbytes
        tfl, gbs = flops / tsec / 1e12, nbytes / tsec / 1e9
        hbm_bound = (intensity < balance) and not force_mfma
        return {"kernel": f, "bound": "hbm" if hbm_bound else "mfma",
                "achieved": round(gbs if hbm_bound else tfl, 2), "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(gbs / PEAK_HBM_GBS if hbm_bound else tfl / PEAK_BF16_TFLOPS, 4), "traffic": traffic(f),
                "ms_per_step": round(tsec / steps * 1e3, 3), "launches_per_step": round(cnt / steps, 1),
                "avg_launch_us": round(tsec / cnt * 1e6, 2), "alg_bytes_per_launch": round(nbytes / cnt),
                "alg_gflop_per_launch": round(flops / cnt / 1e9, 3), "flop_per_byte": round(intensity, 1),
                "tflops": round(tfl, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4), "mixed_frac": round(mixed(f), 4)}

    by_time = sorted(fam, key=lambda k: -fam[k][0])
    # the conv families are priced against the MFMA peak even where their aggregate intensity sits under the ridge:
    # north_star states the conv target against it (hbm_frac / mixed_frac carry the other view)
    roofline = roof(by_time[0], force_mfma=by_time[0] == CONV3X3)
    roofline["kernel_name_source"] = "pti_last_kernel_name(), template arguments folded"
    roofline["traffic_source"] = (f"profiles/{PMC_TRAFFIC_FILE}: rocprofv3 --pmc, 2xFETCH_SIZE+WRITE_SIZE per launch" if pmc
                                  else "null: no counter file for this workload + library source")
    roofline["secondary"] = [{k: r[k] for k in ("kernel", "bound", "achieved", "unit", "frac", "traffic", "ms_per_step",
                                                 "launches_per_step")}
                             for r in (roof(f, force_mfma=f == CONV3X3) for f in by_time[1:4])]
    if by_time[0] == CONV3X3:      # the family's two kernels on their own (rocprofv3 lists them separately)
        parts = {}
        for name, (tsec, flops, nbytes, cnt) in inst.items():
            p = parts.setdefault(name[:name.find("<")] if "<" in name else name, [0.0, 0.0, 0])
            if family(name) == CONV3X3:
                p[0] += tsec
                p[1] += flops
                p[2] += cnt
        roofline["parts"] = {k: {"ms_per_step": round(v[0] / steps * 1e3, 3), "tflops": round(v[1] / v[0] / 1e12, 1),
                                 "launches_per_step": round(v[2] / steps, 1)} for k, v in parts.items() if v[2]}
    detail = {"instantiations": {k: {"ms_per_step": round(v[0] / steps * 1e3, 3), "tflops": round(v[1] / v[0] / 1e12, 1),
                                     "gbs": round(v[2] / v[0] / 1e9), "launches": round(v[3] / steps, 2)}
                                 for k, v in sorted(inst.items(), key=lambda kv: -kv[1][0])}}
    table = []
    for (shape, kname), (t, fl, nb, c) in sorted(shp.items(), key=lambda kv: -kv[1][0]):
        kind, cin, cout, ho, wo, ks, mode = shape
        row = {"op": kind, "cin": cin, "cout": cout, "out_hw": [ho, wo], "k": ks, "gather": mode, "kernel": kname,
               "launches_per_step": round(c / steps, 2), "avg_us": round(t / c * 1e6, 1), "ms_per_step": round(t / steps * 1e3, 3),
               "tflops": round(fl / t / 1e12, 1), "frac_mfma_peak": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4),
               "alg_gbs": round(nb / t / 1e9), "frac_hbm_peak": round(nb / t / 1e9 / PEAK_HBM_GBS, 4),
               "alg_mb_per_launch": round(nb / c / 1e6, 1), "flop_per_byte": round(fl / nb, 1),
               "bound": "hbm" if fl / nb < balance else "mfma"}
        if kname in pk:   # counter bytes are per kernel SYMBOL (average over all its launches), not per shape
            row["counter_mb_per_launch_kernel_avg"] = round((pk[kname]["fetch_bytes"] + pk[kname]["write_bytes"]) / 1e6, 1)
        table.append(row)
    detail["per_shape"] = table
    return roofline, detail


def write_detail(path, line, detail):
    """per-shape / per-instantiation tables of the instrumented steps -> a file next to the run (never into the line)."""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        json.dump({"workload": line["config"]["workload"], "ms_per_step": line["ms_per_step"], "value": line["value"],
                   "env_overrides": line.get("env_overrides"), **detail}, open(path, "w"), indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return f"not written ({e})"


DTYPE_RAN = {"fp16": "fp16 forward operands + activations / bf16 backward operands + gradients, fp32 accumulate",
             "bf16": "bf16 operands + activations + gradients, fp32 accumulate"}


def dtype_ran():
    return DTYPE_RAN["bf16" if os.environ.get("PTI_FWD_ACT_DTYPE", "fp16") == "bf16" else "fp16"]


def bench_regression(args, cfg, dev, world, rank, dist):
    """config/reg_edente_from_dente.json: one step = the body of the reference's ``train_one_epoch``
    (regression_utils.py:318-347): frozen-encoder forward under no_grad (HIP engine, 17.66 GFLOP per 256x256 image) ->
    flatten -> MLP head -> MSE -> backward through the head -> Adam on the head.  N > 1: one encoder replica per GPU and a
    SUM all-reduce of the head's 1.06 M gradients (4.2 MB) per step.  Random-init encoder (no checkpoint offline)."""
    import torch.distributed as d
    from pti_ldm_vae_amd import _lib as L
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.utils import regression_utils as R
    from pti_ldm_vae_amd.utils.config import load_vae_config
    vae_cfg_path = cfg["vae"]["config_file"]
    if not os.path.isabs(vae_cfg_path):
        vae_cfg_path = os.path.join(ROOT, vae_cfg_path)
    cfg_def = load_vae_config(vae_cfg_path).autoencoder_def
    data_cfg, tr = R.extract_regression_data_config(cfg), R.extract_regression_train_config(cfg)
    size = args.size
    cfg["data"] = dict(data_cfg, patch_size=[size, size])
    targets = list(cfg["targets"])
    torch.manual_seed(42)
    vae = VAEModel.from_config(cfg_def).to(dev).eval()
    model, latent_dim = R.build_regression_model(vae, cfg, targets, dev)
    head_params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(head_params, lr=tr["lr"] * world)
    loss_fn = R.build_loss_fn(tr["loss"])
    images = synthetic_batch(args.batch, cfg_def["in_channels"], size, dev, seed=42 + rank)
    y = torch.rand(args.batch, len(targets), generator=torch.Generator(device=dev).manual_seed(7 + rank), device=dev)

    def step():
        opt.zero_grad()
        loss = loss_fn(model(images), y)
        loss.backward()
        if world > 1:
            flat = torch.cat([p.grad.flatten() for p in head_params])
            d.all_reduce(flat)
            o = 0
            for p in head_params:
                p.grad.copy_(flat[o:o + p.numel()].view_as(p) / world)
                o += p.numel()
        opt.step()
        return loss

    def sync_all():
        if world > 1:
            d.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        dt = float(t.item())
    roofline = None
    if rank == 0:
        ops.KERNEL_PROFILE = []
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        rec, ops.KERNEL_PROFILE = ops.KERNEL_PROFILE, None
        roofline, detail = roofline_from_profile(rec, 2)
        enc_gflop = 17.66 * (size / 256.0) ** 2 if tuple(cfg_def["channels"]) == (32, 64, 128, 128) else None
        value = args.batch * world * args.steps / dt
        line = {"metric": "vae_encoder_regression_images_per_sec_256x256", "value": round(value, 2), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_ran(), "data": "synthetic",
                "config": {"workload": f"config/{os.path.basename(args.config)} {size}x{size}x{cfg_def['in_channels']} batch "
                                       f"{args.batch}/GPU: frozen-encoder forward (no_grad, random-init VAE of "
                                       f"{os.path.basename(vae_cfg_path)}) + MLP head {latent_dim}->"
                                       f"{'->'.join(str(h) for h in cfg['regressor_def']['hidden_dims'])}->{len(targets)} fwd/bwd + Adam",
                           "global_batch": args.batch * world, "parallelism": f"dp{world}", "final_loss": round(float(loss.item()), 5)},
                "model_tflops_per_gpu": round(value / world * enc_gflop / 1e3, 1) if enc_gflop else None,
                "frac_of_mfma_peak_end_to_end": round(value / world * enc_gflop / 1e3 / PEAK_BF16_TFLOPS, 4) if enc_gflop else None,
                "roofline": roofline, "env_overrides": L.env_overrides()}
        line["detail_file"] = write_detail(args.detail_out, line, detail)
        print(json.dumps(line), flush=True)
    if world > 1:
        d.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--perceptual", action="store_true",
                    help="add the config's perceptual_weight * LPIPS(SqueezeNet-1.1) term (models/perceptual.py, torch ops on "
                         "the device; RANDOM-INIT weights: the pretrained files cannot be fetched -- same arithmetic and cost)")
    ap.add_argument("--adv", action="store_true",
                    help="time the step of epochs > 5 of an adv_enabled config: + PatchDiscriminator generator term and "
                         "discriminator step (train_vae.py:399-401,447-458); needs a 1-channel model")
    ap.add_argument("--channels", type=int, default=None,
                    help="override the config's image channels (in = out): 3 = north_star's 'synthetic 256x256x3' variant "
                         "(+0.3 %% FLOPs: only conv_in / conv_out change; config value is 1)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "r03_per_shape.json"),
                    help="file for the per-shape / per-instantiation tables of the instrumented steps")
    ap.add_argument("--cpu-steps", type=int, default=10,
                    help="timed oracle steps of the cpu_baseline leg (batch 4: ~10 s of CPU work on 16 cores)")
    args = ap.parse_args()

    import torch.distributed as dist
    from pti_ldm_vae_amd import _lib as L
    L.refuse_wrong_result_env("bench.py")       # a timed region must not be able to skip work by environment variable
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    from pti_ldm_vae_amd.utils import read_config
    from pti_ldm_vae_amd import ops

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if int(os.environ.get("WORLD_SIZE", "1")) != world:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N (WORLD_SIZE must equal --gpus)")
        # rehearsal knobs (single-GPU box): PTI_DIST_BACKEND=gloo PTI_SHARE_GPU=1 runs all ranks on cuda:0
        if os.environ.get("PTI_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=os.environ.get("PTI_DIST_BACKEND", "nccl"), init_method="env://")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    global PMC_WORKLOAD
    PMC_WORKLOAD = (f"{os.path.basename(args.config)}:b{args.batch}:{args.size}" + (":adv" if args.adv else "")
                    + (":perceptual" if args.perceptual else "") + (f":c{args.channels}" if args.channels else ""))
    cfg = read_config(args.config)
    if "regressor_def" in cfg or "regression_train" in cfg:      # BASELINE config 5: regression on frozen latents
        return bench_regression(args, cfg, dev, world, rank, dist)
    cfg_def = cfg["autoencoder_def"]
    if args.channels:
        cfg_def = dict(cfg_def, in_channels=args.channels, out_channels=args.channels)
    tr = cfg["autoencoder_train"]
    torch.manual_seed(42)                       # set_determinism(args.seed), train_vae.py:808
    model = VAEModel.from_config(cfg_def).to(dev)
    # AR-VAE configs (ar_vae_dente_kl1e3.json): the attribute-regularisation term is part of the step; attributes are
    # U(0,1) per sample for the mapped names
    from pti_ldm_vae_amd.trainer import ARSettings
    from pti_ldm_vae_amd.utils import resolve_ar_settings
    ar_on, ar_gamma, _, _ = resolve_ar_settings(tr, cfg.get("regularized_attributes"))
    ar = ARSettings.from_config(cfg["regularized_attributes"], ar_gamma, cfg_def["latent_channels"]) if ar_on else None
    disc = None
    if args.adv:
        from pti_ldm_vae_amd.models import PatchDiscriminator
        disc = PatchDiscriminator(spatial_dims=2, num_layers_d=3, channels=32, in_channels=1, out_channels=1,
                                  norm="INSTANCE").to(dev)
    perc = None
    if args.perceptual:
        from pti_ldm_vae_amd.models import PerceptualLoss
        perc = PerceptualLoss(spatial_dims=2, network_type="squeeze", allow_random_init=True).to(dev)
    trainer = VAETrainer(model, lr=tr["lr"], world_size=world, recon_loss=tr["recon_loss"], kl_weight=tr["kl_weight"],
                         rank_eps_offset=rank, ar=ar, discriminator=disc, adv_weight=float(tr.get("adv_weight", 0.0)),
                         perceptual=perc, perceptual_weight=float(tr.get("perceptual_weight", 1.0)) if perc is not None else 0.0)
    step_kw = {"adversarial": True} if args.adv else {}
    images = synthetic_batch(args.batch, cfg_def["in_channels"], args.size, dev, seed=42 + rank)
    attrs = None
    if ar is not None:
        ga = torch.Generator(device=dev).manual_seed(4242 + rank)
        attrs = {k: torch.rand(args.batch, generator=ga, device=dev) for k in ar.names}

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"model built, {sum(p.numel() for p in model.parameters())} params; warm-up {args.warmup} steps")
    for i in range(args.warmup):
        trainer.step(images, attributes=attrs, **step_kw)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    sync_all()
    log(f"timing {args.steps} steps")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host = 0.0
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        out = trainer.step(images, attributes=attrs, **step_kw)
        host += time.perf_counter() - h0          # time the host spends enqueuing (incl. waiting on the 2-steps-in-flight cap)
        marks[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))   # ms, GPU-side step-to-step
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    # pure host enqueue time of a step: a few steps issued with the steps-in-flight cap lifted and no sync in between
    # (the host runs ahead of the GPU; in the timed region above `host` also contains the waits on that cap)
    cap, trainer.max_steps_in_flight = trainer.max_steps_in_flight, 1 << 20
    h0 = time.perf_counter()
    for _ in range(4):
        trainer.step(images, attributes=attrs, **step_kw)
    host_only = (time.perf_counter() - h0) / 4
    sync_all()
    trainer.max_steps_in_flight = cap
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].item())
    log(f"timed region {dt:.3f}s, loss {loss:.5f}; instrumented steps")

    # ---- instrumented steps: per-launch event timing of the MFMA conv + weight-gradient kernels (rank 0) ----
    roofline, detail = None, {}
    if rank == 0:
        ops.KERNEL_PROFILE = []
    # per-kernel durations are taken with the weight gradients back on the main stream: in the timed region they
    # run on a side stream, overlapped with the data-gradient chain, which would stretch every launch they share
    # the GPU with (the headline number above is measured WITH the overlap)
    side = trainer.eng.wgrad_stream
    trainer.eng.wgrad_stream = None
    # EVERY rank steps (the step contains collectives) and every rank takes the EAGER path: only rank 0 records, and the
    # graph step mode would otherwise stay on for the other ranks, whose region-sized buckets do not match the eager
    # callbacks' buckets (a collective mismatch: gloo aborts, RCCL hangs)
    graph_mode, trainer.step_graph = trainer.step_graph, False
    for _ in range(2):
        trainer.step(images, attributes=attrs, **step_kw)
    torch.cuda.synchronize()
    trainer.step_graph = graph_mode
    trainer.eng.wgrad_stream = side
    if rank == 0:
        rec, ops.KERNEL_PROFILE = ops.KERNEL_PROFILE, None
        roofline, detail = roofline_from_profile(rec, 2)
    if world > 1:
        dist.barrier()

    if rank == 0:
        # algorithmic GFLOP per image and training step at 256x256 (BASELINE.md §3 / SURVEY.md §8d)
        gflop_img = {(32, 64, 128, 128): TRAIN_GFLOP_PER_IMG_A, (64, 128, 256): 730.6}.get(tuple(cfg_def["channels"]))
        if gflop_img and args.channels == 3 and tuple(cfg_def["channels"]) == (32, 64, 128, 128):
            gflop_img = 148.56           # SURVEY.md Appendix B, C_in = C_out = 3
        if args.size != 256:
            gflop_img = gflop_img * (args.size / 256.0) ** 2 if gflop_img else None   # convs scale with pixels (attention ~L^2 ignored)
        imgs = args.batch * world * args.steps
        value = imgs / dt
        per_gpu = value / world
        line = {
            "metric": "vae_train_images_per_sec_256x256", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_ran(), "data": "synthetic",
            "config": {"workload": f"config/{os.path.basename(args.config)} {args.size}x{args.size}x{cfg_def['in_channels']} "
                                   f"batch {args.batch}/GPU: fwd + L1 + 1e-3*KL{' + 0.5*AR-VAE(6 attributes)' if ar else ''}"
                                   + (f" + {trainer.perceptual_weight}*LPIPS(SqueezeNet-1.1, random-init weights, torch ops)" if args.perceptual else "")
                                   + (f" + {trainer.adv_weight}*LSGAN(PatchDiscriminator) + bwd + all-reduce + Adam, then the "
                                      "discriminator step (fake + real pass, bwd, all-reduce, Adam) "
                                      + ("" if args.perceptual else "(perceptual term omitted: --perceptual times it)") if args.adv else
                                      " + bwd + all-reduce + Adam (" + ("" if args.perceptual else "perceptual term omitted: "
                                      "--perceptual times it; ") + "adversarial branch inactive before epoch 6, --adv times it)"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "final_loss": round(loss, 5)},
            "model_tflops_per_gpu": round(per_gpu * gflop_img / 1e3, 1) if gflop_img else None,
            "frac_of_mfma_peak_end_to_end": round(per_gpu * gflop_img / 1e3 / PEAK_BF16_TFLOPS, 4) if gflop_img else None,
            "step_ms_median": round(pct(0.5), 3), "step_ms_p10": round(pct(0.1), 3), "step_ms_p90": round(pct(0.9), 3),
            "host_enqueue_ms_per_step": round(host_only * 1e3, 3),
            "host_in_step_call_ms_per_step": round(host / args.steps * 1e3, 3),
            "roofline": roofline, "env_overrides": L.env_overrides(),
        }
        pmc = _pmc()
        if pmc and "bytes_per_step" in pmc:   # HBM-side bytes of a whole step (all kernels) from the committed counter passes
            line["end_to_end_hbm_bytes_per_step"] = pmc["bytes_per_step"]
            line["end_to_end_hbm_frac"] = round(pmc["bytes_per_step"] / (dt / args.steps) / (PEAK_HBM_GBS * 1e9), 4)
            line["end_to_end_hbm_source"] = f"profiles/{PMC_TRAFFIC_FILE}"
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle) ...")
            line["cpu_baseline"] = cpu_baseline(cfg_def, args.size, 4, args.cpu_steps)
        line["detail_file"] = write_detail(args.detail_out, line, detail)
        out_line = json.dumps(line)
        if len(out_line) >= 4096:      # the driver keeps ~8 KB of stdout: never let the line outgrow it again
            line["roofline"].pop("secondary", None)
            out_line = json.dumps(line)
        print(out_line, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
