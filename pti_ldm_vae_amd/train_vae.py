#!/usr/bin/env python3
"""VAE training driver for MI355X — the counterpart of the reference's ``vae_scripts/train_vae.py``.

Same CLI (``-c/--config-file -g/--gpus --batch-size --lr --max-epochs --num-workers --cache-rate --seed
--subset-size``), same JSON config keys, same run-directory layout and checkpoint
files/keys (``trained_weights/autoencoder_last.pt``, ``autoencoder_epoch{E}.pth``, ``checkpoint_epoch{E}.pth``
with ``epoch / autoencoder_state_dict / discriminator_state_dict / optimizer_g_state_dict /
optimizer_d_state_dict / best_val_loss / total_step`` — train_vae.py:675-769), same data-parallel contract
(one process per GPU under ``torchrun``, ``env://`` rendezvous, lr x world size, rank-0 checkpoints,
rank-local validation means, best checkpoint chosen on rank 0's recon loss).

What is different, on purpose:
  * the step runs on the HIP engine through ``VAETrainer`` (no autograd tape, no DDP wrapper, no
    unused-parameter search, no ``detect_anomaly``; ``--detect-anomaly`` is not needed without a tape);
  * the adversarial branch (``adv_enabled``: PatchDiscriminator + least-squares PatchAdversarialLoss, active from epoch
    6 on, train_vae.py:266-279,399-401,447-458) runs natively (``models/patch_discriminator.py``, ``disc_engine.py``);
    ``discriminator_last.pt`` / ``discriminator_epoch{E}.pth`` and the ``discriminator_state_dict`` /
    ``optimizer_d_state_dict`` checkpoint entries are written and read like the reference's;
  * the perceptual (LPIPS) term needs pretrained weights that cannot be fetched here: pass
    ``--perceptual-weights squeezenet1_1.pth lpips_squeeze.pth`` (local files) and it is part of the step
    (``models/perceptual.py``: torch ops on the device, gradient added to the native backward); without them a
    non-zero ``perceptual_weight`` is refused — or pass ``--ignore-unavailable-terms`` to train without it and a warning;
  * the AR-VAE term (``regularized_attributes``) IS part of the native step (``pti_ar_vae_loss``);
  * data: without ``--synthetic`` the TIFF directory of the config is read through the device input pipeline
    (``pti_ldm_vae_amd.data``: host decode -> one H2D copy -> GPU resize + masked z-score; attribute JSONs joined
    for AR-VAE); ``--synthetic N`` trains on N seeded synthetic images of the configured patch size (z-scored
    elliptical foreground, zero background; U(0,1) attributes), sharded across ranks like ``DistributedSampler``;
  * logging goes to ``<run_dir>/metrics.jsonl`` with the reference's W&B metric names
    (``train/recon_loss`` ... ``val/loss_total``), one host sync per ``--log-every`` steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

from .models import PatchDiscriminator, PerceptualLoss, VAEModel, compute_total_loss
from .trainer import ARSettings, VAETrainer, prepare_batch
from .utils import read_config, resolve_ar_settings
from .utils.distributed import setup_ddp


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="VAE training on MI355X (HIP engine)")
    p.add_argument("-c", "--config-file", default="./config/vae_dente_recon_kl.json",
                   help="default: the reference's vae_dente_no_adv.json with perceptual_weight 0 (the LPIPS term the HIP path "
                        "does not provide); the reference configs themselves need --ignore-unavailable-terms")
    p.add_argument("-g", "--gpus", default=1, type=int)
    p.add_argument("--batch-size", type=int)
    p.add_argument("--lr", type=float)
    p.add_argument("--max-epochs", type=int)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--cache-rate", type=float, default=0.0)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--subset-size", type=int)
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images (no file I/O)")
    p.add_argument("--log-every", type=int, default=20)
    p.add_argument("--ignore-unavailable-terms", action="store_true")
    p.add_argument("--perceptual-weights", nargs=2, metavar=("SQUEEZENET_PTH", "LPIPS_SQUEEZE_PTH"), default=None,
                   help="local weight files of the perceptual loss: torchvision squeezenet1_1 state_dict and lpips v0.1 squeeze.pth")
    p.add_argument("--adv-start-epoch", type=int, default=6,
                   help="first epoch with the adversarial branch on (the reference hard-codes `epoch > 5`)")
    p.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    return p.parse_args(argv)


def setup_environment(args):
    """train_vae.py:72-97 without the CUDA-isms that do not apply (cudnn.benchmark, detect_anomaly)."""
    ddp = args.gpus > 1
    if ddp:
        rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
        dist, device = setup_ddp(rank, world, backend=args.backend)
    else:
        rank, world, dist = 0, 1, None
        device = torch.device("cuda:0")
    if not torch.cuda.is_available():
        raise RuntimeError("train_vae: the HIP engine needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(device if device.type == "cuda" else 0)
    torch.set_num_threads(4)
    return ddp, rank, world, torch.device(f"cuda:{rank}" if ddp else "cuda:0"), dist


def load_config(args):
    """train_vae.py:100-124: config keys become attributes; CLI overrides batch_size / max_epochs / lr."""
    for k, v in read_config(args.config_file).items():
        setattr(args, k, v)
    if args.batch_size:
        args.autoencoder_train["batch_size"] = args.batch_size
    if args.max_epochs:
        args.autoencoder_train["max_epochs"] = args.max_epochs
    if args.lr:
        args.autoencoder_train["lr"] = args.lr
    args.model_dir = os.path.join(args.run_dir, "trained_weights")
    return args


class SyntheticShards:
    """N seeded synthetic images, split train/val by ``train_split`` and sharded over ranks the way
    ``DistributedSampler(shuffle=True, seed)`` does (every world-th index of a seed+epoch permutation,
    wrapped to equal length).  Images are generated on the device, batch by batch."""

    def __init__(self, n, channels, size, batch, rank, world, seed, device, train_split=0.9, attr_names=None):
        self.attr_names = attr_names
        self.n_train = max(1, int(n * train_split))
        self.n_val = max(1, n - self.n_train)
        self.c, self.size, self.batch, self.rank, self.world, self.seed, self.dev = channels, size, batch, rank, world, seed, device
        lin = torch.linspace(-1, 1, size[0], device=device)[:, None], torch.linspace(-1, 1, size[1], device=device)[None, :]
        self.mask = ((lin[1] / 0.80) ** 2 + (lin[0] / 0.64) ** 2 <= 1.0).float()

    def _image(self, idx):
        g = torch.Generator(device=self.dev).manual_seed(self.seed * 1_000_003 + int(idx))
        return torch.randn(self.c, *self.size, generator=g, device=self.dev) * self.mask

    def _indices(self, n, epoch, offset):
        g = torch.Generator().manual_seed(self.seed + epoch)
        perm = torch.randperm(n, generator=g).tolist()
        total = -(-n // self.world) * self.world
        perm += perm[: total - n]
        return [offset + i for i in perm[self.rank:total:self.world]]

    def batches(self, epoch, train=True):
        idx = self._indices(self.n_train, epoch, 0) if train else self._indices(self.n_val, 0, self.n_train)
        for i in range(0, len(idx), self.batch):
            ids = idx[i:i + self.batch]
            images = torch.stack([self._image(j) for j in ids])
            if self.attr_names is None:
                yield images
            else:   # U(0,1) attribute values per synthetic sample, fixed by (seed, sample, name) -- SURVEY.md 8(d)
                yield images, {k: torch.tensor([self._attr(j, q) for j in ids], dtype=torch.float32)
                               for q, k in enumerate(self.attr_names)}

    def _attr(self, idx, q):
        g = torch.Generator().manual_seed(self.seed * 7_919 + int(idx) * 131 + q)
        return float(torch.rand((), generator=g))


class TiffShards:
    """The same ``batches(epoch, train)`` interface over a directory of .tif images (pti_ldm_vae_amd.data)."""

    def __init__(self, base_dir, batch, patch, rank, world, seed, device, *, data_source, train_split, subset_size, val_dir,
                 num_workers, ar_vae_enabled=False, regularized_attributes=None):
        from .data import create_vae_dataloaders
        self.train, self.val, self.train_paths, self.val_paths = create_vae_dataloaders(
            base_dir, batch, patch, rank=rank, data_source=data_source, train_split=train_split, num_workers=num_workers,
            seed=seed, subset_size=subset_size, val_dir=val_dir, distributed=world > 1, world_size=world, device=device,
            ar_vae_enabled=ar_vae_enabled, regularized_attributes=regularized_attributes)

    def batches(self, epoch, train=True):
        loader = self.train if train else self.val
        loader.set_epoch(epoch if train else 0)
        yield from loader


def save_checkpoints(args, model, opt, epoch, val_loss, best_val_loss, best_epoch_saved, total_step, rank, disc=None,
                     opt_d=None):
    """train_vae.py:675-769: always ``autoencoder_last.pt``; on improvement replace the best files."""
    if rank != 0:
        return best_val_loss, best_epoch_saved
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(args.model_dir, "autoencoder_last.pt"))
    dsd = None
    if disc is not None:
        dsd = {k: v.detach().cpu() for k, v in disc.state_dict().items()}
        torch.save(dsd, os.path.join(args.model_dir, "discriminator_last.pt"))
    if val_loss >= best_val_loss:
        return best_val_loss, best_epoch_saved
    if best_epoch_saved is not None:
        for f in (f"checkpoint_epoch{best_epoch_saved}.pth", f"autoencoder_epoch{best_epoch_saved}.pth"):
            f = os.path.join(args.model_dir, f)
            if os.path.exists(f):
                os.remove(f)
    torch.save(sd, os.path.join(args.model_dir, f"autoencoder_epoch{epoch}.pth"))
    if dsd is not None:
        torch.save(dsd, os.path.join(args.model_dir, f"discriminator_epoch{epoch}.pth"))

    def host(osd):
        for st in osd["state"].values():
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
        return osd
    osd = host(opt.state_dict())
    torch.save({"epoch": epoch, "autoencoder_state_dict": sd, "discriminator_state_dict": dsd,
                "optimizer_g_state_dict": osd, "optimizer_d_state_dict": host(opt_d.state_dict()) if opt_d is not None else None,
                "best_val_loss": val_loss,
                "total_step": total_step}, os.path.join(args.model_dir, f"checkpoint_epoch{epoch}.pth"))
    print(f"Best models saved for epoch {epoch}")
    return val_loss, epoch


def load_checkpoint(args, model, opt, device, disc=None, opt_d=None):
    """train_vae.py:309-339 (with map_location fixed, Appendix D); ``checkpoint_dir`` is a FILE path."""
    if not args.resume_ckpt:
        print("[INFO] Training from scratch")
        return 0, 100.0, 0, None
    path = args.checkpoint_dir
    if not os.path.exists(path):
        raise FileNotFoundError(f"[ERROR] Checkpoint not found: {path}")
    ck = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ck["autoencoder_state_dict"])
    opt.load_state_dict(ck["optimizer_g_state_dict"])
    if disc is not None and ck.get("discriminator_state_dict") is not None:     # train_vae.py:320-331
        disc.load_state_dict(ck["discriminator_state_dict"])
        if opt_d is not None and ck.get("optimizer_d_state_dict") is not None:
            opt_d.load_state_dict(ck["optimizer_d_state_dict"])
    print(f"[INFO] Resuming from epoch {ck['epoch'] + 1} | best_val_loss = {ck['best_val_loss']:.4f}")
    return ck["epoch"] + 1, ck["best_val_loss"], ck["total_step"], ck["epoch"]


def main(argv=None):
    from . import _lib
    _lib.refuse_wrong_result_env("train_vae.py")    # timing diagnostics that drop launches never reach a training run
    args = parse_args(argv)
    ddp, rank, world, device, dist = setup_environment(args)
    args = load_config(args)
    tr = args.autoencoder_train
    ar_enabled, ar_gamma, _, _ = resolve_ar_settings(tr, getattr(args, "regularized_attributes", {}))
    adv_enabled = bool(tr.get("adv_enabled", True))
    unavailable = []
    perceptual_weight = float(tr.get("perceptual_weight", 0.0))
    perceptual = None
    if perceptual_weight != 0.0 and args.perceptual_weights:
        perceptual = PerceptualLoss(spatial_dims=2, network_type="squeeze", weights=tuple(args.perceptual_weights)).to(device)
    elif perceptual_weight != 0.0:
        unavailable.append(f"perceptual_weight={tr['perceptual_weight']} (the pretrained SqueezeNet/LPIPS weights are not "
                           "available offline: supply them with --perceptual-weights)")
        perceptual_weight = 0.0
    if unavailable:
        msg = "terms not available in the native trainer: " + "; ".join(unavailable)
        if not args.ignore_unavailable_terms:
            raise SystemExit(msg + " — set them to 0/false or pass --ignore-unavailable-terms")
        if rank == 0:
            print("[WARN] " + msg + " — training without them")
    if rank == 0:
        run_dir = Path(args.run_dir)
        if run_dir.exists() and not args.resume_ckpt:
            raise ValueError(f"Run directory already exists: {run_dir}\nTo prevent overwriting previous runs:\n"
                             "  1. Change 'run_dir' in your config file, or\n  2. Set 'resume_ckpt: true' to continue training")
        Path(args.model_dir).mkdir(parents=True, exist_ok=True)
        (run_dir / "splits").mkdir(parents=True, exist_ok=True)
    torch.manual_seed(args.seed)                    # set_determinism(args.seed)
    model = VAEModel.from_config(args.autoencoder_def).to(device)
    if rank == 0:
        print("\n=== Autoencoder model summary ===\n", model, "\n=================================\n")
    pg = None
    reg_attrs = getattr(args, "regularized_attributes", {}) or {}
    ar = ARSettings.from_config(reg_attrs, ar_gamma, args.autoencoder_def["latent_channels"]) if ar_enabled else None
    # create_models (train_vae.py:266-279): the discriminator exists whenever adv_enabled; it is used from epoch 6 on
    disc = PatchDiscriminator(spatial_dims=2, num_layers_d=3, channels=32, in_channels=1, out_channels=1,
                              norm="INSTANCE").to(device) if adv_enabled else None
    adv_weight = float(tr.get("adv_weight", 0.0))
    trainer = VAETrainer(model, lr=tr["lr"], world_size=world, process_group=pg, recon_loss=tr.get("recon_loss", "l1"),
                         kl_weight=tr["kl_weight"], rank_eps_offset=rank, ar=ar, discriminator=disc, adv_weight=adv_weight,
                         perceptual=perceptual, perceptual_weight=perceptual_weight)
    opt_d = trainer.opt_d if disc is not None else None
    start_epoch, best_val, total_step, best_epoch_saved = load_checkpoint(args, model, trainer.opt, device, disc, opt_d)
    model.autoencoder.mark_weights_dirty()
    if args.synthetic:
        n = args.subset_size or args.synthetic
        data = SyntheticShards(n, args.autoencoder_def["in_channels"], tuple(tr["patch_size"]), tr["batch_size"], rank, world,
                               args.seed, device, args.train_split, attr_names=ar.names if ar else None)
        train_files = [f"synthetic:{i}" for i in range(data.n_train)]
        val_files = [f"synthetic:{i}" for i in range(data.n_train, data.n_train + data.n_val)]
    else:
        # TIFF directory -> device batches (create_vae_dataloaders, train_vae.py:832-850 of the reference)
        if args.autoencoder_def["in_channels"] != 1:
            raise SystemExit("train_vae: the TIFF pipeline produces single-channel images (in_channels must be 1)")
        data = TiffShards(args.data_base_dir, tr["batch_size"], tuple(tr["patch_size"]), rank, world, args.seed, device,
                          data_source=getattr(args, "data_source", "edente"), train_split=args.train_split,
                          subset_size=args.subset_size, val_dir=getattr(args, "val_dir", None), num_workers=args.num_workers,
                          ar_vae_enabled=ar_enabled, regularized_attributes=reg_attrs)
        train_files, val_files = data.train_paths, data.val_paths
    if rank == 0:
        with open(Path(args.run_dir) / "splits" / "vae_split.json", "w", encoding="utf-8") as f:
            json.dump({"seed": args.seed, "train_split": args.train_split, "subset_size": args.subset_size,
                       "val_dir": args.val_dir, "train_files": train_files, "val_files": val_files}, f, indent=2)
    log = open(Path(args.run_dir) / "metrics.jsonl", "a") if rank == 0 else None
    kl_w, max_epochs, val_interval = tr["kl_weight"], tr["max_epochs"], tr["val_interval"]
    for epoch in range(start_epoch, max_epochs):
        t0 = time.time()
        seen = 0
        adv_on = disc is not None and epoch >= args.adv_start_epoch      # reference: adv_enabled and epoch > 5
        for step, batch in enumerate(data.batches(epoch, train=True)):
            images, attrs = prepare_batch(batch, device, ar_enabled)
            out = trainer.step(images, attributes=attrs, adversarial=adv_on)
            total_step += 1
            seen += images.shape[0]
            if log is not None and step % args.log_every == 0:
                rec = {"train/step": total_step, "train/recon_loss": out["recon"].item(),
                       "train/kl_loss": out["kl"].item(), "train/loss_total": out["loss"].item(),
                       "train/perceptual_loss": out["perceptual"].item() if perceptual is not None else 0.0,
                       # W&B names and weighting of train_vae.py:467-468
                       "train/adv_gen_loss": adv_weight * out["adv_gen"].item() if adv_on else 0.0,
                       "train/adv_disc_loss": adv_weight * out["adv_disc"].item() if adv_on else 0.0}
                if ar is not None:   # W&B names of train_vae.py:471-478
                    rec["train/ar_loss_total"] = out["ar"].item()
                    for name, la, cnt, dl in zip(ar.names, out["ar_per_attr"].tolist(), out["ar_pairs"].tolist(), ar.deltas):
                        rec[f"train/ar_loss_{name}"], rec[f"train/ar_pairs_{name}"], rec[f"train/ar_delta_{name}"] = la, cnt, dl
                log.write(json.dumps(rec) + "\n")
                log.flush()
        if epoch % val_interval == 0:
            rsum = ksum = asum = gsum = psum = torch.zeros((), device=device)
            nb = 0
            for batch in data.batches(epoch, train=False):
                images, attrs = prepare_batch(batch, device, ar_enabled)
                v, _ = trainer.eval_losses(images, attributes=attrs, adversarial=adv_on)
                rsum, ksum, nb = rsum + v["recon"], ksum + v["kl"], nb + 1
                if adv_on:
                    gsum = gsum + v["adv_gen"]
                if perceptual is not None:
                    psum = psum + v["perceptual"]
                if ar is not None:
                    asum = asum + v["ar"]
            val_recon, val_kl, val_ar, val_gen, val_p = ((t / max(nb, 1)).item() for t in (rsum, ksum, asum, gsum, psum))
            val_total = compute_total_loss(val_recon, val_kl, val_p, val_gen, val_ar, kl_weight=kl_w,
                                           perceptual_weight=perceptual_weight,
                                           adv_weight=adv_weight if adv_on else 0.0, ar_gamma=ar_gamma,
                                           ar_vae_enabled=ar_enabled)
            torch.cuda.synchronize()
            dt = time.time() - t0
            if rank == 0:
                print(f"Epoch {epoch} val_loss: {val_recon:.4f} | Time: {dt:.1f}s | {seen * world / dt:.1f} img/s")
                rec = {"epoch": epoch, "val/recon_loss": val_recon, "val/kl_loss": val_kl, "val/loss_total": val_total,
                       "val/perceptual_loss": val_p, "time_per_epoch": dt}
                if ar is not None:
                    rec["val/ar_loss_total"] = val_ar
                log.write(json.dumps(rec) + "\n")
                log.flush()
            best_val, best_epoch_saved = save_checkpoints(args, model, trainer.opt, epoch, val_recon, best_val,
                                                          best_epoch_saved, total_step, rank, disc, opt_d)
    if log is not None:
        log.close()
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
