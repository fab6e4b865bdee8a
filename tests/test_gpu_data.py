"""GPU parity of the input pipeline: pti_preprocess_batch against the CPU restatement of
Resize(area) + LocalNormalizeByMask (oracle/data_pipeline.py), the device loader end to end over a directory of
TIFF files (order, sharding, last short batch), and train_vae.py on such a directory.

Tolerance: fp32 averages and a z-score whose statistics numpy sums in fp32 and the kernel in fp64:
max-abs <= 2e-5 on outputs of unit scale; exact zeros must stay exact zeros."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(rng, shapes):
    out = []
    for h, w in shapes:
        a = rng.standard_normal((h, w)).astype(np.float32) * 300 + 900
        yy, xx = np.mgrid[0:h, 0:w]
        a[((xx - w / 2) / (0.4 * w)) ** 2 + ((yy - h / 2) / (0.32 * h)) ** 2 > 1.0] = 0.0     # exact-zero background
        out.append(a)
    return out


@pytest.mark.parametrize("patch", [(64, 64), (48, 80), (256, 256)])
def test_preprocess_batch_matches_oracle(dev, patch):
    from oracle.data_pipeline import preprocess
    from pti_ldm_vae_amd import ops
    rng = np.random.default_rng(5)
    shapes = [(300, 300), (257, 191), (64, 64), (40, 100), (512, 384)]
    imgs = _images(rng, shapes) + [np.zeros((70, 90), np.float32), np.full((33, 33), 7.0, np.float32)]
    flat = np.concatenate([a.reshape(-1) for a in imgs])
    offs = np.cumsum([0] + [a.size for a in imgs[:-1]]).astype(np.int64)
    hw = np.array([a.shape for a in imgs], np.int32)
    out = torch.full((len(imgs), 1, *patch), float("nan"), device=dev)
    ops.preprocess_batch(torch.from_numpy(flat).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(hw).to(dev), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, a in enumerate(imgs):
        ref = preprocess(a, patch)
        err = np.abs(got[i] - ref).max()
        print(f"image {i} {a.shape} -> {patch}: max|err| {err:.2e}, zeros kept {np.array_equal(got[i] == 0, ref == 0)}")
        assert err <= 2e-5
        assert np.array_equal(got[i] == 0, ref == 0)


def _write_dir(tmp_path, n, rng):
    from pti_ldm_vae_amd.data import write_tiff
    d = tmp_path / "data" / "dente"
    d.mkdir(parents=True)
    shapes = [(96 + 8 * (i % 5), 120 - 4 * (i % 7)) for i in range(n)]
    imgs = _images(rng, shapes)
    for i, a in enumerate(imgs):
        write_tiff(str(d / f"img_{i:03d}.tif"), a, rows_per_strip=16 if i % 2 else None)
    return str(tmp_path / "data"), imgs


def test_device_loader_order_and_values(dev, tmp_path):
    from oracle.data_pipeline import preprocess
    from pti_ldm_vae_amd.data import DeviceImageLoader, list_tif_paths, shard_indices
    rng = np.random.default_rng(6)
    base, imgs = _write_dir(tmp_path, 11, rng)
    paths = list_tif_paths(base, "dente")
    assert len(paths) == 11
    for world, rank in ((1, 0), (2, 1)):
        ld = DeviceImageLoader(paths, 4, (64, 64), dev, rank=rank, world_size=world, shuffle=True, seed=42, num_workers=3)
        ld.set_epoch(2)
        want = shard_indices(11, rank, world, True, 42, 2)
        got = [b.cpu().numpy() for b in ld]
        assert len(got) == len(ld) == -(-len(want) // 4)
        assert sum(g.shape[0] for g in got) == len(want)
        flat = np.concatenate(got)
        for k, idx in enumerate(want):
            assert np.abs(flat[k] - preprocess(imgs[idx], (64, 64))).max() <= 2e-5, (world, rank, k, idx)


def test_train_script_on_tiff_directory(dev, tmp_path):
    from pti_ldm_vae_amd import train_vae
    rng = np.random.default_rng(7)
    base, _ = _write_dir(tmp_path, 12, rng)
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "vae_dente_no_adv.json")))
    cfg.update(run_dir=str(tmp_path / "run"), data_base_dir=base, data_source="dente")
    cfg["autoencoder_def"].update(channels=[32, 64], attention_levels=[False, False], num_res_blocks=1)
    cfg["autoencoder_train"].update(batch_size=4, patch_size=[64, 64], max_epochs=2, perceptual_weight=0.0)
    cf = tmp_path / "cfg.json"
    cf.write_text(json.dumps(cfg))
    train_vae.main(["-c", str(cf), "--log-every", "1", "--num-workers", "2"])
    split = json.load(open(tmp_path / "run" / "splits" / "vae_split.json"))
    assert len(split["train_files"]) == int(0.9 * 12) or len(split["train_files"]) == int(cfg.get("train_split", 0.9) * 12)
    assert all(f.endswith(".tif") for f in split["train_files"] + split["val_files"])
    lines = [json.loads(l) for l in open(tmp_path / "run" / "metrics.jsonl")]
    tl = [l["train/loss_total"] for l in lines if "train/loss_total" in l]
    assert len(tl) >= 4 and all(np.isfinite(tl)) and any("val/recon_loss" in l for l in lines)
    assert "autoencoder_last.pt" in os.listdir(tmp_path / "run" / "trained_weights")


def test_train_script_ar_vae_on_tiff_directory_with_attribute_file(dev, tmp_path):
    """The AR-VAE config path end to end: attribute JSON joined to the TIFF list, (images, attributes) batches from
    the device loader, prepare_batch, the AR term inside the native step, AR metrics in the log."""
    from pti_ldm_vae_amd import train_vae
    from pti_ldm_vae_amd.data import create_vae_dataloaders
    rng = np.random.default_rng(8)
    base, _ = _write_dir(tmp_path, 12, rng)
    names = sorted(os.listdir(os.path.join(base, "dente")))
    table = {f: {"height_0": float(rng.random()), "width_0": float(rng.random()), "extra": 1.0} for f in names}
    af = tmp_path / "attrs.json"
    af.write_text(json.dumps(table))
    ra = {"enabled": True, "attribute_file": str(af), "gamma": 0.5, "pairwise": "all",
          "delta_global": {"enabled": True, "value": 2.0}, "normalize_attributes": {"enabled": False},
          "attribute_latent_mapping": {"height_0": {"latent_channel": 0}, "width_0": {"latent_channel": 2, "delta": 1.0}}}
    tr_l, va_l, tr_p, va_p = create_vae_dataloaders(base, 4, (64, 64), data_source="dente", num_workers=2, device=dev,
                                                    ar_vae_enabled=True, regularized_attributes=ra)
    tr_l.set_epoch(0)
    images, attrs = next(iter(tr_l))
    assert images.shape == (4, 1, 64, 64) and set(attrs) == {"height_0", "width_0"}
    assert attrs["height_0"].is_cuda and attrs["height_0"].dtype == torch.float32 and attrs["height_0"].shape == (4,)
    from pti_ldm_vae_amd.data import shard_indices
    first = shard_indices(len(tr_p), 0, 1, True, 42, 0)[:4]
    want = [table[os.path.basename(tr_p[i])]["height_0"] for i in first]
    assert attrs["height_0"].cpu().tolist() == pytest.approx(want)
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "ar_vae_dente_kl1e3.json")))
    cfg.update(run_dir=str(tmp_path / "run"), data_base_dir=base, data_source="dente", regularized_attributes=ra)
    cfg["autoencoder_def"].update(channels=[32, 64], attention_levels=[False, False], num_res_blocks=1, norm_num_groups=16,
                                  latent_channels=4)
    cfg["autoencoder_train"].update(batch_size=4, patch_size=[64, 64], max_epochs=2, perceptual_weight=0.0, adv_enabled=False,
                                    ar_vae_enabled=True, ar_vae_weight=0.5)
    cf = tmp_path / "cfg.json"
    cf.write_text(json.dumps(cfg))
    train_vae.main(["-c", str(cf), "--log-every", "1", "--num-workers", "2"])
    lines = [json.loads(l) for l in open(tmp_path / "run" / "metrics.jsonl")]
    tl = [l for l in lines if "train/ar_loss_total" in l]
    assert len(tl) >= 4 and all(np.isfinite(l["train/ar_loss_total"]) and l["train/ar_loss_total"] > 0 for l in tl)
    assert tl[0]["train/ar_pairs_height_0"] == 12 and tl[0]["train/ar_delta_height_0"] == 2.0 and tl[0]["train/ar_delta_width_0"] == 1.0
    assert any("val/ar_loss_total" in l for l in lines)
