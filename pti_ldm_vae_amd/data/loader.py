"""TIFF directory -> device-resident training batches.

Mirrors what ``create_vae_dataloaders`` of the reference sets up (``src/pti_ldm_vae/data/dataloaders.py:370-593``):
the path listing (``:15-47``), the seeded shuffle + ``train_split`` cut (``:469-513``), the per-sample transform
chain ``LoadImage -> EnsureChannelFirst -> Resize(patch_size) -> LocalNormalizeByMask -> float32`` (``:319-329``) and
``DistributedSampler`` sharding (``:545-550``) -- but the transform chain runs on the GPU:

  host threads decode TIFFs into one pinned staging buffer  ->  ONE H2D copy per batch on a copy stream  ->
  ``pti_preprocess_batch`` (area resize + masked z-score, two launches) on that stream  ->  an event the training
  stream waits on.  Two batches are in flight (double buffering), so decode + copy + preprocessing of batch k+1
  overlap the optimiser step of batch k.

AR-VAE: with ``ar_vae_enabled`` the per-image attributes (``data/attributes.py``) ride along and a batch is the pair
``(images, {name: float32 [b] device tensor})`` -- what ``collate_with_attributes`` (dataloaders.py:108-117) yields, already
on the device.  Not mirrored (stated, not silently dropped): ``cache_rate`` (the whole decoded set is small enough to
keep in page cache), MONAI meta-tensors.  Axis order: the TIFF is taken row-major as (H, W);
which reader MONAI's ``LoadImage`` picks for ``.tif`` (and whether it transposes) cannot be checked offline -- with
the square patches of every reference config this only mirrors the image, it does not change the statistics.
"""
from __future__ import annotations

import random
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from .tiff import read_tiff


def list_tif_paths(data_base_dir: str, data_source: str = "edente") -> list[str]:
    """``_list_tif_paths`` (dataloaders.py:15-47): ``*.tif`` directly in the directory, else in its ``edente`` /
    ``dente`` sub-folders (``both`` = edente then dente), sorted."""
    base = Path(data_base_dir)
    direct = sorted(base.glob("*.tif"))
    if direct:
        return [str(p) for p in direct]
    if data_source == "edente":
        paths = sorted((base / "edente").glob("*.tif"))
    elif data_source == "dente":
        paths = sorted((base / "dente").glob("*.tif"))
    elif data_source == "both":
        paths = sorted((base / "edente").glob("*.tif")) + sorted((base / "dente").glob("*.tif"))
    else:
        raise ValueError(f"data_source must be 'edente', 'dente', or 'both', got '{data_source}'")
    if not paths:
        raise FileNotFoundError(f"no .tif image found in {data_base_dir}/{data_source}")
    return [str(p) for p in paths]


def split_paths(paths: list[str], train_split: float = 0.9, seed: int | None = 42, subset_size: int | None = None,
                val_paths: list[str] | None = None) -> tuple[list[str], list[str]]:
    """Seeded shuffle and cut (dataloaders.py:469-513): ``random.seed(seed); random.shuffle(copy)``, first
    ``int(train_split * n)`` paths train, the rest validate; an external validation list keeps every path for
    training.  ``subset_size`` truncates the sorted list first, as the reference does before shuffling."""
    paths = list(paths)
    if subset_size is not None:
        paths = paths[:subset_size]
    if seed is not None:
        random.seed(seed)
        random.shuffle(paths)
    if val_paths is not None:
        return paths, list(val_paths)
    cut = int(train_split * len(paths))
    return paths[:cut], paths[cut:]


def shard_indices(n: int, rank: int, world: int, shuffle: bool, seed: int, epoch: int) -> list[int]:
    """Indices ``torch.utils.data.DistributedSampler(num_replicas=world, rank, shuffle, seed)`` yields after
    ``set_epoch(epoch)`` with ``drop_last=False``: a ``seed + epoch`` permutation (or ``range``), padded by wrapping
    around to a multiple of ``world``, every ``world``-th element starting at ``rank``."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad:
        idx += (idx * (pad // len(idx) + 1))[:pad]
    return idx[rank:total:world]


class DeviceImageLoader:
    """Iterable over device batches ``[b, 1, Hp, Wp]`` fp32 of the preprocessed images at ``paths``.

    ``set_epoch(e)`` selects the permutation (as ``DistributedSampler.set_epoch``); iteration order, padding and the
    last (short) batch follow ``DataLoader(batch_size, sampler=DistributedSampler(...))``.  Each yielded tensor is
    safe to use on the current stream; it stays valid until two further batches have been requested."""

    def __init__(self, paths: list[str], batch_size: int, patch_size: tuple[int, int], device, *, rank: int = 0,
                 world_size: int = 1, shuffle: bool = True, seed: int = 42, num_workers: int = 4,
                 attributes: list[dict[str, float]] | None = None, target_names: list[str] | None = None):
        """``attributes``: one {name: value} dict per path; batches become ``(images, {name: [b] tensor})``.  With
        ``target_names`` the values are instead stacked in that order: ``(images, [b, T] tensor)`` (regression)."""
        from .. import ops
        self._ops = ops
        self.paths, self.batch, self.patch = list(paths), int(batch_size), (int(patch_size[0]), int(patch_size[1]))
        if attributes is not None and len(attributes) != len(self.paths):
            raise ValueError("DeviceImageLoader: one attribute dict per image path is required")
        self.attributes, self.target_names = attributes, target_names
        self.dev = torch.device(device)
        self.rank, self.world, self.shuffle, self.seed, self.epoch = rank, world_size, shuffle, seed, 0
        self.pool = ThreadPoolExecutor(max_workers=max(1, num_workers))
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self._slots = [dict(pinned=None, dev=None, out=None, stats=None, event=None) for _ in range(3)]

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        n = len(shard_indices(len(self.paths), self.rank, self.world, False, 0, 0))
        return -(-n // self.batch)

    @staticmethod
    def _decode(path: str) -> np.ndarray:
        img = read_tiff(path)
        return np.ascontiguousarray(img, dtype=np.float32)

    def _stage(self, slot: dict, idx: list[int]):
        imgs = list(self.pool.map(self._decode, [self.paths[i] for i in idx]))
        sizes = [im.size for im in imgs]
        total = sum(sizes)
        if slot["pinned"] is None or slot["pinned"].numel() < total:
            cap = max(total, int(1.25 * total))
            slot["pinned"] = torch.empty(cap, dtype=torch.float32).pin_memory()
            slot["dev"] = torch.empty(cap, dtype=torch.float32, device=self.dev)
        host = slot["pinned"].numpy()
        offs, o = [], 0
        for im, s in zip(imgs, sizes):
            host[o:o + s] = im.reshape(-1)
            offs.append(o)
            o += s
        b = len(imgs)
        desc = torch.tensor(offs, dtype=torch.int64).pin_memory()
        hw = torch.tensor([[im.shape[0], im.shape[1]] for im in imgs], dtype=torch.int32).pin_memory()
        with torch.cuda.stream(self.copy_stream):
            # allocated in the copy stream's pool; the consumer stream is recorded on it when it is yielded
            out = torch.empty(b, 1, *self.patch, dtype=torch.float32, device=self.dev)
            slot["dev"][:total].copy_(slot["pinned"][:total], non_blocking=True)
            d_off = desc.to(self.dev, non_blocking=True)
            d_hw = hw.to(self.dev, non_blocking=True)
            self._ops.preprocess_batch(slot["dev"], d_off, d_hw, out)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        slot["event"], slot["out"], slot["keep"] = ev, out, (desc, hw, d_off, d_hw)
        slot["attrs"] = None
        if self.attributes is not None:
            names = self.target_names or list(self.attributes[idx[0]].keys())
            host = torch.tensor([[float(self.attributes[i][k]) for i in idx] for k in names], dtype=torch.float32).pin_memory()
            with torch.cuda.stream(self.copy_stream):
                dev_attrs = host.to(self.dev, non_blocking=True)
                ev.record(self.copy_stream)
            slot["keep"] += (host,)
            slot["attrs"] = (names, dev_attrs)
        return slot

    def __iter__(self):
        idx = shard_indices(len(self.paths), self.rank, self.world, self.shuffle, self.seed, self.epoch)
        batches = [idx[i:i + self.batch] for i in range(0, len(idx), self.batch)]
        if not batches:
            return
        pending = self._stage(self._slots[0], batches[0])
        for k in range(len(batches)):
            cur = pending
            if k + 1 < len(batches):
                # the pinned buffer of the slot being refilled was last read by a copy issued two batches ago
                nxt = self._slots[(k + 1) % 3]
                if nxt["event"] is not None:
                    nxt["event"].synchronize()
                pending = self._stage(nxt, batches[k + 1])
            torch.cuda.current_stream(self.dev).wait_event(cur["event"])
            cur["out"].record_stream(torch.cuda.current_stream(self.dev))
            if cur["attrs"] is None:
                yield cur["out"]
            else:
                names, table = cur["attrs"]
                table.record_stream(torch.cuda.current_stream(self.dev))
                if self.target_names is not None:
                    yield cur["out"], table.t().contiguous()
                else:
                    yield cur["out"], {k: table[i] for i, k in enumerate(names)}

    def stacked_targets(self) -> torch.Tensor:
        """[n, T] fp32 (host) targets of the whole set in path order (``DatasetWithTargets.stacked_targets``)."""
        if self.attributes is None or self.target_names is None:
            raise ValueError("Dataset must expose stacked_targets() to compute normalization statistics.")
        return torch.tensor([[float(a[k]) for k in self.target_names] for a in self.attributes], dtype=torch.float32)


def create_vae_dataloaders(data_base_dir: str, batch_size: int, patch_size: tuple[int, int], rank: int = 0,
                           data_source: str = "edente", train_split: float = 0.9, num_workers: int = 4,
                           seed: int | None = 42, subset_size: int | None = None, val_dir: str | None = None,
                           cache_rate: float = 0.0, distributed: bool = False, world_size: int = 1,
                           ar_vae_enabled: bool = False, regularized_attributes: dict | None = None, device="cuda",
                           **_ignored):
    """Same signature and return shape as the reference's ``create_vae_dataloaders`` (dataloaders.py:370-593):
    ``(train_loader, val_loader, train_paths, val_paths)`` with loaders that yield device batches -- ``images``, or
    ``(images, attributes)`` when ``ar_vae_enabled``."""
    if not 0 < train_split < 1:
        raise ValueError(f"train_split must be in (0, 1), got {train_split}")
    paths = list_tif_paths(data_base_dir, data_source)
    val_list = list_tif_paths(val_dir, data_source) if val_dir is not None else None
    train_paths, val_paths = split_paths(paths, train_split, seed, subset_size, val_list)
    if not 0.0 <= cache_rate <= 1.0:
        raise ValueError(f"cache_rate must be in [0, 1], got {cache_rate}")
    train_attrs = val_attrs = None
    if ar_vae_enabled:
        from .attributes import attributes_for_paths
        train_attrs = attributes_for_paths(train_paths, regularized_attributes, data_source)
        val_attrs = attributes_for_paths(val_paths, regularized_attributes, data_source)
    world = world_size if distributed else 1
    r = rank if distributed else 0
    s = seed if seed is not None else 0
    train = DeviceImageLoader(train_paths, batch_size, patch_size, device, rank=r, world_size=world, shuffle=True, seed=s,
                              num_workers=num_workers, attributes=train_attrs)
    val = DeviceImageLoader(val_paths, batch_size, patch_size, device, rank=r, world_size=world, shuffle=False, seed=s,
                            num_workers=num_workers, attributes=val_attrs)
    return train, val, train_paths, val_paths


def create_regression_dataloaders(data_base_dir: str, attributes_path, targets: list[str], batch_size: int,
                                  patch_size: tuple[int, int], train_split: float = 0.9, num_workers: int = 4,
                                  seed: int | None = 42, subset_size: int | None = None, val_dir: str | None = None,
                                  cache_rate: float = 0.0, data_source: str = "edente",
                                  normalize_attributes: dict | None = None, rank: int = 0, device="cuda"):
    """The reference's ``create_regression_dataloaders`` (dataloaders.py:596-740) on the device input pipeline:
    ``(train_loader, val_loader, train_paths, val_paths)``; batches are ``(images [b,1,Hp,Wp], targets [b,T])`` device
    tensors in ``targets`` order.  The train loader shuffles (the reference builds it with ``shuffle=True``)."""
    from .attributes import filter_attributes_for_paths, select_attribute_sources
    if not 0 < train_split < 1:
        raise ValueError(f"train_split must be in (0, 1), got {train_split}")
    if not 0.0 <= cache_rate <= 1.0:
        raise ValueError(f"cache_rate must be in [0, 1], got {cache_rate}")
    if len(targets) == 0:
        raise ValueError("targets must contain at least one entry.")
    paths = list_tif_paths(data_base_dir, data_source)
    val_list = list_tif_paths(val_dir, data_source) if val_dir is not None else None
    train_paths, val_paths = split_paths(paths, train_split, seed, subset_size, val_list)
    sources = select_attribute_sources(attributes_path, data_source)
    mapping = {t: t for t in targets}
    tr_attrs = filter_attributes_for_paths(train_paths, sources, mapping, normalize_attributes)
    va_attrs = filter_attributes_for_paths(val_paths, sources, mapping, normalize_attributes)
    s = seed if seed is not None else 0
    train = DeviceImageLoader(train_paths, batch_size, patch_size, device, shuffle=True, seed=s, num_workers=num_workers,
                              attributes=tr_attrs, target_names=list(targets))
    val = DeviceImageLoader(val_paths, batch_size, patch_size, device, shuffle=False, seed=s, num_workers=num_workers,
                            attributes=va_attrs, target_names=list(targets))
    return train, val, train_paths, val_paths
