"""CPU restatement of the reference's per-sample input transform chain -- TEST INFRASTRUCTURE ONLY (imported by
tests/ and nothing else).

  Resize(patch_size)         MONAI transform, default ``mode="area"`` -> ``torch.nn.functional.interpolate(x, size,
                             mode="area")`` (adaptive average pooling); ``data/dataloaders.py:319-329``
  LocalNormalizeByMask       ``data/transforms.py:8-32``: mean / population std over the non-zero pixels, z-score,
                             zeros stay zero, std replaced by 1.0 when <= 1e-5

Parity status: ``LocalNormalizeByMask`` is restated from the reference source line by line but could not be executed
here (its module imports tifffile, absent) -> **parity unpinned**; the resize leg is pinned to PyTorch's own
``interpolate(mode="area")`` (what MONAI calls), MONAI itself being absent.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def resize_area(img: np.ndarray, patch_size: tuple[int, int]) -> np.ndarray:
    """[H, W] -> [Hp, Wp] with torch's "area" interpolation (dataloaders.py:324 ``Resize(patch_size)``)."""
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
    return F.interpolate(t, size=tuple(patch_size), mode="area")[0, 0].numpy()


def local_normalize_by_mask(img: np.ndarray) -> np.ndarray:
    """transforms.py:16-32."""
    img = np.asarray(img)
    mask = img != 0
    pixels = img[mask]
    if pixels.size == 0:      # numpy would propagate NaN here and then mask every pixel back to 0
        return np.zeros_like(img, dtype=np.float32)
    mean = pixels.mean()
    std = pixels.std() if pixels.std() > 1e-5 else 1.0
    out = (img - mean) / std
    out[~mask] = 0.0
    return out.astype(np.float32)


def preprocess(img: np.ndarray, patch_size: tuple[int, int]) -> np.ndarray:
    """LoadImage output [H, W] -> network input [1, Hp, Wp] float32 (dataloaders.py:319-329)."""
    return local_normalize_by_mask(resize_area(img, patch_size))[None]
