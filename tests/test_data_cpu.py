"""CPU tests of the input pipeline's host side: TIFF codec, path listing / split, sampler
sharding against torch's own DistributedSampler, and the oracle's normalisation on hand-checked values."""
import os
import random

import numpy as np
import pytest
import torch

from pti_ldm_vae_amd.data.tiff import read_tiff, write_tiff


@pytest.mark.parametrize("dtype,big,rps", [(np.float32, False, None), (np.float32, True, 7), (np.uint16, False, 3),
                                           (np.uint8, False, None), (np.int16, True, None), (np.float64, False, 5)])
def test_tiff_roundtrip(tmp_path, dtype, big, rps):
    rng = np.random.default_rng(0)
    a = (rng.standard_normal((37, 23)) * 50).astype(dtype)
    p = str(tmp_path / "a.tif")
    write_tiff(p, a, rows_per_strip=rps, big_endian=big)
    b = read_tiff(p)
    assert b.shape == a.shape and b.dtype == np.dtype(dtype) and np.array_equal(a, b)


def test_tiff_deflate_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((50, 31)).astype(np.float32)
    a[a < 0.3] = 0.0
    p = str(tmp_path / "z.tif")
    write_tiff(p, a, rows_per_strip=8, deflate=True)
    assert os.path.getsize(p) < a.nbytes          # really compressed
    assert np.array_equal(read_tiff(p), a)


def test_tiff_known_bytes(tmp_path):
    """A hand-assembled little-endian 2x2 uint8 TIFF (not produced by write_tiff)."""
    import struct
    data = bytes([1, 2, 3, 4])
    ents = [(256, 3, 1, 2), (257, 3, 1, 2), (258, 3, 1, 8), (259, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1), (278, 3, 1, 2),
            (279, 4, 1, 4)]
    ifd = struct.pack("<H", len(ents)) + b"".join(struct.pack("<HHII", *e) for e in ents) + struct.pack("<I", 0)
    p = tmp_path / "k.tif"
    p.write_bytes(b"II" + struct.pack("<HI", 42, 12) + data + ifd)
    assert read_tiff(str(p)).tolist() == [[1, 2], [3, 4]]


def test_tiff_rejects_unsupported(tmp_path):
    p = tmp_path / "x.tif"
    p.write_bytes(b"not a tiff at all")
    with pytest.raises(ValueError):
        read_tiff(str(p))
    a = np.zeros((4, 4), np.float32)
    q = str(tmp_path / "c.tif")
    write_tiff(q, a)
    raw = bytearray(open(q, "rb").read())
    # flip the Compression tag value (259) from 1 to 5 (LZW)
    i = raw.find(bytes([0x03, 0x01, 0x03, 0x00, 0x01, 0x00, 0x00, 0x00, 0x01, 0x00]))
    assert i > 0
    raw[i + 8] = 5
    open(q, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="compress"):      # LZW: named, not mis-decoded
        read_tiff(q)


def test_list_and_split_paths(tmp_path):
    from pti_ldm_vae_amd.data import list_tif_paths, split_paths
    for sub, names in (("edente", ["b.tif", "a.tif"]), ("dente", ["d.tif", "c.tif", "skip.png"])):
        (tmp_path / sub).mkdir()
        for n in names:
            (tmp_path / sub / n).write_bytes(b"")
    both = list_tif_paths(str(tmp_path), "both")
    assert [os.path.basename(p) for p in both] == ["a.tif", "b.tif", "c.tif", "d.tif"]
    assert [os.path.basename(p) for p in list_tif_paths(str(tmp_path), "dente")] == ["c.tif", "d.tif"]
    with pytest.raises(ValueError):
        list_tif_paths(str(tmp_path), "nope")
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        list_tif_paths(str(tmp_path / "empty"), "edente")
    # the reference's split: random.seed(seed); random.shuffle(copy); cut at int(train_split * n)
    paths = [f"img{i}.tif" for i in range(10)]
    tr, va = split_paths(paths, 0.8, seed=42)
    random.seed(42)
    ref = paths.copy()
    random.shuffle(ref)
    assert tr == ref[:8] and va == ref[8:]
    tr2, va2 = split_paths(paths, 0.8, seed=42, subset_size=5)
    assert sorted(tr2 + va2) == sorted(paths[:5]) and len(tr2) == 4
    tr3, va3 = split_paths(paths, 0.8, seed=1, val_paths=["v.tif"])
    assert len(tr3) == 10 and va3 == ["v.tif"]


@pytest.mark.parametrize("n,world,shuffle", [(10, 1, True), (10, 4, True), (7, 2, False), (3, 8, True), (64, 8, True)])
def test_shard_indices_match_distributed_sampler(n, world, shuffle):
    from torch.utils.data.distributed import DistributedSampler
    from pti_ldm_vae_amd.data import shard_indices
    ds = list(range(n))
    for rank in range(world):
        for epoch in (0, 3):
            s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=42)
            s.set_epoch(epoch)
            assert shard_indices(n, rank, world, shuffle, 42, epoch) == list(iter(s))


def test_oracle_normalisation_known_answers():
    from oracle.data_pipeline import local_normalize_by_mask, preprocess, resize_area
    img = np.array([[0, 2, 4], [0, 6, 8]], np.float32)
    out = local_normalize_by_mask(img)            # non-zero pixels 2,4,6,8: mean 5, population std sqrt(5)
    assert np.allclose(out, np.array([[0, -3, -1], [0, 1, 3]], np.float32) / np.sqrt(5.0), atol=1e-6)
    flat = local_normalize_by_mask(np.array([[0, 7, 7]], np.float32))     # std 0 -> divide by 1
    assert np.array_equal(flat, np.zeros((1, 3), np.float32))
    assert np.array_equal(local_normalize_by_mask(np.zeros((2, 2), np.float32)), np.zeros((2, 2), np.float32))
    big = np.arange(16, dtype=np.float32).reshape(4, 4)
    assert np.allclose(resize_area(big, (2, 2)), [[2.5, 4.5], [10.5, 12.5]])
    assert preprocess(big, (2, 2)).shape == (1, 2, 2)


# ---- attribute join of the AR-VAE branch (reference data/dataloaders.py:108-221,432-465) --------------------------------
def test_attribute_join_follows_the_path_split(tmp_path):
    import json
    import random
    from pti_ldm_vae_amd.data.attributes import (attributes_for_paths, collate_with_attributes,
                                                 filter_attributes_for_paths, select_attribute_sources)
    from pti_ldm_vae_amd.data import split_paths
    paths = [f"/data/edente/img_{i:03d}.tif" for i in range(10)]
    table = {os.path.basename(p): {"height_0": float(i), "width_0": 10.0 * i, "unused": -1.0} for i, p in enumerate(paths)}
    f = tmp_path / "attrs_edente.json"
    f.write_text(json.dumps(table))
    ra = {"attribute_file": str(f), "normalize_attributes": {"enabled": True, "divisor": 10.0},
          "attribute_latent_mapping": {"_c": "comment", "height_0": {"latent_channel": 0}, "width_0": {"latent_channel": 1}}}
    # the reference shuffles (path, attrs) PAIRS with random.seed(seed) (dataloaders.py:469-475); joining by basename after
    # the path split must reproduce that pairing
    attrs_sorted = attributes_for_paths(paths, ra, "edente")
    random.seed(42)
    paired = list(zip(paths, attrs_sorted))
    random.shuffle(paired)
    train_p, val_p = split_paths(paths, 0.8, seed=42)
    assert [p for p, _ in paired[:8]] == train_p and [p for p, _ in paired[8:]] == val_p
    assert attributes_for_paths(train_p, ra, "edente") == [a for _, a in paired[:8]]
    assert attrs_sorted[3] == {"height_0": 0.3, "width_0": 3.0}            # filtered to the mapping, divided by 10
    # "both": the source is inferred from the path, "edente" before "dente"
    g = tmp_path / "attrs_dente.json"
    g.write_text(json.dumps({"x.tif": {"height_0": 7.0, "width_0": 8.0}}))
    src = select_attribute_sources({"edente": str(f), "dente": str(g)}, "both")
    got = filter_attributes_for_paths(["/d/dente/x.tif", paths[2]], src, {"height_0": {}, "width_0": {}}, None)
    assert got == [{"height_0": 7.0, "width_0": 8.0}, {"height_0": 2.0, "width_0": 20.0}]
    with pytest.raises(FileNotFoundError, match="Attribute entry missing"):
        filter_attributes_for_paths(["/d/dente/nope.tif"], src, {"height_0": {}}, None)
    with pytest.raises(KeyError, match="Missing attributes"):
        filter_attributes_for_paths(["/d/dente/x.tif"], src, {"depth": {}}, None)
    with pytest.raises(ValueError, match="Cannot identify data source"):
        filter_attributes_for_paths(["/d/other/x.tif"], src, {"height_0": {}}, None)
    with pytest.raises(ValueError, match="divisor must be non-zero"):
        filter_attributes_for_paths([paths[0]], src, {"height_0": {}}, {"enabled": True, "divisor": 0})
    with pytest.raises(FileNotFoundError, match="Attribute file not found"):
        select_attribute_sources(str(tmp_path / "missing.json"), "edente")
    bad = tmp_path / "bad.json"
    bad.write_text("{not json")
    with pytest.raises(ValueError, match="Invalid attribute JSON"):
        select_attribute_sources(str(bad), "edente")
    with pytest.raises(ValueError, match="must be a string or mapping"):
        select_attribute_sources(3, "edente")
    with pytest.raises(ValueError, match="attribute_latent_mapping must be provided"):
        attributes_for_paths(paths, {"attribute_file": str(f)}, "edente")
    import torch
    ims, at = collate_with_attributes([(torch.zeros(1, 2, 2), {"a": 1, "b": 2.5}), (torch.ones(1, 2, 2), {"a": 3, "b": 4})])
    assert ims.shape == (2, 1, 2, 2) and at["a"].dtype == torch.float32 and at["b"].tolist() == [2.5, 4.0]
