"""GPU parity of the eight-blocks-per-workgroup weight-gradient kernel (csrc/wgrad_mfma.hip, ``wgrad_mfma6_kernel``; the
default for layers with Cout % 128 == 0 and Cin % 64 == 0, shape B -- Cout % 64 == 0 -- behind PTI_WGRAD_V6=2) against
torch's fp32 weight gradient of ``nn.Conv2d(cin, cout, 3, padding=1)`` (the MONAI AEKLResBlock convolution, reference
src/pti_ldm_vae/models/autoencoder.py:67-79) on the same bf16 values, against the v4 kernel it replaces, run to run, and
through the batched entry point the training step uses."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def torch_wgrad(x, dy):
    cout, cin = dy.shape[3], x.shape[3]
    wt = torch.zeros(cout, cin, 3, 3, device=x.device, requires_grad=True)
    y = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt, padding=1)
    y.backward(dy.float().permute(0, 3, 1, 2))
    return wt.grad, dy.float().sum((0, 1, 2))


def run(ops, x, dy, mode, monkeypatch):
    monkeypatch.setenv("PTI_WGRAD_V6", mode)
    cout, cin = dy.shape[3], x.shape[3]
    dw, db = torch.zeros(cout, cin, 3, 3, device=x.device), torch.zeros(cout, device=x.device)
    ops.conv_wgrad_mfma(x, dy, dw, db)
    torch.cuda.synchronize()
    return dw, db


# (n, h, w, cin, cout): ragged edges in both directions, one-tile images, both shapes, several blocks per workgroup grid
CASES = [(2, 8, 16, 64, 128), (3, 13, 21, 64, 128), (2, 30, 20, 128, 128), (1, 4, 16, 128, 256), (5, 7, 5, 64, 128),
         (2, 8, 16, 64, 64), (3, 13, 21, 128, 64), (2, 30, 20, 64, 64), (1, 3, 40, 64, 192), (4, 17, 9, 192, 64),
         (8, 32, 32, 128, 128)]


@pytest.mark.parametrize("n,h,w,cin,cout", CASES)
def test_v6_matches_torch_and_v4(dev, monkeypatch, n, h, w, cin, cout):
    from pti_ldm_vae_amd import ops
    g = torch.Generator(device=dev).manual_seed(1000 * h + w)
    x = torch.randn(n, h, w, cin, device=dev, generator=g).bfloat16()
    dy = torch.randn(n, h, w, cout, device=dev, generator=g).bfloat16()
    ref, refb = torch_wgrad(x, dy)
    dw6, db6 = run(ops, x, dy, "2", monkeypatch)
    dw4, db4 = run(ops, x, dy, "0", monkeypatch)
    # fp32 accumulation of exact bf16 products: only the summation order differs (tolerance: 1e-5 relative L2)
    assert ((dw6 - ref).norm() / ref.norm()).item() < 1e-5
    assert ((db6 - refb).norm() / refb.norm().clamp_min(1e-6)).item() < 1e-5
    assert ((dw6 - dw4).norm() / dw4.norm()).item() < 1e-5
    dw6b, db6b = run(ops, x, dy, "2", monkeypatch)
    assert torch.equal(dw6, dw6b) and torch.equal(db6, db6b), "fixed summation order: bitwise reproducible"


def test_v6_is_what_runs_by_default_on_the_wide_layers(dev, monkeypatch):
    from pti_ldm_vae_amd import ops
    monkeypatch.delenv("PTI_WGRAD_V6", raising=False)
    x = torch.randn(2, 16, 16, 128, device=dev).bfloat16()
    dy = torch.randn(2, 16, 16, 128, device=dev).bfloat16()
    dw, db = torch.zeros(128, 128, 3, 3, device=dev), torch.zeros(128, device=dev)
    ops.conv_wgrad_mfma_batched([(x, dy, dw, db)], accumulate=False)
    torch.cuda.synchronize()
    assert "wgrad_mfma6_kernel" in ops.last_kernel_name()
    ref, refb = torch_wgrad(x, dy)
    assert ((dw - ref).norm() / ref.norm()).item() < 1e-5 and ((db - refb).norm() / refb.norm()).item() < 1e-5


def test_batched_launch_mixes_all_kernel_modes(dev, monkeypatch):
    """One batched call with jobs for every mode (32-channel pair mode, 64-channel two-block mode or shape B, shape A):
    every job's gradient is right, accumulate=True adds to what is there."""
    from pti_ldm_vae_amd import ops
    for mode in ("1", "2"):
        monkeypatch.setenv("PTI_WGRAD_V6", mode)
        jobs, refs = [], []
        for i, (cin, cout, hw) in enumerate([(32, 32, 24), (64, 64, 16), (128, 128, 8), (64, 128, 12), (128, 64, 12), (32, 64, 8)]):
            g = torch.Generator(device=dev).manual_seed(77 + i)
            x = torch.randn(3, hw, hw, cin, device=dev, generator=g).bfloat16()
            dy = torch.randn(3, hw, hw, cout, device=dev, generator=g).bfloat16()
            dw = torch.full((cout, cin, 3, 3), 0.5, device=dev)
            db = torch.full((cout,), -1.0, device=dev)
            jobs.append((x, dy, dw, db))
            r, rb = torch_wgrad(x, dy)
            refs.append((r + 0.5, rb - 1.0))
        ops.conv_wgrad_mfma_batched(jobs, accumulate=True)
        torch.cuda.synchronize()
        for (x, dy, dw, db), (r, rb) in zip(jobs, refs):
            assert ((dw - r).norm() / r.norm()).item() < 1e-5, (mode, tuple(dw.shape))
            assert ((db - rb).norm() / rb.norm()).item() < 1e-5, (mode, tuple(dw.shape))
