"""``PTI_FWD_ACT_DTYPE=bf16``: BASELINE.json names bf16; the default engine runs fp16 forward operands/activations (same
MFMA rate, 8x finer rounding) and says so in bench.py's ``dtype``.  The all-bf16 mode is kept as a knob, so it gets its OWN
model-level parity evidence here with its OWN stated tolerances (VERDICT r2 weak #2 / next #6c):

  * sampled forward (shared eps) vs the fp32 CPU oracle: recon MSE <= 2e-4, mu / sigma rel-L2 <= 2e-2.  MEASURED 1.00e-4 at
    A@64 (round 3, MI355X): the all-bf16 mode sits AT north_star's 1e-4 bound, not under it with margin -- this is the
    reason the default engine (and bench.py's headline) run fp16 forward operands (2e-6 under the same test) and say so;
  * eps-free ``decode(mu)``: MSE <= 5e-4 (measured 2.4e-4; the fp16 default is gated at 1e-4 and measures 5-9e-6);
  * one training step: loss 2e-3 relative, gradient cosine >= 0.995 -- measured 0.9977 (fp16 default: gated 1e-3 / 0.999,
    measures 0.9999): the all-bf16 mode also misses SURVEY 8(d)'s 0.999 gradient-cosine bar.
The variable is read when the engine is built, so each test sets it before the model's first use.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("tag,batch,size", [("A", 2, 64), ("A", 1, 256)])
def test_forward_parity_all_bf16(dev, monkeypatch, tag, batch, size):
    monkeypatch.setenv("PTI_FWD_ACT_DTYPE", "bf16")
    from oracle.autoencoderkl import CONFIG_A
    from tests.test_gpu_model import _build, _fwd_hip, _inputs
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    oracle, model = _build(CONFIG_A, dev)
    assert model.autoencoder.engine().act_dtype == torch.bfloat16
    x, eps = _inputs(CONFIG_A, batch, size)
    with torch.no_grad():
        mu_o, sig_o = oracle.encode(x)
        rec_o = oracle.decode(mu_o + eps * sig_o)
        det_o = oracle.decode(mu_o)
        rec, mu, sig = _fwd_hip(model, x.to(dev), eps.to(dev))
        det = model.reconstruct_deterministic(x.to(dev))
    mse = ((rec.cpu() - rec_o) ** 2).mean().item()
    mse_det = ((det.cpu() - det_o) ** 2).mean().item()
    print(f"[bf16 fwd {tag}@{size}] recon MSE {mse:.2e}  decode(mu) MSE {mse_det:.2e}  mu relL2 {_rel(mu.cpu(), mu_o):.2e}  "
          f"sigma relL2 {_rel(sig.cpu(), sig_o):.2e}")
    assert mse <= 2e-4          # measured 1.0e-4: at north_star's bound, see the module docstring
    assert mse_det <= 5e-4
    assert _rel(mu.cpu(), mu_o) <= 2e-2 and _rel(sig.cpu(), sig_o) <= 2e-2


def test_training_step_parity_all_bf16(dev, monkeypatch):
    monkeypatch.setenv("PTI_FWD_ACT_DTYPE", "bf16")
    from oracle.autoencoderkl import CONFIG_A
    from oracle.losses import train_step_losses
    from pti_ldm_vae_amd.models import compute_kl_loss
    from tests.test_gpu_model import _build, _cos, _fwd_hip, _inputs
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    oracle, model = _build(CONFIG_A, dev)
    x, eps = _inputs(CONFIG_A, 2, 64)
    loss_o, *_ = train_step_losses(oracle, x, eps)
    loss_o.backward()
    xd = x.to(dev)
    rec, mu, sig = _fwd_hip(model, xd, eps.to(dev))
    loss = torch.nn.functional.l1_loss(rec, xd) + 1e-3 * compute_kl_loss(mu, sig)
    loss.backward()
    torch.cuda.synchronize()
    go = {n: p.grad for n, p in oracle.named_parameters()}
    fg = torch.cat([p.grad.detach().cpu().flatten() for _, p in model.autoencoder.named_parameters()])
    fo = torch.cat([go[n].flatten() for n, _ in model.autoencoder.named_parameters()])
    cos = _cos(fg, fo)
    print(f"[bf16 fwd step] loss {loss.item():.6f} vs {loss_o.item():.6f}  grad cosine {cos:.5f}")
    assert loss.item() == pytest.approx(loss_o.item(), rel=2e-3)
    assert torch.isfinite(fg).all() and cos >= 0.995        # measured 0.9977, see the module docstring
