"""GPU parity of the flash-style mid-block attention (forward + backward, C-ABI) against the
materialised softmax(q k^T C^-0.5) v of MONAI's SABlock restated with torch fp32 on CPU.

Tolerance: o / dq / dk / dv are bf16 and P is rounded to bf16 before the P.V product, so
max-abs <= 2% of the reference scale and rel-L2 <= 1e-2.  One case spikes a key row so that the
running max jumps mid-stream (forces the online-softmax rescale branch).
"""
import pytest
import torch

from test_gpu_ops import _r, _report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,l,c,spike", [(2, 64, 128, False), (2, 256, 128, True), (1, 128, 256, False),
                                         (1, 192, 64, True), (1, 1024, 128, False),
                                         # token counts that are not a multiple of the 64 / 32-token tiles (e.g. a 72x104
                                         # image has a 9x13 latent): masked tail keys / queries
                                         (2, 117, 128, True), (3, 96, 128, False), (1, 40, 256, False), (2, 7, 64, False),
                                         # the AR config's real mid-block shape (256x256 image, 3 levels): L = 64*64, C = 256,
                                         # and the same token count at C = 128 (the 4096^2 fp32 reference matrix is 64 MB)
                                         (1, 4096, 256, True), (2, 4096, 128, False)])
def test_attention_fwd_bwd(dev, b, l, c, spike):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(10)
    qkv = _r(torch.randn(b, l, 3 * c))
    if spike:  # a key far along the stream that dominates some queries
        qkv[:, l - 3, c:2 * c] *= 6.0
        qkv[:, 5, 0:c] *= 4.0
        qkv = _r(qkv)   # keep the inputs bf16-representable so both sides see identical numbers
    qkv.requires_grad_(True)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    att = torch.softmax(torch.einsum("blc,bmc->blm", q, k) * c ** -0.5, dim=-1)
    o_ref = torch.einsum("blm,bmc->blc", att, v)
    dout = _r(torch.randn(b, l, c))
    o_ref.backward(dout)
    qd = qkv.detach().to(dev, torch.bfloat16)
    o = torch.full((b, l, c), float("nan"), dtype=torch.bfloat16, device=dev)
    lse2 = torch.empty(b, l, device=dev)
    ops.attention_fwd(qd, o, lse2)
    torch.cuda.synchronize()
    _report(f"attn fwd[{l},{c}]", o, o_ref.detach(), max_frac=2e-2, l2=1e-2)
    lse_ref = torch.logsumexp(torch.einsum("blc,bmc->blm", q, k).detach() * c ** -0.5, dim=-1) * 1.4426950408889634
    _report("attn lse2", lse2, lse_ref, max_frac=1e-3, l2=1e-4)
    dqkv = torch.full((b, l, 3 * c), float("nan"), dtype=torch.bfloat16, device=dev)
    delta = torch.empty(b, l, device=dev)
    # the backward consumes the bf16 forward output, as the engine does
    ops.attention_bwd(qd, o, dout.to(dev, torch.bfloat16), lse2, delta, dqkv)
    torch.cuda.synchronize()
    g = qkv.grad
    _report("attn dq", dqkv[..., :c], g[..., :c], max_frac=2e-2, l2=1.5e-2)
    _report("attn dk", dqkv[..., c:2 * c], g[..., c:2 * c], max_frac=2e-2, l2=1.5e-2)
    _report("attn dv", dqkv[..., 2 * c:], g[..., 2 * c:], max_frac=2e-2, l2=1.5e-2)


def test_attention_rejects(dev):
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd._lib import PtiError
    qkv = torch.zeros(1, 96, 3 * 96, dtype=torch.bfloat16, device=dev)   # head dim 96 is not supported
    with pytest.raises(PtiError):
        ops.attention_fwd(qkv, torch.zeros(1, 96, 96, dtype=torch.bfloat16, device=dev), torch.zeros(1, 96, device=dev))
