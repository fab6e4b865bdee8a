"""CPU tests of the PatchDiscriminator holder (no GPU, no compute through the HIP library): MONAI key names / shapes /
parameter count, the padded flat arena and its strided parameter views, state-dict exchange with the oracle restatement,
``FlatAdam``'s torch-format optimiser state over those views, and the refusal to run without a GPU."""
import pytest
import torch

KEYS = ["initial_conv.conv.weight", "initial_conv.conv.bias", "0.conv.weight", "1.conv.weight", "2.conv.weight",
        "final_conv.conv.weight", "final_conv.conv.bias"]


def test_keys_shapes_and_parameter_count_match_the_oracle():
    from oracle.patch_discriminator import PatchDiscriminator as Oracle
    from pti_ldm_vae_amd.models import PatchDiscriminator
    net, ref = PatchDiscriminator(), Oracle()
    assert [n for n, _ in net.named_parameters()] == KEYS == [n for n, _ in ref.named_parameters()]
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in ref.parameters()) == 692_769
    # reference call (train_vae.py:268-275) builds; anything the kernels do not cover is refused, not approximated
    PatchDiscriminator(spatial_dims=2, num_layers_d=3, channels=32, in_channels=1, out_channels=1, norm="INSTANCE")
    for bad in (dict(spatial_dims=3), dict(norm="BATCH"), dict(in_channels=3), dict(channels=64, num_layers_d=3)):
        with pytest.raises(ValueError):
            PatchDiscriminator(**bad)


def test_arena_layout_and_state_dict_round_trip():
    from oracle.patch_discriminator import PatchDiscriminator as Oracle
    from pti_ldm_vae_amd.models import PatchDiscriminator
    torch.manual_seed(0)
    net, ref = PatchDiscriminator(), Oracle()
    net.load_state_dict(ref.state_dict())
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    sd = net.state_dict()
    assert all(v.is_contiguous() for v in sd.values())
    ref2 = Oracle()
    ref2.load_state_dict(sd, strict=True)
    # the arena holds [cout][ky][kx][cin] rows; padding (first layer columns 16..31, last layer rows 1..31) is zero
    lay = net.layers[2]
    w = net.param_arena[lay["w_off"]:lay["w_off"] + lay["rows"] * lay["k"]].view(lay["cout"], 4, 4, lay["cin"])
    assert torch.equal(w.permute(0, 3, 1, 2), ref.state_dict()["1.conv.weight"])
    assert int((net.param_arena != 0).sum()) <= 692_769 and net.param_arena.numel() == 820_288
    first = net.param_arena[:32 * 32].view(32, 32)
    assert float(first[:, 16:].abs().max()) == 0.0 and torch.equal(first[:, :16].reshape(32, 1, 4, 4), sd["initial_conv.conv.weight"])
    with pytest.raises(RuntimeError):
        net.load_state_dict({k: v for k, v in sd.items() if k != "0.conv.weight"})
    net.attach_grads()
    for n, p in net.named_parameters():
        assert p.grad.shape == p.shape and p.grad.data_ptr() == net.grad_view(n).data_ptr()


def test_flat_adam_state_dict_is_torch_adam_compatible():
    from pti_ldm_vae_amd.models import PatchDiscriminator
    from pti_ldm_vae_amd.optim import FlatAdam
    net = PatchDiscriminator()
    opt = FlatAdam(net, 1e-3)
    opt.step_count = 3
    opt.exp_avg.uniform_(-1, 1)
    opt.exp_avg_sq.uniform_(0, 1)
    sd = opt.state_dict()
    ref = torch.optim.Adam(net.parameters(), lr=1e-3)
    ref.load_state_dict(sd)                      # torch accepts it: same parameter order, dense tensors of the right shapes
    for i, (n, p) in enumerate(net.named_parameters()):
        assert tuple(sd["state"][i]["exp_avg"].shape) == tuple(p.shape) and sd["state"][i]["exp_avg"].is_contiguous()
    opt2 = FlatAdam(PatchDiscriminator(), 1e-3)
    opt2.load_state_dict(sd)
    for n, _ in net.named_parameters():
        assert torch.equal(net.slot_view(opt.exp_avg, n), opt2.net.slot_view(opt2.exp_avg, n))
    assert opt2.step_count == 3


def test_no_cpu_fallback():
    from pti_ldm_vae_amd.models import PatchDiscriminator
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PatchDiscriminator()(torch.zeros(1, 1, 64, 64))


def test_oracle_structure():
    """Output sizes of the restated MONAI module on the reference's input: 256 -> 128 -> 64 -> 32 -> 31 -> 30."""
    from oracle.patch_discriminator import PatchDiscriminator as Oracle, patch_adversarial_loss
    torch.manual_seed(0)
    outs = Oracle()(torch.randn(1, 1, 256, 256))
    assert [tuple(o.shape[1:]) for o in outs] == [(32, 128, 128), (64, 64, 64), (128, 32, 32), (256, 31, 31), (1, 30, 30)]
    lo = torch.tensor([[-2.0, 0.5]])
    assert float(patch_adversarial_loss(lo, True, True)) == pytest.approx(((-0.1 - 1) ** 2 + (0.5 - 1) ** 2) / 2)
    assert float(patch_adversarial_loss(lo, False, False)) == float(patch_adversarial_loss(lo, True, True))   # generator: always "real"
    assert float(patch_adversarial_loss(lo, False, True)) == pytest.approx((0.01 + 0.25) / 2)


def _golden_state():
    """The seeded state and inputs of oracle/make_golden.py::discriminator_golden."""
    from oracle.patch_discriminator import PatchDiscriminator as Oracle
    torch.manual_seed(2024)
    ref = Oracle()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(5.0)
    g = torch.Generator().manual_seed(2025)
    x = torch.randn(2, 1, 96, 96, generator=g) * 0.8
    real = torch.randn(2, 1, 96, 96, generator=g) * 0.8 + 0.2
    return ref, x, real


def test_oracle_reproduces_frozen_discriminator_vectors():
    """tests/golden/disc_golden.npz freezes the oracle's outputs (parity unpinned w.r.t. MONAI: a regression vector, so
    that a later edit of the restatement cannot move silently)."""
    import os
    import numpy as np
    from oracle.patch_discriminator import patch_adversarial_loss as pal
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "disc_golden.npz"))
    ref, x, real = _golden_state()
    with torch.no_grad():
        logits = ref(x)[-1]
        assert np.allclose(logits.numpy(), gold["logits"], rtol=1e-5, atol=1e-5)
        assert float(pal(logits, True, False)) == pytest.approx(float(gold["gen"]), rel=1e-5)
        assert float(pal(logits, False, True)) == pytest.approx(float(gold["fake"]), rel=1e-5)
        assert float(pal(ref(real)[-1], True, True)) == pytest.approx(float(gold["real"]), rel=1e-5)
