"""CPU tests of the perceptual-loss term: the product's parameter holder (``models/perceptual.py``:
torchvision / lpips state_dict key names and shapes, the local-file weight loader with files written here -- the real
pretrained files cannot be fetched --, refusal without weights, refusal of CPU tensors: the product is HIP-only) and the
CHECKER ``oracle/perceptual.py`` (LPIPS's defining properties: zero for identical inputs, symmetric, batch mean, 1 -> 3
channel repeat; tap shapes).  Parity of the term vs the reference is unpinned (no weights, no reference output)."""
import pytest
import torch

# torchvision.models.squeezenet1_1().features: conv at index 0, Fire modules at 3,4,6,7,9,10,11,12 (published layout)
FIRES = {3: (64, 16, 64), 4: (128, 16, 64), 6: (128, 32, 128), 7: (256, 32, 128), 9: (256, 48, 192), 10: (384, 48, 192),
         11: (384, 64, 256), 12: (512, 64, 256)}


def _expected_keys():
    keys = {"features.0.weight": (64, 3, 3, 3), "features.0.bias": (64,)}
    for i, (cin, sq, ex) in FIRES.items():
        keys[f"features.{i}.squeeze.weight"], keys[f"features.{i}.squeeze.bias"] = (sq, cin, 1, 1), (sq,)
        keys[f"features.{i}.expand1x1.weight"], keys[f"features.{i}.expand1x1.bias"] = (ex, sq, 1, 1), (ex,)
        keys[f"features.{i}.expand3x3.weight"], keys[f"features.{i}.expand3x3.bias"] = (ex, sq, 3, 3), (ex,)
    for k, c in enumerate((64, 128, 256, 384, 384, 512, 512)):
        keys[f"lin{k}.model.1.weight"] = (1, c, 1, 1)
    return keys


def test_state_dict_matches_the_two_packages_layouts():
    from pti_ldm_vae_amd.models import SqueezeLPIPS
    net = SqueezeLPIPS()
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == _expected_keys()
    assert sum(p.numel() for p in net.features.parameters()) == 722_496      # squeezenet1_1 without its classifier
    assert not any(p.requires_grad for p in net.parameters()) and not net.training
    # the oracle evaluates the product's state_dict as it stands (same key names), and yields the published tap shapes
    from oracle import perceptual as OP
    taps = OP.taps(OP.cpu_state(net), torch.randn(1, 3, 256, 256))
    assert [tuple(t.shape[1:]) for t in taps] == [(64, 127, 127), (128, 63, 63), (256, 31, 31), (384, 15, 15), (384, 15, 15),
                                                   (512, 15, 15), (512, 15, 15)]
    assert {k: tuple(v.shape) for k, v in OP.random_state(3).items()} == _expected_keys()


def test_product_module_refuses_cpu_tensors():
    """HIP-only like every other module of the package: no torch formulation of the term is left in the product."""
    import inspect
    from pti_ldm_vae_amd.models import PerceptualLoss
    from pti_ldm_vae_amd.models import perceptual as P
    loss = PerceptualLoss(allow_random_init=True)
    a = torch.rand(2, 1, 32, 32)
    for call in (lambda: loss(a, a), lambda: loss.target_taps(a), lambda: loss.from_taps(a, []), lambda: loss.net(a.repeat(1, 3, 1, 1), a.repeat(1, 3, 1, 1))):
        with pytest.raises(RuntimeError, match="MI355X only"):
            call()
    src = inspect.getsource(P)
    assert "oracle" not in src.replace("oracle/perceptual.py", "") and "import oracle" not in src
    assert not hasattr(P, "lpips_tap_torch") and not hasattr(P.SqueezeLPIPS, "compare") and not hasattr(P.SqueezeLPIPS, "taps")


def test_local_weight_files_round_trip(tmp_path):
    from pti_ldm_vae_amd.models import PerceptualLoss, SqueezeLPIPS
    torch.manual_seed(0)
    src = SqueezeLPIPS()
    sd = src.state_dict()
    backbone = {k: v for k, v in sd.items() if k.startswith("features.")}
    backbone["classifier.1.weight"] = torch.zeros(1000, 512, 1, 1)            # present in torchvision's file, ignored
    lin = {k: v.abs() for k, v in sd.items() if k.startswith("lin")}
    torch.save(backbone, tmp_path / "squeezenet1_1.pth")
    torch.save(lin, tmp_path / "squeeze.pth")
    with pytest.raises(RuntimeError, match="not available offline"):
        PerceptualLoss()
    loss = PerceptualLoss(weights=(str(tmp_path / "squeezenet1_1.pth"), str(tmp_path / "squeeze.pth")))
    assert loss.pretrained
    for k, v in loss.net.state_dict().items():
        assert torch.equal(v, backbone[k] if k.startswith("features.") else lin[k])
    del lin["lin3.model.1.weight"]
    torch.save(lin, tmp_path / "broken.pth")
    with pytest.raises(KeyError):
        PerceptualLoss(weights=(str(tmp_path / "squeezenet1_1.pth"), str(tmp_path / "broken.pth")))
    with pytest.raises(ValueError):
        PerceptualLoss(spatial_dims=3, allow_random_init=True)


def test_lpips_properties_of_the_oracle():
    """The checker itself: LPIPS's defining properties on a seeded random network with non-negative lin weights (as in
    the trained LPIPS)."""
    from oracle import perceptual as OP
    sd = OP.random_state(1)
    loss = lambda x, y: OP.perceptual_loss(sd, x, y)     # noqa: E731
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(3, 1, 64, 64, generator=g) * 2 - 1, torch.rand(3, 1, 64, 64, generator=g) * 2 - 1
    assert float(loss(a, a)) == 0.0
    assert float(loss(a, b)) > 0 and float(loss(a, b)) == pytest.approx(float(loss(b, a)), rel=1e-6)
    per = torch.stack([loss(a[i:i + 1], b[i:i + 1]) for i in range(3)])
    assert float(loss(a, b)) == pytest.approx(float(per.mean()), rel=1e-5)
    a3 = a.repeat(1, 3, 1, 1)                   # the reference repeats to three channels itself before the call
    assert float(loss(a3, b)) == pytest.approx(float(loss(a, b)), rel=1e-6)
    x = a.clone().requires_grad_(True)
    loss(x, b).backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
    assert tuple(OP.lpips(sd, a3, b.repeat(1, 3, 1, 1)).shape) == (3, 1, 1, 1)


def test_trainer_argument_validation():
    """perceptual_weight without a module is an error at construction (checked before any GPU work)."""
    from pti_ldm_vae_amd import trainer
    import inspect
    sig = inspect.signature(trainer.VAETrainer.__init__)
    assert "perceptual" in sig.parameters and "perceptual_weight" in sig.parameters


def test_folded_first_layer_algebra_on_cpu():
    """``perceptual_engine.fold_first_layer``: conv(3 -> 64, 3x3, stride 2, no padding) on the three scaled copies of a
    one-channel image == a 1 -> 64 convolution with the folded weights / bias (exact algebra: the layer has no padding).
    This is the operand of the HIP first-layer kernels; checked here against the oracle's first layer in float64."""
    import torch
    import torch.nn.functional as F
    from pti_ldm_vae_amd.models.perceptual import SqueezeLPIPS
    from pti_ldm_vae_amd.perceptual_engine import fold_first_layer
    from pti_ldm_vae_amd.utils.losses import ensure_three_channels
    from oracle import perceptual as OP
    torch.manual_seed(0)
    net = SqueezeLPIPS().double()
    x = torch.rand(2, 1, 21, 18, dtype=torch.float64)
    want = OP.feature_layer(OP.cpu_state(net, torch.float64), 0, OP.scale_input(OP.three_channels(x)))
    w10 = fold_first_layer(net).double()                                  # [10, 64]: 9 taps (row-major) + bias
    got = F.conv2d(x, w10[:9].t().reshape(64, 1, 3, 3), w10[9], stride=2)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)                # w10 is stored in fp32


def test_trunk_plan_matches_the_lpips_slices():
    """The HIP trunk's plan (pools, Fire modules, which outputs are taps) against ``SqueezeLPIPS.SLICES`` / the layer
    list of ``squeezenet1_1_features()``."""
    from pti_ldm_vae_amd.models.perceptual import Fire, SqueezeLPIPS, _Pool, squeezenet1_1_features
    from pti_ldm_vae_amd.perceptual_engine import _PLAN
    feats = squeezenet1_1_features()
    idx = 2                                                               # the trunk starts after conv + ReLU
    tap_ends = {b - 1 for _, b in SqueezeLPIPS.SLICES}                    # feature index whose output is a tap
    for st in _PLAN:
        if st[0] == "pool":
            assert isinstance(feats[idx], _Pool)
        else:
            assert isinstance(feats[idx], Fire) and st[1] == idx and st[2] == (idx in tap_ends)
        idx += 1
    assert idx == len(feats) and 1 in tap_ends
