"""CPU tests of the host-side mirror of the reference interface (no GPU compute): loss functions vs the
reference-generated goldens, config loader semantics, VAEModel surface / state-dict / flat arenas."""
import json
import os

import pytest
import torch

from pti_ldm_vae_amd.models import (LatentRegressor, VAEModel, compute_ar_vae_loss, compute_kl_loss,
                                    compute_total_loss)
from pti_ldm_vae_amd.utils import (ensure_three_channels, parse_config, read_config, resolve_ar_settings,
                                   resolve_bool)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(GOLD, "losses_golden.json")))["cases"]


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "kl"], ids=lambda c: f"seed{c['seed']}")
def test_kl_vs_reference_golden(case):
    g = torch.Generator().manual_seed(case["seed"])
    mu = torch.randn(*case["shape"], generator=g)
    t = torch.randn(*case["shape"], generator=g) * case["scale"]
    sig = torch.exp(0.5 * t)
    assert float(compute_kl_loss(mu, t)) == pytest.approx(case["kl_logvar"], rel=1e-6)
    assert float(compute_kl_loss(mu, sig, input_is_logvar=False)) == pytest.approx(case["kl_sigma_flag"], rel=1e-6)
    assert float(compute_kl_loss(mu, sig)) == pytest.approx(case["kl_sigma_as_logvar"], rel=1e-6)


def test_total_vs_reference_golden():
    c = [c for c in CASES if c["kind"] == "total"][0]
    t = [torch.tensor(v) for v in c["vals"]]
    assert float(compute_total_loss(*t, ar_vae_enabled=True, **c["args"])) == pytest.approx(c["enabled"], rel=1e-6)
    assert float(compute_total_loss(*t, ar_vae_enabled=False, **c["args"])) == pytest.approx(c["disabled"], rel=1e-6)


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "ar"], ids=lambda c: f"seed{c['seed']}")
def test_vectorised_ar_loss_vs_reference_golden(case):
    g = torch.Generator().manual_seed(case["seed"])
    z = torch.randn(*case["zshape"], generator=g)
    attrs = {k: torch.tensor(v) for k, v in case["attrs"].items()}
    tot, per, cnt, dl = compute_ar_vae_loss(z, attrs, case["mapping"], "all", None, case["delta_global"])
    assert float(tot) == pytest.approx(case["total"], rel=1e-5)
    for k in case["per_attr"]:
        assert float(per[k]) == pytest.approx(case["per_attr"][k], rel=1e-5, abs=1e-7)
        assert cnt[k] == case["pairs"][k] and dl[k] == case["deltas"][k]


def test_ar_loss_subset_matches_oracle_sampling():
    import random
    from oracle.losses import ar_vae_loss
    g = torch.Generator().manual_seed(3)
    z, a = torch.randn(6, 4, generator=g), {"h": torch.rand(6, generator=g)}
    mp = {"h": {"latent_channel": 1, "delta": 2.0}}
    random.seed(5)
    t1, *_ = compute_ar_vae_loss(z, a, mp, "subset", 7, None)
    random.seed(5)
    t2, *_ = ar_vae_loss(z, a, mp, "subset", 7, None)
    assert float(t1) == pytest.approx(float(t2), rel=1e-6)
    with pytest.raises(ValueError):
        compute_ar_vae_loss(z, a, mp, "subset", None, None)
    with pytest.raises(KeyError):
        compute_ar_vae_loss(z, {}, mp, "all", None, None)


def test_ensure_three_channels():
    x = torch.arange(8.).reshape(2, 1, 2, 2)
    y = ensure_three_channels(x)
    assert y.shape == (2, 3, 2, 2) and torch.equal(y[:, 0], y[:, 2])
    assert ensure_three_channels(y) is y
    with pytest.raises(ValueError):
        ensure_three_channels(torch.zeros(2, 2, 4, 4))
    with pytest.raises(ValueError):
        ensure_three_channels(torch.zeros(2, 4, 4))


def test_config_reference_semantics():
    cfg = read_config(os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    d = cfg["autoencoder_def"]
    assert d["spatial_dims"] == 2 and d["in_channels"] == 1 and d["latent_channels"] == 4
    # dotted refs are NOT resolved (MONAI would not either) and are rescued by the script-side fall-backs
    assert cfg["autoencoder_train"]["ar_vae_weight"] == "@regularized_attributes.gamma"
    assert resolve_ar_settings(cfg["autoencoder_train"], cfg["regularized_attributes"]) == (False, 0.5, "all", None)
    ar = read_config(os.path.join(ROOT, "config", "ar_vae_dente_kl1e3.json"))
    assert resolve_ar_settings(ar["autoencoder_train"], ar["regularized_attributes"])[0] is True
    assert parse_config({"a": {"b": [1, {"c": 7}]}, "x": "@a::b::1::c", "y": "@a#b#0", "z": "keep @a"}) == \
        {"a": {"b": [1, {"c": 7}]}, "x": 7, "y": 1, "z": "keep @a"}
    with pytest.raises(KeyError):
        parse_config({"x": "@missing"})
    assert [resolve_bool(v) for v in (True, "true", "YES", "false", "", "@foo.bar", None, 0, 2)] == \
        [True, True, True, False, False, False, False, False, True]


def test_vaemodel_surface_and_state_dict():
    from oracle.autoencoderkl import CONFIG_A, build_oracle
    torch.manual_seed(42)
    m = VAEModel.from_config(CONFIG_A)
    o = build_oracle(CONFIG_A, 42)
    sd, so = m.state_dict(), o.state_dict()
    assert list(sd) == list(so) and all(torch.equal(sd[k], so[k]) for k in so)   # same names, same seeded init
    assert sum(p.numel() for p in m.parameters()) == 4_562_593
    assert m.autoencoder.in_channels == 1
    m.load_state_dict({k: v + 1 for k, v in so.items()})
    assert torch.equal(m.state_dict()["quant_conv_mu.conv.bias"], so["quant_conv_mu.conv.bias"] + 1)
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(so.items())[:-1]})            # strict
    # parents can recurse into it (the reference's override could not)
    holder = torch.nn.Module()
    holder.vae = m
    assert any(k.startswith("vae.autoencoder.encoder.") for k in holder.state_dict())
    # no CPU fallback
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 64, 64))
    assert "AutoencoderKL" in repr(m)


def test_flat_arena_layout():
    from oracle.autoencoderkl import CONFIG_A
    ae = VAEModel.from_config(CONFIG_A).autoencoder
    arena = ae.param_arena
    for name, p in ae.named_parameters():
        o, n, shp = ae._slots[name]
        assert p.data_ptr() == arena[o:o + n].data_ptr() and tuple(p.shape) == shp and o % 4 == 0
    # q|k|v projection weights / biases are adjacent => one [3C,C] / [3C] view
    pre = "encoder.blocks.13.attn."
    oq, n, _ = ae._slots[pre + "to_q.weight"]
    assert ae._slots[pre + "to_k.weight"][0] == oq + n and ae._slots[pre + "to_v.weight"][0] == oq + 2 * n
    ob, nb, _ = ae._slots[pre + "to_q.bias"]
    assert ae._slots[pre + "to_k.bias"][0] == ob + nb and ae._slots[pre + "to_v.bias"][0] == ob + 2 * nb
    (e0, e1), (d0, d1) = ae.arena_regions()
    assert e0 == 0 and e1 == d0 and d1 == arena.numel()
    assert all((o < e1) == name.startswith(("encoder.", "quant_conv_")) for name, (o, _, _) in ae._slots.items())
    # in-place updates through a parameter show up in the arena and bump that parameter's version counter
    # (the engine's repack trigger sums them)
    pb = ae.quant_conv_mu.conv.bias
    v0, a0 = pb._version, arena[ae._slots["quant_conv_mu.conv.bias"][0]].item()
    with torch.no_grad():
        pb.add_(1.0)
    assert pb._version > v0 and arena[ae._slots["quant_conv_mu.conv.bias"][0]].item() == pytest.approx(a0 + 1.0)
    ae.attach_grads()
    assert all(p.grad is not None and p.grad.shape == p.shape for p in ae.parameters())
    with pytest.raises(TypeError):
        ae.half()


def test_ctor_validation_and_regressor():
    from oracle.autoencoderkl import CONFIG_A
    with pytest.raises(ValueError):
        VAEModel.from_config({**CONFIG_A, "norm_num_groups": 24})
    with pytest.raises(ValueError):
        VAEModel.from_config({**CONFIG_A, "attention_levels": [False]})
    with pytest.raises(ValueError):
        VAEModel.from_config({**CONFIG_A, "spatial_dims": 3})
    r = LatentRegressor(16, [8, 4], 3, dropout=0.1, activation="gelu")
    assert r(torch.zeros(5, 16)).shape == (5, 3)
    with pytest.raises(ValueError):
        LatentRegressor(16, [8], 3, activation="swish")
    with pytest.raises(ValueError):
        LatentRegressor(0, [8], 3)


def test_flat_adam_state_dict_is_torch_adam_compatible():
    """FlatAdam.state_dict() loads into torch.optim.Adam over VAEModel.parameters() and back."""
    from oracle.autoencoderkl import CONFIG_A
    from pti_ldm_vae_amd.optim import FlatAdam
    m = VAEModel.from_config(dict(CONFIG_A, channels=[32, 32], attention_levels=[False, False], norm_num_groups=16))
    opt = FlatAdam(m.autoencoder, lr=1e-4)
    opt.step_count = 3
    opt.exp_avg.fill_(0.5)
    opt.exp_avg_sq.fill_(0.25)
    sd = opt.state_dict()
    ref = torch.optim.Adam(m.parameters(), lr=1.0)
    ref.load_state_dict(sd)
    assert ref.param_groups[0]["lr"] == 1e-4
    st = ref.state[next(iter(m.parameters()))]
    assert float(st["step"]) == 3 and torch.all(st["exp_avg"] == 0.5)
    opt2 = FlatAdam(m.autoencoder, lr=5.0)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.step_count == 3 and opt2.lr == 1e-4
    used = torch.zeros_like(opt2.exp_avg, dtype=torch.bool)
    for n, (o, cnt, _) in m.autoencoder._slots.items():
        used[o:o + cnt] = True
    assert torch.all(opt2.exp_avg[used] == 0.5) and torch.all(opt2.exp_avg_sq[used] == 0.25)


# ---- _prepare_batch container forms (reference vae_scripts/train_vae.py:183-243) --------------------------------------
def test_prepare_batch_container_forms():
    from pti_ldm_vae_amd.trainer import prepare_batch
    dev = torch.device("cpu")
    x = torch.randn(3, 1, 4, 4)
    attrs = {"h": torch.tensor([1.0, 2.0, 3.0]), "w": torch.tensor([4.0, 5.0, 6.0])}
    im, a = prepare_batch(x, dev, False)                                   # plain tensor
    assert im is x or torch.equal(im, x)
    assert a is None
    im, a = prepare_batch((x, attrs), dev, True)                           # collate_with_attributes output
    assert torch.equal(im, x) and torch.equal(a["w"], attrs["w"])
    im, a = prepare_batch([x, attrs], dev, True)                           # [images, dict]
    assert torch.equal(im, x) and set(a) == {"h", "w"}
    pairs = [(x[i], {"h": float(i), "w": 2.0 * i}) for i in range(3)]      # un-stacked list of (image, attrs)
    im, a = prepare_batch(pairs, dev, True)
    assert im.shape == x.shape and a["h"].dtype == torch.float32 and a["w"].tolist() == [0.0, 2.0, 4.0]
    im, a = prepare_batch([(x[0], None), (x[1], None)], dev, False)        # pairs without attributes
    assert im.shape == (2, 1, 4, 4) and a is None
    with pytest.raises(ValueError, match="Empty batch"):
        prepare_batch([], dev, False)
    with pytest.raises(ValueError, match="attributes are missing"):
        prepare_batch(x, dev, True)
    with pytest.raises(TypeError, match="Unsupported list batch"):
        prepare_batch([x, x, x], dev, False)
    with pytest.raises(TypeError, match="Unsupported batch type"):
        prepare_batch({"images": x}, dev, False)


def test_ar_settings_validation():
    from pti_ldm_vae_amd.trainer import ARSettings
    mp = {"_comment": "x", "h": {"latent_channel": 1, "delta": 2.0}, "w": {"latent_channel": 0}}
    st = ARSettings(mp, gamma=0.5, delta_global={"enabled": True, "value": 3.0})
    assert st.names == ["h", "w"] and st.channels == [1, 0] and st.deltas == [2.0, 3.0]
    with pytest.raises(ValueError, match="Delta not provided"):
        ARSettings(mp, gamma=0.5)
    with pytest.raises(ValueError, match="exceeds latent size"):
        ARSettings(mp, gamma=0.5, delta_global={"enabled": True, "value": 3.0}, latent_channels=1)
    with pytest.raises(ValueError, match="pairwise must be"):
        ARSettings(mp, gamma=0.5, pairwise="some")
    with pytest.raises(ValueError, match="subset_pairs"):
        ARSettings(mp, gamma=0.5, pairwise="subset")


def test_config_3_builds_the_same_model_as_config_a():
    """BASELINE configs[2] (``vae_both_no_adv.json``, the 8-GPU data-parallel run) must describe config A's network: the
    single-GPU parity and bench evidence then carries over to it -- checked on the parameter holder (names, shapes,
    total) and on the training keys the step reads."""
    import json
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.utils import read_config
    a = read_config(os.path.join(ROOT, "config", "vae_dente_no_adv.json"))
    b = read_config(os.path.join(ROOT, "config", "vae_both_no_adv.json"))
    assert a["autoencoder_def"] == b["autoencoder_def"]
    ma, mb = VAEModel.from_config(a["autoencoder_def"]), VAEModel.from_config(b["autoencoder_def"])
    sa, sb = ma.state_dict(), mb.state_dict()
    assert list(sa) == list(sb) and all(sa[k].shape == sb[k].shape for k in sa)
    assert sum(v.numel() for v in sb.values()) == 4_562_593
    for k in ("batch_size", "patch_size", "lr", "kl_weight", "recon_loss"):
        assert a["autoencoder_train"][k] == b["autoencoder_train"][k], k
    # (the files ship batch_size 8; BASELINE's 8 x 32 = 256 is the bench workload, set with bench.py --batch)
    assert list(b["autoencoder_train"]["patch_size"]) == [256, 256]
