"""CPU tests of the regression loop's host side (reference
src/pti_ldm_vae/utils/regression_utils.py, utils/metrics.py): config normalisation, target normaliser, loss factory,
metrics (known answer KA6 of SURVEY.md 8c, captured from the reference file), the epoch loop on a stand-in model, and the
head checkpoint formats."""
import json

import pytest
import torch
from torch import nn

from pti_ldm_vae_amd.utils import regression_utils as R
from pti_ldm_vae_amd.utils.metrics import compute_regression_metrics


def test_metrics_known_answer_from_reference():
    m = compute_regression_metrics(torch.tensor([[1.0, 2.0], [3.0, 5.0]]), torch.tensor([[1.0, 1.0], [2.0, 2.0]]), ["a", "b"])
    assert m == pytest.approx({"mae": 1.25, "mse": 2.75, "mae_a": 0.5, "mse_a": 0.5, "mae_b": 2.0, "mse_b": 5.0})
    with pytest.raises(ValueError, match="Shape mismatch"):
        compute_regression_metrics(torch.zeros(2, 2), torch.zeros(2, 3), ["a", "b"])


def test_config_extraction_both_schemas():
    import os
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "reg_edente_from_dente.json")))
    d, t, r = R.extract_regression_data_config(cfg), R.extract_regression_train_config(cfg), R.extract_regressor_def_config(cfg)
    assert d["patch_size"] == [256, 256] and d["data_source"] == "dente" and d["num_workers"] == 4
    assert t["batch_size"] == 8 and t["lr"] == 1e-4 and t["loss"] == "mse" and t["weight_decay"] == 0.0
    assert r == {"hidden_dims": [256, 32], "dropout": 0.1, "activation": "relu"}
    legacy = {"data_base_dir": "/d", "attributes_path": "/a.json", "patch_size": [64, 64],
              "train": {"batch_size": 2, "lr": 1e-3, "max_epochs": 1, "num_workers": 2, "seed": 7}, "regressor": {}}
    d2 = R.extract_regression_data_config(legacy)
    assert d2["num_workers"] == 2 and d2["seed"] == 7 and d2["data_source"] == "edente" and d2["train_split"] == 0.9
    assert R.extract_regression_train_config(legacy)["val_interval"] == 1
    assert R.extract_regressor_def_config(legacy) == {"hidden_dims": [], "dropout": 0.0, "activation": "relu"}
    assert R.extract_regression_eval_config(legacy)["patch_size"] == [64, 64]
    with pytest.raises(KeyError, match="Missing required data config fields"):
        R.extract_regression_data_config({"data": {"patch_size": [1, 1]}})
    with pytest.raises(KeyError, match="Missing required training config fields"):
        R.extract_regression_train_config({"train": {"lr": 1.0}})
    assert R.regression_loss_key({"loss": "huber"}) == "loss_huber" and R.regression_loss_key({}) == "loss_mse"


def test_target_normalizer_roundtrip_and_files(tmp_path):
    t = torch.tensor([[1.0, 5.0, 2.0], [3.0, 5.0, 4.0], [5.0, 5.0, 9.0]])
    nz = R.compute_target_normalizer(t)
    assert torch.allclose(nz.mean, t.mean(0)) and nz.std[1] == 1.0          # zero std -> 1
    assert torch.allclose(nz.std[0], t[:, 0].std(unbiased=False))
    z = nz.normalize(t)
    assert torch.allclose(z.mean(0), torch.zeros(3), atol=1e-6) and torch.allclose(nz.denormalize(z), t, atol=1e-6)
    R.save_target_normalizer(tmp_path / "w" / "stats.json", nz, ["a", "b", "c"])
    nz2 = R.load_target_normalizer(tmp_path / "w" / "stats.json", ["a", "b", "c"])
    assert torch.allclose(nz2.mean, nz.mean) and torch.allclose(nz2.std, nz.std)
    with pytest.raises(ValueError, match="Target order mismatch"):
        R.load_target_normalizer(tmp_path / "w" / "stats.json", ["c", "b", "a"])
    with pytest.raises(ValueError, match="same shape"):
        R.TargetNormalizer(torch.zeros(2), torch.zeros(3))
    assert isinstance(R.build_loss_fn("MSE"), nn.MSELoss) and isinstance(R.build_loss_fn("huber"), nn.SmoothL1Loss)
    with pytest.raises(ValueError, match="Unsupported loss"):
        R.build_loss_fn("l1")


class _Stand(nn.Module):
    """Stand-in for VAELatentRegressor on CPU: a fixed 'encoder' + a trainable head under the same attribute names."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.regressor = nn.Linear(16, 2)
        self.latent_dim = 16

    def forward(self, images):
        return self.regressor(images.flatten(1))


def test_epoch_loop_and_checkpoints_on_a_stand_in_model(tmp_path):
    torch.manual_seed(1)
    w = torch.randn(16, 2)
    xs = [torch.randn(4, 1, 4, 4) for _ in range(6)]
    data = [(x, x.flatten(1) @ w + 3.0) for x in xs]
    model = _Stand()
    nz = R.compute_target_normalizer(torch.cat([t for _, t in data]))
    opt = torch.optim.Adam(model.parameters(), lr=5e-2)
    loss_fn = R.build_loss_fn("mse")
    dev = torch.device("cpu")
    l0 = R.train_one_epoch(model, data, opt, loss_fn, dev, nz)
    for _ in range(30):
        l1 = R.train_one_epoch(model, data, opt, loss_fn, dev, nz)
    assert l1 < 0.2 * l0
    vl, metrics = R.validate_one_epoch(model, data[:2], loss_fn, dev, ["a", "b"], nz)
    assert set(metrics) == {"mae", "mse", "mae_a", "mse_a", "mae_b", "mse_b"} and vl < l0
    # metrics are on DE-normalised predictions vs raw targets
    with torch.no_grad():
        pred = torch.cat([nz.denormalize(model(x)) for x, _ in data[:2]])
    assert metrics["mse"] == pytest.approx(float(((pred - torch.cat([t for _, t in data[:2]])) ** 2).mean()), rel=1e-5)
    with pytest.raises(RuntimeError, match="zero batches"):
        R.train_one_epoch(model, [], opt, loss_fn, dev, None)
    with pytest.raises(RuntimeError, match="zero batches"):
        R.validate_one_epoch(model, [], loss_fn, dev, ["a", "b"], None)
    wd = tmp_path / "weights"
    last = R.save_last_regression_checkpoint(wd, model, ["a", "b"], 3)
    assert last.name == "head_last.pth"
    ck = torch.load(last, weights_only=True)
    assert set(ck) == {"regressor_state_dict", "targets", "epoch", "latent_dim"} and ck["epoch"] == 3 and ck["latent_dim"] == 16
    best, path = R.maybe_save_best_regression_checkpoint(wd, model, ["a", "b"], 4, 0.5, float("inf"))
    assert best == 0.5 and path.name == "head_best.pth" and path.exists()
    best2, _ = R.maybe_save_best_regression_checkpoint(wd, model, ["a", "b"], 5, 0.7, best, path)
    assert best2 == 0.5 and torch.load(path, weights_only=True)["epoch"] == 4          # not overwritten
    m2 = _Stand()
    with torch.no_grad():
        m2.regressor.weight.zero_()
    meta = R.load_regression_checkpoint(last, m2, ["a", "b"])
    assert meta["epoch"] == 3 and torch.equal(m2.regressor.weight, model.regressor.weight)
    with pytest.raises(ValueError, match="Target mismatch"):
        R.load_regression_checkpoint(last, m2, ["b", "a"])
