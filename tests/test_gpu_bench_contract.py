"""The driver's bench.py contract, checked in-process on a tiny workload (64x64, batch 2): ONE JSON line with the agreed
keys -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config{workload,...} -- plus ``roofline`` {bound, achieved, peak, unit, frac, traffic} for the dominant kernel
and ``cpu_baseline`` {value, unit, cores, kind, sample}."""
import json
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run(monkeypatch, capsys, extra):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "64"] + extra)
    bench.main()
    lines = [l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys(dev, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--cpu-steps", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "vae_train_images_per_sec_256x256_bf16" and d["unit"] == "images/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 <= r["frac"] <= 1.0
    assert r["achieved"] == pytest.approx(r["frac"] * r["peak"], rel=1e-2, abs=1e-4 * r["peak"])   # frac is rounded to 4 places
    assert r["traffic"] is None          # the committed counter passes belong to the batch-32 256x256 workload, not this one
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "images/s" and "sample" in c
    assert isinstance(d["per_shape"], list) and d["per_shape"]


def test_adversarial_and_regression_lines(dev, monkeypatch, capsys):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _run(monkeypatch, capsys, ["--no-cpu-baseline", "--adv"])
    assert "LSGAN(PatchDiscriminator)" in d["config"]["workload"] and d["value"] > 0
    d = _run(monkeypatch, capsys, ["--no-cpu-baseline", "--config", os.path.join(root, "config", "reg_edente_from_dente.json")])
    assert d["metric"] == "vae_encoder_regression_images_per_sec_256x256_bf16" and d["value"] > 0
