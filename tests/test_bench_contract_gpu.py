"""-m gpu: the one JSON line of bench.py carries every field the driver's contract names — on a small batch (the contract, not the numbers)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    d = _line("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--cpu-images", "1", "--secondary-steps", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["dtype"] == "f16x2" and d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 0.05 * d["value"]
    cfg = d["config"]
    for k in ("workload", "per_gpu_batch", "global_batch", "glyphs_per_image", "parallelism", "precision_mode", "need_prior_image", "prior_image_precision",
              "gflop_per_image", "gflop_per_image_by_arithmetic", "weights"):
        assert k in cfg, k
    assert 0.5 < cfg["hbm_peak_gb"] < cfg["hbm_capacity_gb"] == 288     # what the timed workload holds in HBM (batch 8 here: a few GB)
    assert "model" not in cfg and len(cfg["workload"]) <= 128          # (the driver's parsed record cuts strings at 128 characters)
    # round 6: the two neighbours of `value` at the top level (no generator level in a cheaper arithmetic / no structure image at all)
    # (batch 8 with ONE secondary step is a 35 ms sample timed from the host: a scheduling hiccup halves it — the check is that the figure is there and plausible;
    #  the full-size relation, 281 vs 256 images/s, is in DESIGN.md §6)
    assert 0 < d["value_all_levels_in_mode_precision"] and d["value_without_prior_image"] >= 0.4 * d["value"] and len(cfg["value_is"]) <= 128
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "hbm_tail_ms_per_step", "all_conv_achieved"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "threads", "host_cores", "cpu_model"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == c["threads"] <= c["host_cores"]
    p = d["parity"]
    assert p["sr_max_abs_fp16x2"] <= 1e-3 and p["argmax_match_fp16x2"] == 1.0 and p["sr_max_abs_fp16x2_timed_batch"] <= 1e-3
    assert p["regime_trained_like"]["sr_max_abs_fp16x2"] <= 1e-3 and p["regime_trained_like"]["argmax_match_fp16x2"] == 1.0
    s = d["secondary"]
    assert set(s["configs"]) == {"configs1_batch64", "configs3_gan_only", "configs4_mixed_widths"}
    for v in s["configs"].values():
        assert v["value"] > 0 and 0 < v["frac_of_2500"] < 1


@pytest.mark.parametrize("config", ["gan", "mixed"])
def test_other_config_lines(config):
    d = _line("--config", config, "--steps", "1", "--warmup", "1", "--batch", "8")
    assert d["value"] > 0 and "roofline" in d and "cpu_baseline" in d and "parity" in d and d["config"]["workload"].startswith("configs[")
