"""The line bench.py prints on one GPU: every field the round's driver reads, with values that hang together.

The CPU oracle runs inside bench.py's cpu_baseline leg only (as the checker of the frame and as the timed baseline)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--warmup-ms", "5",
           "--cpu-baseline-seconds", "1.5"] + list(flags)
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints exactly one line"
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    d = _bench()
    assert d["metric"].startswith("Mrays/s") and d["unit"] == "Mrays/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    cfg = d["config"]
    assert cfg["workload"] == "bunny_1080p_primary_lambert_invert" and "model" not in cfg
    assert (cfg["width"], cfg["height"], cfg["spp"], cfg["ao_rays"]) == (1920, 1080, 1, 0)
    assert cfg["triangles"] == 4968 and cfg["hit_pixels"] == 25139
    # value, ms_per_step and us_per_frame are one measurement
    frames = d["steps"] * cfg["frames_per_step"]
    assert d["value"] == pytest.approx(cfg["rays_per_frame"] * frames / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6, rel=1e-9)
    assert cfg["us_per_frame"] == pytest.approx(d["ms_per_step"] * 1e3 / cfg["frames_per_step"], rel=1e-9)
    # every frame of the timed region went to attachments beyond the Infinity Cache
    assert cfg["attachment_mb"] > 4 * 256
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12) and 0.0 < r["frac"] < 1.0
    assert r["kernel"] == "k_render_tile_batch" and r["frames_per_launch"] == 4
    assert r["algorithmic_bytes"] == 4 * (12 * 1920 * 1080 + cfg["bytes_read_per_frame"])
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes"] / 4 * frames / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e9, rel=1e-9)
    assert r["traffic"] is None or abs(r["traffic"] / r["algorithmic_bytes"] - 1.0) < 0.05  # PMC bytes of the committed pass
    assert r["launch"]["kernel_ms"] > 0 and r["alone"]["kernel_ms"] > 0
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "Mrays/s" and b["value"] > 0 and 1 <= b["cores"] <= 16
    assert "whole 1920x1080 frames" in b["sample"] and b["outputs_match_gpu"] is True


def test_single_draw_form():
    """--frames-in-flight 1 --frames-per-launch 1: plain inv_render on the context's stream."""
    d = _bench("--frames-in-flight", "1", "--frames-per-launch", "1", "--no-cpu-baseline", "--frames-per-step", "16")
    assert d["cpu_baseline"] is None
    assert d["roofline"]["kernel"] == "k_render_tile" and d["roofline"]["frames_per_launch"] == 1
    assert d["config"]["draw_path"] == "8x8 tiles"
    assert d["roofline"]["launch"]["concurrent_launches"] == 1
