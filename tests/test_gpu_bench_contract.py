"""GPU: bench.py keeps the JSON contract (one line; metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload and the
roofline object measured live).  Small preset so that it runs in seconds; the full-size line with
the CPU baseline is what the driver runs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "small", "--seconds", "12",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_headline_line():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "decode_tok_s", "parity", "active_paths"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is False and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert 0 < d["value"] < 1 and abs(d["value"] * 12.0 * 1e3 - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0


def test_stream_mode_line():
    d = _run("--mode", "stream")
    assert d["n_gpus"] == 1 and d["higher_is_better"] is False and "chunk_latency_ms" in d
    assert d["chunk_latency_ms"]["p99"] >= d["chunk_latency_ms"]["p50"] > 0
