#!/usr/bin/env python3
"""Workload for per-kernel traces of the paths around the decode loop, on the `small` preset (2 + 2 layers at the real 4B
per-layer shapes, loads in a second): N s of audio as a stream (0.5 s feeds, -I 0.5, continuous: 25-row encoder chunks, bursts
of 6-7 decoder steps) and the same audio as one batch (big encoder pass, 38-row prefill, 68-row flush pass).
usage: enc_paths_probe.py [seconds] [stream|batch|both]      (run under rocprofv3 --kernel-trace; tools/trace_summary.py)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v  # noqa: E402
from audio_util import synth_speech  # noqa: E402
from conftest import model_dir  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
what = sys.argv[2] if len(sys.argv) > 2 else "both"
preset = os.environ.get("VOX_PROBE_PRESET", "small")
audio = synth_speech(secs, 5)
with v.Model(model_dir(preset)) as m:
    ids = {}
    if what in ("stream", "both"):
        t0 = time.time()
        r = m.transcribe(audio, feed_sizes=[8000] * (len(audio) // 8000 + 1), interval=0.5, continuous=True)
        ids["stream"] = r["tokens"]
        print(f"stream: {len(r['tokens'])} steps in {time.time() - t0:.3f} s")
    if what in ("batch", "both"):
        for _ in range(3):
            t0 = time.time()
            r = m.transcribe(audio)
            t = m.timing()
            print(f"batch: {len(r['tokens'])} steps in {time.time() - t0:.3f} s; encode {t['encode_ms']:.2f} ms prefill {t['prefill_ms']:.2f} ms "
                  f"decode {t['decode_ms'] / max(t['decode_steps'], 1):.4f} ms/step")
        ids["batch"] = r["tokens"]
    out = os.environ.get("VOX_PROBE_IDS")
    if out:
        np.savez(out, **ids)
