#!/usr/bin/env python3
"""GPU box: two small-preset models decoding concurrently from two threads; per-run comparison with the sequential ids."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from audio_util import synth_speech
from conftest import model_dir
a1, a2 = synth_speech(20.0, 301), synth_speech(20.0, 302)
with v.Model(model_dir("small")) as m:
    want1, want2 = m.transcribe(a1)["tokens"], m.transcribe(a2)["tokens"]
with v.Model(model_dir("small")) as m1, v.Model(model_dir("small")) as m2:
    out = {}
    def run(model, audio, key):
        out[key] = [(model.transcribe(audio)["tokens"], "dec_fused" in model.active_paths()[1]) for _ in range(4)]
    th = [threading.Thread(target=run, args=(m1, a1, 1)), threading.Thread(target=run, args=(m2, a2, 2))]
    [t.start() for t in th]; [t.join() for t in th]
for key, want in ((1, want1), (2, want2)):
    for i, (t, fused) in enumerate(out[key]):
        n = min(len(t), len(want))
        bad = np.nonzero(t[:n] != want[:n])[0]
        print(f"model {key} run {i}: fused_after={fused} len {len(t)}/{len(want)} first_bad {bad[0] if len(bad) else None} n_bad {len(bad)}")
