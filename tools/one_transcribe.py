#!/usr/bin/env python3
"""One transcription of the headline clip (the golden's stored 30 s of audio) on the full-size model, for profiling."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
g = np.load(os.path.join(ROOT, "tests", "golden", "stream_full_batch.npz"))
audio = (g["audio_i16"].astype(np.float32) / 32768.0) if "audio_i16" in g.files else g["audio"]
with v.Model(model_dir("full")) as m:
    r = m.transcribe(audio)
    print("tokens", len(r["tokens"]))
