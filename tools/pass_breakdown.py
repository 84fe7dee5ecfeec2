#!/usr/bin/env python3
"""Where does one headline pass (30 s clip) spend its wall time?  GPU box only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from audio_util import synth_speech
from conftest import model_dir
audio = synth_speech(30.0, 1234)
with v.Model(model_dir("full")) as m:
    m.transcribe(audio)
    for rep in range(2):
        v.hip.vox_hip_sync(m.engine)
        t0 = time.time(); s = v.Stream(m); t1 = time.time()
        s.feed(audio); t2 = time.time()
        p1 = s.get(); t3 = time.time()
        s.finish(); t4 = time.time()
        p2 = s.get(); t5 = time.time()
        toks = s.token_ids(); s.free(); t6 = time.time()
        tm = m.timing()
        print(f"init {1e3*(t1-t0):.2f}  feed {1e3*(t2-t1):.2f}  get {1e3*(t3-t2):.2f}  finish {1e3*(t4-t3):.2f}  get {1e3*(t5-t4):.2f}  free {1e3*(t6-t5):.2f}  total {1e3*(t6-t0):.2f} ms | engine: {tm}")
