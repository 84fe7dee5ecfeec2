#!/usr/bin/env python3
"""Decode-step time at the KV lengths of the 30 s clip (one to eight key slices).  usage: dec_step_probe2.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with v.Model(model_dir("full")) as m:
    out = []
    for kv in (60, 120, 190, 250, 316, 380, 440, 508):
        m.time_decoder_step(5, kv)
        out.append((kv, round(m.time_decoder_step(iters, kv) * 1e3, 4)))
    print(os.environ.get("TAG", ""), "ms/step by kv:", out)
