#!/usr/bin/env python3
"""Decode-step time at several KV lengths (vox_hip_time_decoder_step: HIP events around N steps on resident weights).
usage: dec_step_probe.py [preset] [iters] [kv,kv,..]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
preset = sys.argv[1] if len(sys.argv) > 1 else "full"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with v.Model(model_dir(preset)) as m:
    out = []
    kvs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (232, 600, 1900, 3800, 8000)
    for kv in kvs:
        m.time_decoder_step(5, kv)
        out.append((kv, round(m.time_decoder_step(iters, kv) * 1e3, 4)))
    print(os.environ.get("TAG", ""), "ms/step by kv:", out)
