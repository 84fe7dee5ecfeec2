#!/usr/bin/env python3
"""Tiny workload for the rocprofv3 --pmc pass (counter collection serialises every dispatch, so
the full bench is far too long): load the full-size synthetic model, prefill a short prompt and
run a handful of decoder steps at a KV length typical for the 30 s clip.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v            # noqa: E402
from conftest import model_dir       # noqa: E402

with v.Model(model_dir("full")) as m:
    s = m.time_decoder_step(int(sys.argv[1]) if len(sys.argv) > 1 else 4, 232)
    print("decoder step: %.3f ms" % (s * 1e3))
