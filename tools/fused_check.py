#!/usr/bin/env python3
"""GPU box: quick A/B of the fused decode step (k_dec_attn_fused + k_gemv_w13x) against the launch-per-GEMV
chain (VOX_HIP_NO_FUSED=1): ids / logits on a short clip, then seconds per decode step at several KV lengths."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v            # noqa: E402
from audio_util import synth_speech   # noqa: E402
from conftest import model_dir        # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "small"
audio = synth_speech(float(sys.argv[2]) if len(sys.argv) > 2 else 20.0, 5)
res = {}
for mode in ("fused", "chain"):
    if mode == "chain":
        os.environ["VOX_HIP_NO_FUSED"] = "1"
    with v.Model(model_dir(preset)) as m:
        print(mode, "paths:", m.active_paths()[1], flush=True)
        r = m.transcribe(audio, record_logits=400)
        r2 = m.transcribe(audio)
        print(mode, "paths after:", m.active_paths()[1], "steps", len(r["tokens"]), "batched == stepwise:",
              bool(np.array_equal(r["tokens"], r2["tokens"])), flush=True)
        t = {kv: round(m.time_decoder_step(30, kv) * 1e3, 4) for kv in (64, 232, 1000, 3000, 8000)}
        print(mode, "ms/step by kv_len:", t, flush=True)
        res[mode] = r
a, c = res["fused"], res["chain"]
n = min(len(a["tokens"]), len(c["tokens"]))
neq = np.nonzero(a["tokens"][:n] != c["tokens"][:n])[0]
first = int(neq[0]) if len(neq) else n
print("steps", n, "first id difference", first, "max |dlogit| up to there",
      float(np.abs(a["logits"][:first + 1] - c["logits"][:first + 1]).max()))
