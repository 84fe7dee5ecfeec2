#!/usr/bin/env python3
"""Numerical A/B of the few-rows paths: the same encoder chunks / decoder prefills on two engines of one process that differ
only in an environment switch (read at engine creation).  Prints max |a - b| / max |b| per case.
usage: rg_check.py [preset]   (env A = VOX_HIP_NO_ROWSGEMM=1, env B = rows-gemm path incl. <= 32 rows)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v  # noqa: E402
from conftest import model_dir  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "small"
os.environ["VOX_HIP_NO_ROWSGEMM"] = "1"
ma = v.Model(model_dir(preset))
del os.environ["VOX_HIP_NO_ROWSGEMM"]
os.environ["VOX_HIP_RG_SMALL"] = "1"
extra = dict(kv.split("=") for kv in sys.argv[2:])
os.environ.update(extra)
mb = v.Model(model_dir(preset))
del os.environ["VOX_HIP_RG_SMALL"]
for k in extra:
    del os.environ[k]
d = ma.dims
rng = np.random.default_rng(3)
worst = 0.0
for sizes in ([25, 25, 25], [1], [7], [8], [9], [16], [17], [32], [33], [38, 64, 68], [96, 100, 128, 3], [800, 25, 68]):
    outs = []
    for m in (ma, mb):
        m.reset_encoder()
        m.reset_counters()
        rr = np.random.default_rng(sum(sizes))
        o = []
        for n in sizes:
            x = rr.standard_normal((n, d.enc_dim)).astype(np.float32)
            o.append(m.encoder_forward_incremental(x))
        outs.append(o)
    for n, a, b in zip(sizes, outs[0], outs[1]):
        e = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-30))
        worst = max(worst, e)
        print(f"encoder chunk n={n:4d} (after {sizes}): rel err {e:.3e}")
for n in (1, 2, 31, 38, 64, 65, 100, 128):
    res = []
    for m in (ma, mb):
        m.reset_counters()
        rr = np.random.default_rng(100 + n)
        emb = (rr.standard_normal((n + 3, d.dec_dim)) * 0.5).astype(np.float32)
        m.decoder_prefill(emb[:n])
        lg = [m.decoder_forward(emb[n + i])[1] for i in range(3)]
        res.append(np.stack(lg))
    e = float(np.abs(res[0] - res[1]).max())
    worst = max(worst, e / (np.abs(res[0]).max() + 1e-30))
    print(f"decoder prefill n={n:4d} + 3 steps: max |logit diff| {e:.3e} (logit scale {np.abs(res[0]).max():.2f})")
print("WORST", worst)
ma.close(); mb.close()
