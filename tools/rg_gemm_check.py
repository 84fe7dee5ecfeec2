#!/usr/bin/env python3
"""k_rowsgemm against the scalar reference kernel through vox_hip_linear_bf16 (impl 4: planes, 5: f32 rows vs impl 3)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
from oracle import vox_oracle as vo
m = v.Model(model_dir("tiny"), enc_window=48, dec_window=64)
rng = np.random.default_rng(1)
for (M, K, N) in [(8, 1280, 256), (16, 1280, 256), (17, 1280, 256), (25, 1280, 6144), (32, 2048, 1280), (33, 1280, 256), (38, 3072, 6144),
                  (64, 4096, 3072), (68, 5120, 1280), (100, 1280, 10240), (128, 1280, 1280), (38, 9216, 3072)]:
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = vo.f32_to_bf16((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(N).astype(np.float32)
    ref = m.linear_bf16(x, w, b, impl=3)
    for impl in (4, 5):
        y = m.linear_bf16(x, w, b, impl=impl)
        err = np.abs(y - ref).max(axis=1)
        bad = np.nonzero(err > 1e-4)[0]
        print(f"M={M:4d} K={K:5d} N={N:6d} impl {impl}: max err {err.max():.3e}" + (f"  BAD ROWS {bad.tolist()[:40]}" if len(bad) else ""))
m.close()
