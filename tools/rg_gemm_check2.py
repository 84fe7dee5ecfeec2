import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
from oracle import vox_oracle as vo
m = v.Model(model_dir("tiny"), enc_window=48, dec_window=64)
rng = np.random.default_rng(1)
for (M, K, N) in [(32, 1280, 512), (32, 1280, 1024), (32, 128, 6144), (32, 1280, 6144)]:
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = vo.f32_to_bf16((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    ref = m.linear_bf16(x, w, None, impl=3)
    for impl in (4, 5):
        for rep in range(2):
            y = m.linear_bf16(x, w, None, impl=impl)
            bad = np.abs(y - ref) > 1e-4
            rows = np.nonzero(bad.any(axis=1))[0]
            cols = np.nonzero(bad.any(axis=0))[0]
            blocks = sorted(set((cols // 32).tolist()))
            print(f"M={M} K={K} N={N} impl {impl} rep {rep}: bad elems {int(bad.sum())}, rows {rows.tolist()}, 32-col tiles {blocks[:24]}{'...' if len(blocks) > 24 else ''} ({len(blocks)} tiles)")
            if bad.sum():
                r, c = np.argwhere(bad)[0]
                print("   first bad", r, c, y[r, c], ref[r, c], " cols bad in that row:", np.nonzero(bad[r])[0][:16].tolist())
m.close()
