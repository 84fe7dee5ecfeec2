#!/usr/bin/env python3
"""Small encoder-only workload for rocprofv3 --pmc passes (MFMA utilisation of the GEMM and
attention kernels): two 1664-row chunks through the 32 encoder layers + adapter.  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
with v.Model(model_dir("full")) as m:
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1664, m.dims.enc_dim)) * 0.5).astype(np.float32)
    m.reset_encoder()
    y = m.encoder_forward_incremental(x)
    y = m.encoder_forward_incremental(x)
    print("encoder out", y.shape, float(np.abs(y).mean()))
