"""Shared fixtures.  GPU tests are marked `gpu`; everything else must pass on a CPU-only box."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_ROOT = os.environ.get("VOX_TEST_MODELS", "/tmp/vox_models")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def synth_model_bin():
    exe = os.path.join(ROOT, "build", "synth_model")
    src = os.path.join(ROOT, "tools", "synth_model.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-o", exe, src, "-lm", "-lpthread"])
    return exe


def model_dir(preset, seed=1234):
    """Deterministic synthetic checkpoint for `preset` (generated on first use)."""
    d = os.path.join(MODEL_ROOT, f"{preset}_{seed}")
    if not os.path.exists(os.path.join(d, "tekken.json")):
        # several processes may ask at once (the ranks of `bench.py --gpus N` on a fresh box): one generates, the others wait
        import fcntl
        os.makedirs(MODEL_ROOT, exist_ok=True)
        with open(os.path.join(MODEL_ROOT, f".{preset}_{seed}.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if not os.path.exists(os.path.join(d, "tekken.json")):      # tekken.json is written last (tools/synth_model.c)
                    subprocess.check_call([synth_model_bin(), d, preset, str(seed)])
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return d


@pytest.fixture(scope="session")
def tiny_dir():
    return model_dir("tiny")


@pytest.fixture(scope="session")
def small_dir():
    return model_dir("small")


def have_ref(variant):
    from oracle.ref_binding import ref_available
    return ref_available(variant)


@pytest.fixture(scope="session")
def ref_tiny():
    if not have_ref("tiny"):
        pytest.skip("oracle/_ref/libvoxref_tiny.so not built")
    from oracle.ref_binding import RefLib
    return RefLib("tiny")


@pytest.fixture(scope="session")
def ref_small():
    if not have_ref("small"):
        pytest.skip("oracle/_ref/libvoxref_small.so not built")
    from oracle.ref_binding import RefLib
    return RefLib("small")


def gpu_available():
    try:
        import voxtral_c_amd as v
        return v.device_count() > 0
    except Exception:
        return False
