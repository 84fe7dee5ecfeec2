#!/usr/bin/env python3
"""Golden for tests/test_gpu_cli.py: the reference's own CLI binary (oracle/_ref/voxtral_ref_full =
main.c + the reference sources, `make blas` flags) transcribes a seeded synthetic clip with the
full-size synthetic checkpoint on the CPU; stdout and the stat lines are stored in
tests/golden/cli_full.json.  Runs only where /root/reference was available to build the binary
(minutes of CPU time)."""
import json
import os
import subprocess
import sys
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from audio_util import synth_speech   # noqa: E402
from conftest import model_dir        # noqa: E402

SECONDS, SEED = 4.0, 2718
REF = os.path.join(ROOT, "oracle", "_ref", "voxtral_ref_full")


def write_wav(path, samples):
    pcm = np.clip(np.round(samples * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())


def main():
    clip = "/tmp/cli_golden.wav"
    write_wav(clip, synth_speech(SECONDS, SEED))
    t0 = time.time()
    r = subprocess.run([REF, "-d", model_dir("full"), "-i", clip], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {"seconds": SECONDS, "seed": SEED, "stdout": r.stdout,
           "stderr_stats": [ln for ln in r.stderr.splitlines() if ln.startswith(("Audio:", "Encoder:", "Decoder:"))],
           "generator": "tools/make_cli_golden.py (oracle/_ref/voxtral_ref_full, reference sources, -O3 -ffast-math, OpenBLAS)",
           "cpu_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(ROOT, "tests", "golden", "cli_full.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1500])

    # --alt 0.9: alternatives within the softmax cutoff are printed in brackets (main.c:46-80)
    t0 = time.time()
    r = subprocess.run([REF, "-d", model_dir("full"), "-i", clip, "--alt", "0.9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {"seconds": SECONDS, "seed": SEED, "stdout": r.stdout, "generator": "tools/make_cli_golden.py (--alt 0.9)",
           "cpu_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(ROOT, "tests", "golden", "cli_full_alt.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:800])

    # the same clip as a raw s16le stream on stdin, processing interval 0.5 s (main.c: 4096-sample
    # reads, continuous mode)
    with wave.open(clip, "rb") as w:
        pcm = w.readframes(w.getnframes())
    t0 = time.time()
    r = subprocess.run([REF, "-d", model_dir("full"), "--stdin", "-I", "0.5"], input=pcm, capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    err = r.stderr.decode()
    out = {"seconds": SECONDS, "seed": SEED, "stdout": r.stdout.decode(),
           "stderr_stats": [ln for ln in err.splitlines() if ln.startswith(("Audio:", "Encoder:", "Decoder:"))],
           "generator": "tools/make_cli_golden.py (--stdin -I 0.5, raw s16le)", "cpu_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(ROOT, "tests", "golden", "cli_full_stdin.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
