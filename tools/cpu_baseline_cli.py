#!/usr/bin/env python3
"""The reference CPU path timed END TO END on this host.

Runs the UNMODIFIED reference CLI (oracle/_ref/voxtral_ref_full = main.c + the reference sources with the
`make blas` flags, built by oracle/Makefile) on the headline input - the 30 s night1968 clip carried by
tests/golden/stream_full_batch.npz - with the full-size synthetic checkpoint, parses the stat lines the
reference prints (main.c:390, voxtral.c:1306-1318) and writes a JSON record with the CPU model and
core / thread counts.  Takes minutes; run once per round on the GPU box's host and commit the result as
profiles/rNN_cpu_baseline_cli.json (bench.py reports it as cpu_baseline.measured_end_to_end).
Test infrastructure: this is the baseline, never the product.

usage: python tools/cpu_baseline_cli.py [out.json] [--threads N]
"""
import json
import os
import re
import subprocess
import sys
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import model_dir        # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "voxtral_ref_full")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "cpu_baseline_cli.json")
    threads = None
    if "--threads" in sys.argv:
        threads = int(sys.argv[sys.argv.index("--threads") + 1])
    g = np.load(os.path.join(ROOT, "tests", "golden", "stream_full_batch.npz"), allow_pickle=True)
    pcm = g["audio_i16"].astype("<i2")
    clip = "/tmp/vox_headline_30s.wav"
    with wave.open(clip, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    env = dict(os.environ)
    if threads:
        env["OPENBLAS_NUM_THREADS"] = str(threads)
    mdir = model_dir("full")
    t0 = time.time()
    r = subprocess.run([REF, "-d", mdir, "-i", clip], capture_output=True, text=True, env=env)
    wall = time.time() - t0
    stats = [ln for ln in r.stderr.splitlines() if ln.startswith(("Audio:", "Encoder:", "Decoder:", "Model", "Loading"))]
    rec = {"command": "oracle/_ref/voxtral_ref_full -d <full synthetic checkpoint> -i <30 s night1968 clip>",
           "returncode": r.returncode, "process_wall_s": round(wall, 2), "stderr_stats": stats,
           "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(),
           "openblas_threads": threads or "default (all cores; only the M>1 sgemm calls are threaded)",
           "note": "the reference's decode GEMV (bf16_matvec_fused) and its attention are single-threaded by construction"}
    m = re.search(r"Encoder:\s+\d+ mel -> \d+ tokens \((\d+) ms\)", r.stderr)
    d = re.search(r"Decoder:\s+(\d+) text tokens \((\d+) steps\) in (\d+) ms \(prefill (\d+) ms \+ ([0-9.]+) ms/step\)", r.stderr)
    if m and d:
        enc_ms, dec_ms = float(m.group(1)), float(d.group(3))
        rec.update({"audio_seconds": len(pcm) / 16000.0, "encoder_ms": enc_ms, "decoder_ms": dec_ms,
                    "prefill_ms": float(d.group(4)), "ms_per_step": float(d.group(5)), "steps": int(d.group(2)),
                    "rtf_stream": round((enc_ms + dec_ms) / 1e3 / (len(pcm) / 16000.0), 3),
                    "rtf_process": round(wall / (len(pcm) / 16000.0), 3),
                    "decode_tok_s": round(1e3 / float(d.group(5)), 3)})
    # the reference binary's transcript on this input must be the golden's pieces (the same run, through main.c)
    want = "".join(str(p) for p in g["pieces"]).strip()
    rec["stdout_matches_golden_pieces"] = (r.stdout.strip() == want)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
