#!/usr/bin/env python3
"""Decode-step time for a list of environment variants, alternating, in ONE process on ONE box (the engine reads its
switches at creation, so every variant gets a fresh Model), e.g. "stack:" "per_layer:VOX_HIP_DISABLE=stack".
usage: decode_ab.py [--reps R] [--iters N] [--kv a,b,..] [--profile] name:ENV=v,ENV2=w ...   ('base:' = no switches)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir

args = sys.argv[1:]
reps, iters, kvs, prof = 2, 60, [232, 1900], False
while args and args[0].startswith("--"):
    o = args.pop(0)
    if o == "--reps": reps = int(args.pop(0))
    elif o == "--iters": iters = int(args.pop(0))
    elif o == "--kv": kvs = [int(x) for x in args.pop(0).split(",")]
    elif o == "--profile": prof = True
variants = []
for a in args:
    name, _, envs = a.partition(":")
    variants.append((name, dict(kv.split("=", 1) for kv in envs.split(";") if kv)))
d = model_dir(os.environ.get("PRESET", "full"))
v.hip.vox_hip_profile_decode.restype = C.c_double
v.hip.vox_hip_profile_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
touched = sorted({k for _, e in variants for k in e})
res = {}
for rep in range(reps):
    for name, env in variants:
        for k in touched: os.environ.pop(k, None)
        os.environ.update(env)
        with v.Model(d, weights=os.environ.get('SWEEP_WEIGHTS', 'bf16')) as m:
            row = []
            for kv in kvs:
                m.time_decoder_step(5, kv)
                row.append(round(m.time_decoder_step(iters, kv) * 1e3, 4))
            extra = ""
            if prof:
                avg = (C.c_double * 16)(); cnt = (C.c_int * 16)()
                v.hip.vox_hip_profile_decode(m.engine, 20, kvs[0], avg, cnt)
                extra = "  per-kernel us (qkv/fused, swiglu, w2, logits): %.2f %.2f %.2f %.1f" % (avg[1], avg[5], avg[6], avg[7])
        res.setdefault(name, []).append(row)
        print(f"rep {rep} {name:14s} {row}{extra}", flush=True)
print("== best of reps, ms/step at kv", kvs)
base = None
for name, _ in variants:
    best = [min(r[i] for r in res[name]) for i in range(len(kvs))]
    if base is None: base = best
    print(f"{name:14s} {best}  vs first: {[round(b - a, 4) for a, b in zip(base, best)]}")
