#!/usr/bin/env python3
"""tools/make_golden.py — generate tests/golden/*.npz from the REAL reference.

Runs only where /root/reference has been compiled into oracle/_ref (this container:
`make -C oracle`).  Drives the reference's own stream API (vox_stream_feed/finish, i.e. what
main.c does) on deterministic synthetic checkpoints (tools/synth_model.c) and synthetic
audio (tests/audio_util.py) and stores, per case:
  tokens      every decoder-step token id (captured with --wrap=vox_decoder_forward)
  pieces      the strings vox_stream_get surfaced
  top_ids/top_vals   the 8 largest logits of every step
  logits_head full logits rows for the first 4 steps
  logits_stride / stride_steps   full logits rows of every 16th step (32nd at vocabulary 131072; at most 64 rows): the whole row, not only the top-8
  margin      top1-top2 logit gap per step (how fragile the argmax is)
  n_distinct / margin_hist   distinct greedy ids and a histogram of the margins (edges in margin_edges):
              says how much "identical token ids" means on this checkpoint
  audio_i16   the input itself for the cases that run on the reference's own sample clips
              (30 s = first 480 000 samples of night1968/45s_right_through_the_billboard.wav,
              config 1 = samples/jfk.wav); /root/reference does not exist on the GPU box
  mel_sha/adapter checks are covered by stage-level goldens (stage_*.npz)
The fixtures are small (a few hundred kB) and committed; the generator is committed so they
can be re-derived.  usage: python tools/make_golden.py [--full]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.ref_binding import RefLib  # noqa: E402
from oracle import vox_oracle as vo  # noqa: E402
from audio_util import synth_speech  # noqa: E402
from conftest import model_dir  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CASES = [
    # name, preset, seconds, audio seed, feed sizes (None = one feed), interval, continuous
    ("tiny_batch", "tiny", 12.0, 1, None, None, False),
    ("tiny_stream", "tiny", 12.0, 1, "1s", None, False),
    ("tiny_smallint", "tiny", 12.0, 1, 4096, 0.1, False),
    ("tiny_long", "tiny", 100.0, 2, None, None, False),        # > 1126 steps: decoder window roll-over
    ("tiny_continuous", "tiny", 200.0, 3, 4096, 0.5, True),    # restarts at kv > 2000
    ("small_batch", "small", 8.0, 4, None, None, False),
    ("small_long", "small", 95.0, 6, None, None, False),       # > 1024 decoder positions: split-K attention + combine kernel
    # vox_set_delay (time conditioning a6, prompt length 1+32+delay, right padding): 240 ms and 960 ms
    ("tiny_delay240", "tiny", 10.0, 8, None, None, False, 240),
    ("tiny_delay960", "tiny", 10.0, 8, "1s", None, False, 960),
    ("small_delay160", "small", 8.0, 4, None, None, False, 160),
    # BASELINE config 1 input (samples/jfk.wav, 11 s) at the real per-layer shapes
    ("small_jfk", "small", 0, 0, None, None, False, None, "jfk.wav"),
    # the headline input (30 s of night1968) through the full depth at the tiny widths, and 300 s of
    # synthetic audio through it: 3761 decoder steps, KV to 3799
    ("deep_batch", "deep", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav"),
    ("deep_long", "deep", 300.0, 11, None, None, False),
    # 700 s through the full depth: 8750 decoder steps, past the real decoder window of 8192 (KV compaction in the reference,
    # ring wrap-around on the device)
    ("deep_wrap", "deep", 700.0, 13, None, None, False),
    # 300 s at the real per-layer shapes: decode attention with KV to 3799 (many key slices)
    ("small_xlong", "small", 300.0, 12, None, None, False),
]
# the headline configuration itself: 32 + 26 layers, vocabulary 131072, the 30 s night1968 input
FULL_CASES = [
    ("full_batch", "full", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav"),
    # BASELINE config 1's input at the full geometry, fed as main.c feeds a file: 16000-sample pieces
    ("full_jfk", "full", 0, 0, "1s", None, False, None, "jfk.wav"),
    # BASELINE config 3's feed pattern at the full geometry: 0.5 s feeds, -I 0.5, continuous mode (20 s of the night1968 clip)
    ("full_stream", "full", 20.0, 0, 8000, 0.5, True, None, "benchmark/night1968/45s_right_through_the_billboard.wav"),
]
# BASELINE config 3 through a continuous-mode restart at the full geometry (voxtral.c:378,1137-1187: full stream
# reset once kv_cache_len > 2000, i.e. after ~160 s): the 30 s night1968 clip tiled to 176 s, 0.5 s feeds, -I 0.5.
# ~30 min of reference CPU time on 8 cores; run with --only full_continuous.  The fixture stores the 30 s base clip
# once (audio_i16) and the tiled length (audio_total_samples).
LONG_CASES = [
    ("full_continuous", "full", 30.0, 0, 8000, 0.5, True, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=176 * 16000, max_logit_rows=2400, stride=160)),
    # the 300 s batch line of bench.py (the 30 s clip tiled, ONE feed: a 16 946-position encoder chunk, 3761 decoder steps, KV to 3799)
    ("full_batch300", "full", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=300 * 16000, max_logit_rows=3800, stride=256)),
    # BASELINE config 3 itself: 300 s in 0.5 s feeds, -I 0.5, continuous mode (the full stream reset at ~160 s included); the golden of
    # `bench.py --mode stream` (3754 decoder steps; ~50 min of reference CPU time on 8 cores)
    ("full_stream300", "full", 30.0, 0, 8000, 0.5, True, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=300 * 16000, max_logit_rows=3900, stride=256)),
    # BASELINE config 4's audio length on one GPU: 600 s, one feed (30 196 encoder positions, 7511 decoder steps, KV to 7549)
    ("full_batch600", "full", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=600 * 16000, max_logit_rows=7600, stride=512)),
]


# Round 5: the "realistic statistics" checkpoints (tools/synth_model.c, style -rs: Student-t weights, outlier channels with matching
# norm spikes, massive-activation rows, residual gains at the reproducibility limit - tools/amplification.py).  Preset "<geometry>-rs"
# = the reference library of <geometry> on the -rs weights.
RS_CASES = [
    ("smallrs_batch", "small-rs", 8.0, 4, None, None, False),
    ("smallrs_long", "small-rs", 95.0, 6, None, None, False),          # > 1024 decoder positions at the real per-layer shapes
    ("smallrs_stream", "small-rs", 20.0, 7, 8000, 0.5, True),          # config 3's feed pattern: 25-row encoder chunks
    # the headline input through the full model
    ("fullrs_batch", "full-rs", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav"),
]
RS_LONG_CASES = [
    # 95 s, one feed: ~1150 decoder steps - crosses 512 keys (one-tile -> two-tile attention members of k_dec_stack) and 1024 keys
    # (k_dec_stack -> one k_ffn_attn12<LONG> launch per layer) in the middle of the decode
    ("fullrs_batch95", "full-rs", 30.0, 0, None, None, False, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=95 * 16000, max_logit_rows=1300, stride=64)),
    # BASELINE config 3's feed pattern through a continuous-mode restart: 176 s in 0.5 s feeds, -I 0.5 (2204 steps, KV to 2000, full stream reset)
    ("fullrs_continuous", "full-rs", 30.0, 0, 8000, 0.5, True, None, "benchmark/night1968/45s_right_through_the_billboard.wav",
     dict(tile_to=176 * 16000, max_logit_rows=2400, stride=160)),
]


MARGIN_EDGES = np.array([0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 3e-1, 1, 1e9], np.float64)
STRIDE = 16


def summarise(r, vocab, stride_override=None):
    lg = r["logits"]
    out = dict(tokens=r["tokens"].astype(np.int32), pieces=np.array(r["pieces"], dtype=object))
    out["n_distinct"] = np.int32(len(set(r["tokens"].tolist())))
    if lg is not None and len(lg):
        part = np.argpartition(-lg, 8, axis=1)[:, :8]
        pv = np.take_along_axis(lg, part, axis=1)
        o2 = np.argsort(-pv, axis=1, kind="stable")
        order = np.take_along_axis(part, o2, axis=1)
        out["top_ids"] = order.astype(np.int32)
        out["top_vals"] = np.take_along_axis(lg, order, axis=1).astype(np.float32)
        out["logits_head"] = lg[:(1 if vocab > 65536 else 4)].astype(np.float32)
        stride = 32 if vocab > 65536 else max(STRIDE, -(-len(lg) // 64))     # keeps the fixture at a few MB
        if stride_override:
            stride = stride_override
        steps = np.arange(0, len(lg), stride)
        out["stride_steps"] = steps.astype(np.int32)
        out["logits_stride"] = lg[steps].astype(np.float32)
        out["margin"] = (out["top_vals"][:, 0] - out["top_vals"][:, 1]).astype(np.float32)
        out["margin_edges"] = MARGIN_EDGES
        out["margin_hist"] = np.histogram(out["margin"], bins=MARGIN_EDGES)[0].astype(np.int32)
    return out


REF_SAMPLES = "/root/reference/samples"


def case_audio(R, spec, secs, aseed):
    """(samples f32, int16 copy or None).  spec None = synthetic; otherwise a path under the reference's samples/."""
    if spec is None:
        return synth_speech(secs, aseed), None
    a = R.load_wav(os.path.join(REF_SAMPLES, spec))          # the reference's own WAV loader (s16 / 32768)
    if a is None:
        raise RuntimeError("cannot load " + spec)
    if secs:
        a = a[:int(secs * 16000)]
    i16 = np.round(a * 32768.0).astype(np.int16)
    assert np.array_equal(i16.astype(np.float32) / 32768.0, a)
    return a, i16


def feeds_for(spec, n):
    if spec is None:
        return None
    if spec == "1s":
        return [16000] * (n // 16000 + 1)
    return [int(spec)] * (n // int(spec) + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the full-size (8.9 GB) model (minutes of CPU)")
    ap.add_argument("--only", default=None)
    ap.add_argument("--rs", action="store_true", help="the realistic-statistics cases (small ones in seconds, fullrs_batch in ~10 min)")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    libs = {}
    cases = list(CASES)
    if args.full:
        cases += FULL_CASES
    if args.only in [c[0] for c in LONG_CASES]:
        cases += LONG_CASES
    if args.rs:
        cases = list(RS_CASES)
    if args.only in [c[0] for c in RS_CASES + RS_LONG_CASES]:
        cases = list(RS_CASES) + list(RS_LONG_CASES)
    for case in cases:
        name, preset, secs, aseed, feed, interval, cont = case[:7]
        delay_ms = case[7] if len(case) > 7 else None
        wav = case[8] if len(case) > 8 else None
        extra = case[9] if len(case) > 9 else {}
        if args.only and args.only != name:
            continue
        geom = preset.split("-")[0]            # "full-rs" = the full geometry's reference library on the -rs weights
        if geom not in libs:
            libs[geom] = RefLib(geom)
        R = libs[geom]
        d = vo.PRESETS[geom]
        audio, audio_i16 = case_audio(R, wav, secs, aseed)
        if extra.get("tile_to"):
            audio = np.tile(audio, -(-extra["tile_to"] // len(audio)))[:extra["tile_to"]].copy()
        ctx = R.load(model_dir(preset))
        r = R.transcribe_stream(ctx, audio, feed_sizes=feeds_for(feed, len(audio)), interval=interval,
                                continuous=cont, vocab=d.vocab,
                                max_logit_rows=extra.get("max_logit_rows", 4096 if geom != "full" else 512),
                                delay_ms=delay_ms)
        R.free(ctx)
        out = summarise(r, d.vocab, extra.get("stride"))
        if extra.get("tile_to"):
            out["audio_total_samples"] = np.int64(extra["tile_to"])
        out["meta"] = np.array([preset, str(secs), str(aseed), str(feed), str(interval), str(int(cont)),
                                str(delay_ms if delay_ms is not None else 480), str(wav)], dtype=object)
        if audio_i16 is not None:
            out["audio_i16"] = audio_i16
        np.savez_compressed(os.path.join(GOLD, f"stream_{name}.npz"), **out)
        mg = out.get("margin")
        print(f"{name}: {len(out['tokens'])} steps, {int(out['n_distinct'])} distinct tokens, "
              f"{len(out['pieces'])} pieces, min margin {mg.min() if mg is not None else None}, "
              f"margin hist {out['margin_hist'].tolist() if mg is not None else None}", flush=True)

    # ---- residual-stream taps of the headline configuration (three decoder steps, every layer) ----
    if args.only == "taps":
        R = libs.get("full") or RefLib("full")
        d = vo.PRESETS["full"]
        audio, audio_i16 = case_audio(R, "benchmark/night1968/45s_right_through_the_billboard.wav", 30.0, 0)
        steps = [0, 193, 385]
        ctx = R.load(model_dir("full"))
        r = R.transcribe_stream(ctx, audio, vocab=0, tap_steps=steps, tap_hidden=d.dec_dim, tap_vectors=2 * d.dec_layers + 1)
        R.free(ctx)
        np.savez_compressed(os.path.join(GOLD, "taps_full_batch.npz"), steps=np.array(steps, np.int32), taps=r["taps"].astype(np.float32),
                            tokens=r["tokens"].astype(np.int32))
        t = r["taps"]
        print("taps_full_batch:", t.shape, "rms per layer at step 193:", np.sqrt((t[1] ** 2).mean(axis=1)).round(3).tolist())
        return

    # ---- stage-level goldens on the tiny model (reference functions called directly) ----
    if (not args.only and not args.rs) or args.only == "stage":
        R = libs.get("tiny") or RefLib("tiny")
        d = vo.PRESETS["tiny"]
        rng = np.random.default_rng(7)
        audio = synth_speech(3.0, 9)
        mel = R.mel_stream(audio, feeds=[7000, 1, 159, 20000, len(audio) - 27160])
        ctx = R.load(model_dir("tiny"))
        xs = [rng.standard_normal((n, d.enc_dim)).astype(np.float32) for n in (30, 7, 50, 64, 1, 3)]
        enc = [R.encoder_forward_incremental(ctx, x, d.enc_dim) for x in xs]
        ad = R.adapter_forward(ctx, np.concatenate(enc)[:152], d.dec_dim)
        emb = (rng.standard_normal((45, d.dec_dim)) * 0.5).astype(np.float32)
        R.decoder_prefill(ctx, emb[:38])
        toks, logs = [], []
        for i in range(38, 45):
            t, lg = R.decoder_forward(ctx, emb[i], d.vocab)
            toks.append(t); logs.append(lg)
        R.free(ctx)
        np.savez_compressed(os.path.join(GOLD, "stage_tiny.npz"), audio=audio, mel=mel,
                            **{f"enc_in{i}": x for i, x in enumerate(xs)},
                            **{f"enc_out{i}": e for i, e in enumerate(enc)},
                            adapter_out=ad, dec_emb=emb, dec_tokens=np.array(toks, np.int32), dec_logits=np.stack(logs))
        print("stage_tiny: mel", mel.shape, "enc chunks", [e.shape[0] for e in enc], "dec tokens", toks)


if __name__ == "__main__":
    main()
