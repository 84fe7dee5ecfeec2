"""CPU: pin the numpy oracle (oracle/vox_oracle.py) to the real reference.

 * against the committed golden fixtures (generated from oracle/_ref by tools/make_golden.py)
 * live against oracle/_ref when it is built (kernel-level functions, shape-generic)
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, have_ref, model_dir
from oracle import vox_oracle as vo

TINY = vo.PRESETS["tiny"]


def gold(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)


def test_mel_matches_reference_golden():
    g = gold("stage_tiny.npz")
    mel = vo.mel_stream(g["audio"])
    assert mel.shape == g["mel"].shape
    # the reference is -ffast-math f32: low-energy bins move by a few 1e-4 with summation order
    assert np.abs(mel - g["mel"]).max() < 1e-3
    assert np.abs(mel - g["mel"]).mean() < 2e-5


def test_encoder_chunks_and_adapter_match_golden():
    g = gold("stage_tiny.npz")
    o = vo.Oracle(model_dir("tiny"), TINY)
    outs = []
    for i in range(6):
        e = o.encoder_incremental(g[f"enc_in{i}"])
        assert np.abs(e - g[f"enc_out{i}"]).max() < 2e-4, i
        outs.append(e)
    ad = o.adapter(np.concatenate(outs)[:152])
    assert np.abs(ad - g["adapter_out"]).max() < 2e-4


def test_decoder_prefill_and_steps_match_golden():
    g = gold("stage_tiny.npz")
    o = vo.Oracle(model_dir("tiny"), TINY)
    emb = g["dec_emb"]
    o.decoder_prefill(emb[:38])
    for i in range(38, 45):
        tok, lg = o.decoder_forward(emb[i])
        assert tok == int(g["dec_tokens"][i - 38])
        assert np.abs(lg - g["dec_logits"][i - 38]).max() < 1e-3


def test_stream_transcription_matches_golden():
    from audio_util import synth_speech
    g = gold("stream_tiny_batch.npz")
    o = vo.Oracle(model_dir("tiny"), TINY)
    toks, logs = o.transcribe(synth_speech(12.0, 1))
    assert np.array_equal(toks, g["tokens"])
    top = np.take_along_axis(logs, g["top_ids"], axis=1)
    assert np.abs(top - g["top_vals"]).max() < 1e-3


def test_stream_transcription_with_other_delay_matches_golden():
    """vox_set_delay(240 ms): time conditioning (voxtral.c:31-80), prompt length 1+32+3, right padding."""
    from audio_util import synth_speech
    g = gold("stream_tiny_delay240.npz")
    o = vo.Oracle(model_dir("tiny"), TINY)
    o.delay_tokens = int(g["meta"][6]) // 80
    o.update_time_conditioning()
    toks, logs = o.transcribe(synth_speech(float(g["meta"][1]), int(g["meta"][2])))
    assert np.array_equal(toks, g["tokens"])
    top = np.take_along_axis(logs, g["top_ids"], axis=1)
    assert np.abs(top - g["top_vals"]).max() < 1e-3


@pytest.mark.skipif(not have_ref("tiny"), reason="oracle/_ref not built")
def test_kernels_match_live_reference(ref_tiny):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 256)).astype(np.float32)
    w = vo.f32_to_bf16(rng.standard_normal((96, 256)).astype(np.float32) * 0.1)
    b = rng.standard_normal(96).astype(np.float32)
    assert np.abs(vo.linear_bf16(x, w, b) - ref_tiny.linear_bf16(x, w, b)).max() < 1e-5
    assert np.abs(vo.linear_bf16(x[:1], w) - ref_tiny.linear_bf16(x[:1], w)).max() < 1e-5
    g = rng.standard_normal(256).astype(np.float32)
    assert np.abs(vo.rms_norm(x, g, 1e-5) - ref_tiny.rms_norm(x, g, 1e-5)).max() < 1e-5
    assert np.abs(vo.gelu(x) - ref_tiny.gelu(x)).max() < 1e-6
    assert np.abs(vo.silu(x) - ref_tiny.silu(x)).max() < 1e-6
    pos = np.array([0, 1, 5, 1000, 30000], np.int32)
    fr = vo.rope_freqs(pos, 64, 1e6)
    assert np.abs(fr - ref_tiny.rope_freqs(pos, 64, 1e6)).max() < 2e-6
    q = rng.standard_normal((5, 4 * 64)).astype(np.float32)
    assert np.abs(vo.apply_rope(q, fr, 4, 64) - ref_tiny.apply_rope(q, fr, 4, 64)).max() < 1e-6
    for (sq, sk, nh, nkv, hd, win, off) in [(7, 20, 4, 4, 64, 6, 13), (3, 3, 8, 2, 128, 64, 0), (1, 90, 8, 2, 128, 64, 89)]:
        qq = rng.standard_normal((sq, nh * hd)).astype(np.float32)
        kk = rng.standard_normal((sk, nkv * hd)).astype(np.float32)
        vv = rng.standard_normal((sk, nkv * hd)).astype(np.float32)
        a = vo.causal_attention(qq, kk, vv, nh, nkv, hd, 0.1, win, off)
        r = ref_tiny.causal_attention(qq, kk, vv, nh, nkv, hd, 0.1, win, off)
        assert np.abs(a - r).max() < 1e-5
    xc = rng.standard_normal((16, 33)).astype(np.float32)
    wc = rng.standard_normal((8, 48)).astype(np.float32)
    bc = rng.standard_normal(8).astype(np.float32)
    for stride in (1, 2):
        assert np.abs(vo.causal_conv1d(xc, wc, bc, stride) - ref_tiny.causal_conv1d(xc, wc, bc, stride)).max() < 1e-4


@pytest.mark.skipif(not have_ref("tiny"), reason="oracle/_ref not built")
def test_conv_stem_chunking_is_invisible():
    """The oracle's tail-based conv stem (reference semantics) equals one-shot processing, as long
    as no mid-stream chunk is a single mel frame (the reference's tail update zeroes the older
    slot for n == 1, voxtral.c:604-609 — reproduced by the oracle and the engine, and only
    reachable at finish() or with sub-10 ms processing intervals)."""
    o = vo.Oracle(model_dir("tiny"), TINY)
    rng = np.random.default_rng(3)
    mel = rng.standard_normal((401, 128)).astype(np.float32) * 0.5
    whole = o.conv_stem(mel)
    o.reset_encoder()
    parts = [o.conv_stem(mel[a:b]) for a, b in [(0, 313), (313, 315), (315, 318), (318, 400), (400, 401)]]
    got = np.concatenate(parts)
    assert got.shape == whole.shape == (200, TINY.enc_dim)
    assert np.abs(got - whole).max() < 1e-5
