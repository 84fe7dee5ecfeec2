"""GPU parity tests of the exported BATCH twins of the streaming entry points (round 4): each is called through the C-ABI of
libvoxtral.so and compared with the reference's own function of the same name, run live from oracle/_ref on the same input.
    vox_mel_spectrogram      voxtral_audio.c:294     (reflect-padded batch log-mel)
    vox_mel_discard_before   voxtral_audio.c:645     (+ vox_mel_frame_offset: the ring bookkeeping the stream relies on)
    vox_encoder_forward      voxtral_encoder.c:135   (batch conv stem + all encoder layers + final norm)
    vox_transcribe_stdin     voxtral.c:1371          (WAV on stdin -> one offline feed; raw s16le -> 4096-sample feeds)
"""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from audio_util import synth_speech
from conftest import ROOT, have_ref, model_dir

pytestmark = pytest.mark.gpu
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def vox():
    import voxtral_c_amd as v
    if v.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return v


def _ref(variant):
    if not have_ref(variant):
        pytest.skip(f"oracle/_ref/libvoxref_{variant}.so not shipped")
    from oracle.ref_binding import RefLib
    return RefLib(variant)


def _take(libc_free, p, shape):
    out = np.ctypeslib.as_array(C.cast(p, f32p), shape=shape).copy()
    libc_free(p)
    return out


def test_mel_spectrogram_matches_the_reference_function(vox):
    R = _ref("tiny")
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for L in (vox.lib, R.lib):
        L.vox_mel_spectrogram.restype = C.c_void_p
        L.vox_mel_spectrogram.argtypes = [f32p, C.c_int, i32p]
    with vox.Model(model_dir("tiny"), enc_window=48, dec_window=64):      # vox_mel_spectrogram runs on the loaded model's device
        for secs, seed in ((3.0, 9), (0.55, 10), (7.3, 11)):
            a = np.ascontiguousarray(synth_speech(secs, seed), np.float32)
            n0, n1 = C.c_int(0), C.c_int(0)
            p0 = vox.lib.vox_mel_spectrogram(a.ctypes.data_as(f32p), len(a), C.byref(n0))
            p1 = R.lib.vox_mel_spectrogram(a.ctypes.data_as(f32p), len(a), C.byref(n1))
            assert p0 and p1 and n0.value == n1.value > 0, (secs, n0.value, n1.value)
            m0, m1 = _take(libc.free, p0, (n0.value, 128)), _take(libc.free, p1, (n1.value, 128))
            d = np.abs(m0 - m1)
            assert d.max() < 1e-3 and d.mean() < 2e-5, (secs, float(d.max()), float(d.mean()))


def test_mel_discard_before_keeps_the_reference_bookkeeping(vox):
    """feed -> discard -> feed -> discard (beyond what exists) -> finish: frame offset, frame count and every kept frame as
    in the reference."""
    R = _ref("tiny")
    a = np.ascontiguousarray(synth_speech(6.0, 21), np.float32)
    outs = []
    with vox.Model(model_dir("tiny"), enc_window=48, dec_window=64):
        for L in (vox.lib, R.lib):
            L.vox_mel_ctx_init.restype = C.c_void_p; L.vox_mel_ctx_init.argtypes = [C.c_int]
            L.vox_mel_feed.argtypes = [C.c_void_p, f32p, C.c_int]
            L.vox_mel_finish.argtypes = [C.c_void_p, C.c_int]
            L.vox_mel_data.restype = C.c_void_p; L.vox_mel_data.argtypes = [C.c_void_p, i32p]
            L.vox_mel_frame_offset.argtypes = [C.c_void_p]
            L.vox_mel_discard_before.argtypes = [C.c_void_p, C.c_int]
            L.vox_mel_free.argtypes = [C.c_void_p]
            ctx = L.vox_mel_ctx_init(32 * 1280)
            trace = []

            def snap():
                nf = C.c_int(0)
                p = L.vox_mel_data(ctx, C.byref(nf))
                m = np.ctypeslib.as_array(C.cast(p, f32p), shape=(nf.value, 128)).copy() if nf.value else np.zeros((0, 128), np.float32)
                trace.append((L.vox_mel_frame_offset(ctx), nf.value, m))
            L.vox_mel_feed(ctx, a.ctypes.data_as(f32p), 30000); snap()
            L.vox_mel_discard_before(ctx, 100); snap()
            L.vox_mel_feed(ctx, a[30000:].ctypes.data_as(f32p), 40000); snap()
            L.vox_mel_discard_before(ctx, 100); snap()            # no-op: already discarded
            L.vox_mel_discard_before(ctx, 410); snap()
            L.vox_mel_feed(ctx, a[70000:].ctypes.data_as(f32p), len(a) - 70000)
            L.vox_mel_finish(ctx, 0); snap()
            L.vox_mel_discard_before(ctx, 10 ** 6); snap()        # beyond the end
            L.vox_mel_free(ctx)
            outs.append(trace)
    for i, ((o0, n0, m0), (o1, n1, m1)) in enumerate(zip(*outs)):
        assert (o0, n0) == (o1, n1), (i, o0, n0, o1, n1)
        if n0:
            assert np.abs(m0 - m1).max() < 1e-3, (i, float(np.abs(m0 - m1).max()))


def test_encoder_forward_batch_matches_the_reference_function(vox):
    """vox_encoder_forward at the real per-layer widths (small preset: 2 layers of the 4B encoder), even and odd frame counts."""
    R = _ref("small")
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    d = model_dir("small")
    rng = np.random.default_rng(17)
    R.lib.vox_encoder_forward.restype = C.c_void_p
    R.lib.vox_encoder_forward.argtypes = [C.c_void_p, f32p, C.c_int, i32p]
    vox.lib.vox_encoder_forward.restype = C.c_void_p
    rctx = R.load(d)
    try:
        with vox.Model(d) as m:
            vox.lib.vox_encoder_forward.argtypes = [type(m._ctx), f32p, C.c_int, i32p]
            for frames in (400, 333, 50):
                mel = np.ascontiguousarray(rng.standard_normal((frames, 128)) * 0.6 - 0.4, np.float32)
                n0, n1 = C.c_int(0), C.c_int(0)
                p0 = vox.lib.vox_encoder_forward(m._ctx, mel.ctypes.data_as(f32p), frames, C.byref(n0))
                p1 = R.lib.vox_encoder_forward(rctx, mel.ctypes.data_as(f32p), frames, C.byref(n1))
                assert p0 and p1 and n0.value == n1.value == (frames + 1) // 2, (frames, n0.value, n1.value)
                e0, e1 = _take(libc.free, p0, (n0.value, m.dims.enc_dim)), _take(libc.free, p1, (n1.value, m.dims.enc_dim))
                err = float(np.abs(e0 - e1).max())
                assert err < 5e-4, (frames, err)
    finally:
        R.free(rctx)


_STDIN_DRIVER = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[3])
which, mdir = sys.argv[1], sys.argv[2]
if which == "product":
    import voxtral_c_amd as v
    opts = v._LoadOpts(0, 48, 64, 0)
    ctx = v.lib.vox_load_ex(os.fsencode(mdir), C.byref(opts))
    L = v.lib
    L.vox_transcribe_stdin.argtypes = [type(ctx)]
else:
    from oracle.ref_binding import RefLib
    R = RefLib("tiny"); L = R.lib
    ctx = R.load(mdir)
    L.vox_transcribe_stdin.argtypes = [C.c_void_p]
L.vox_transcribe_stdin.restype = C.c_void_p
p = L.vox_transcribe_stdin(ctx)
sys.stdout.write("TEXT:" + (C.string_at(p).decode("utf-8", "replace") if p else "<NULL>") + "\n")
"""


def _wav_bytes(a):
    pcm = np.round(np.clip(a, -1, 1) * 32767.0).astype("<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) +
            b"data" + struct.pack("<I", len(pcm)) + pcm), pcm


@pytest.mark.parametrize("kind", ["wav", "raw_s16le"])
def test_transcribe_stdin_prints_what_the_reference_function_returns(vox, kind, tmp_path):
    if not have_ref("tiny"):
        pytest.skip("oracle/_ref/libvoxref_tiny.so not shipped")
    wav, pcm = _wav_bytes(synth_speech(9.0, 33))
    data = wav if kind == "wav" else pcm
    drv = tmp_path / "stdin_driver.py"
    drv.write_text(_STDIN_DRIVER)
    texts = {}
    for which in ("product", "reference"):
        r = subprocess.run([sys.executable, str(drv), which, model_dir("tiny"), ROOT], input=data, capture_output=True, timeout=600,
                           env=dict(os.environ, VOX_ENC_WINDOW="48", VOX_DEC_WINDOW="64"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = [ln for ln in r.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("TEXT:")]
        assert line, r.stdout[-500:]
        texts[which] = line[-1][5:]
    assert texts["product"] == texts["reference"] and texts["product"] != "<NULL>" and len(texts["product"]) > 0, texts
