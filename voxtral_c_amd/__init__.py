"""voxtral_c_amd — Python (ctypes) mirror of the voxtral.h C API of the MI355X engine.

The product is the pair of in-tree shared objects built by voxtral_c_amd/Makefile:
``libvoxtral.so`` (plain-C host library, the reference's voxtral.h surface) on top of
``libvoxhip.so`` (hand-written HIP kernels for gfx950 behind include/vox_hip.h).  This
module only binds them; names, argument meaning and error behaviour follow the C API
(which follows the reference, see include/voxtral.h).  There is no CPU fallback: if the
libraries are missing the import fails, and ``Model`` raises if no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.environ.get("VOX_LIB_DIR") or _HERE      # (VOX_LIB_DIR: another build of the two libraries - same-box A/B of a compile-time change, tools/lib_ab.sh)
LIB_PATH = os.path.join(_LIB_DIR, "libvoxtral.so")
HIP_LIB_PATH = os.path.join(_LIB_DIR, "libvoxhip.so")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
u16p = C.POINTER(C.c_uint16)


class VoxError(RuntimeError):
    pass


class _Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mel_bins", "enc_dim", "enc_layers", "enc_heads", "enc_head_dim", "enc_hidden", "enc_window",
        "dec_dim", "dec_layers", "dec_heads", "dec_kv_heads", "dec_head_dim", "dec_hidden", "dec_window",
        "vocab", "ada_dim")]


def _ptrs(*names):
    return [(n, C.c_void_p) for n in names]


# the reference's weight-view types (include/voxtral.h, reference voxtral.h:56-148): pointer fields only
class _EncLayer(C.Structure):
    _fields_ = _ptrs("wq_weight", "wq_weight_bf16", "wq_bias", "wk_weight", "wk_weight_bf16", "wv_weight", "wv_weight_bf16",
                     "wv_bias", "wo_weight", "wo_weight_bf16", "wo_bias", "attention_norm", "w1_weight", "w1_weight_bf16",
                     "w2_weight", "w2_weight_bf16", "w2_bias", "w3_weight", "w3_weight_bf16", "ffn_norm")


class _Encoder(C.Structure):
    _fields_ = _ptrs("conv0_weight", "conv0_bias", "conv1_weight", "conv1_bias") + [("layers", _EncLayer * 32)] + _ptrs("norm")


class _DecLayer(C.Structure):
    _fields_ = _ptrs("ada_norm_down", "ada_norm_up", "wq_weight", "wq_weight_bf16", "wk_weight", "wk_weight_bf16", "wv_weight",
                     "wv_weight_bf16", "wo_weight", "wo_weight_bf16", "attention_norm", "w1_weight", "w1_weight_bf16",
                     "w2_weight", "w2_weight_bf16", "w3_weight", "w3_weight_bf16", "ffn_norm")


class _Decoder(C.Structure):
    _fields_ = _ptrs("tok_embeddings", "tok_embeddings_bf16") + [("layers", _DecLayer * 26)] + _ptrs("norm")


class _Adapter(C.Structure):
    _fields_ = _ptrs("linear0_weight", "linear0_weight_bf16", "linear1_weight", "linear1_weight_bf16")


class _Ctx(C.Structure):
    """vox_ctx_t: the reference's fields first (same order), the engine's appended (include/voxtral.h)."""
    _fields_ = ([("encoder", _Encoder), ("adapter", _Adapter), ("decoder", _Decoder),
                 ("safetensors", C.c_void_p), ("model_dir", C.c_char * 512)] +
                _ptrs("kv_cache_k", "kv_cache_v", "kv_cache_k_f16", "kv_cache_v_f16") +
                [("kv_cache_fp16", C.c_int), ("kv_cache_len", C.c_int), ("kv_cache_max", C.c_int), ("kv_pos_offset", C.c_int),
                 ("delay_tokens", C.c_int), ("t_cond", C.c_float * 3072), ("ada_scale", f32p), ("use_bf16", C.c_int)] +
                _ptrs("enc_kv_cache_k", "enc_kv_cache_v") +
                [("enc_kv_cache_len", C.c_int), ("enc_kv_cache_max", C.c_int), ("enc_kv_cache_is_shared", C.c_int),
                 ("enc_kv_pos_offset", C.c_int), ("enc_inc_cap", C.c_int)] +
                _ptrs("enc_inc_x_norm", "enc_inc_q", "enc_inc_k", "enc_inc_v", "enc_inc_attn_out", "enc_inc_proj_out",
                      "enc_inc_gate", "enc_inc_up", "enc_inc_ffn_out", "enc_inc_positions", "enc_inc_rope_freqs",
                      "dec_x", "dec_x_norm", "dec_q", "dec_k", "dec_v", "dec_attn_out", "dec_proj_out", "dec_gate", "dec_up",
                      "dec_ffn_out", "dec_rope_freqs") +
                [("dims", _Dims), ("device", C.c_int), ("engine", C.c_void_p), ("ada_down", C.c_void_p), ("ada_up", C.c_void_p),
                 ("tokenizer", C.c_void_p), ("shard_engines", C.c_void_p * 8), ("n_shard_engines", C.c_int),
                 ("owned_f32", C.c_void_p), ("n_owned_f32", C.c_int), ("cap_owned_f32", C.c_int),
                 ("n_sharded_chunks", C.c_int), ("shard_disabled", C.c_int)])


class _LoadOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("enc_window", C.c_int), ("dec_window", C.c_int), ("weight_format", C.c_int),
                ("n_devices", C.c_int), ("devices", C.c_int * 8)]


class _Timing(C.Structure):
    _fields_ = [("encode_ms", C.c_double), ("prefill_ms", C.c_double), ("decode_ms", C.c_double),
                ("decode_steps", C.c_int)]


def _load_libs():
    if not os.path.exists(HIP_LIB_PATH) or not os.path.exists(LIB_PATH):
        raise ImportError(
            f"voxtral_c_amd: {HIP_LIB_PATH} / {LIB_PATH} not built. Run `make -C voxtral_c_amd` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    # RTLD_LOCAL: libvoxtral.so finds libvoxhip.so through its own DT_NEEDED + rpath, and the
    # reference-named symbols it exports (vox_linear_bf16, vox_mel_feed, ...) must never interpose
    # on another library in the process that defines the same names (the test oracle oracle/_ref).
    hip = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_LOCAL)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    return hip, lib


hip, lib = _load_libs()

# ---- prototypes: voxtral.h ---------------------------------------------------------
lib.vox_load.restype = C.POINTER(_Ctx)
lib.vox_load.argtypes = [C.c_char_p]
lib.vox_load_ex.restype = C.POINTER(_Ctx)
lib.vox_load_ex.argtypes = [C.c_char_p, C.POINTER(_LoadOpts)]
lib.vox_free.argtypes = [C.POINTER(_Ctx)]
lib.vox_set_delay.argtypes = [C.POINTER(_Ctx), C.c_int]
lib.vox_stream_init.restype = C.c_void_p
lib.vox_stream_init.argtypes = [C.POINTER(_Ctx)]
lib.vox_stream_feed.argtypes = [C.c_void_p, f32p, C.c_int]
lib.vox_stream_finish.argtypes = [C.c_void_p]
lib.vox_stream_flush.argtypes = [C.c_void_p]
lib.vox_stream_free.argtypes = [C.c_void_p]
lib.vox_stream_get.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
lib.vox_stream_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_float]
lib.vox_stream_get_alt.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]
lib.vox_set_processing_interval.argtypes = [C.c_void_p, C.c_float]
lib.vox_stream_set_continuous.argtypes = [C.c_void_p, C.c_int]
lib.vox_stream_token_ids.argtypes = [C.c_void_p, i32p, C.c_int]
lib.vox_stream_record_ids.argtypes = [C.c_void_p, C.c_int]
lib.vox_stream_force_tokens.argtypes = [C.c_void_p, i32p, C.c_int]
lib.vox_stream_record_logits.argtypes = [C.c_void_p, C.c_int]
lib.vox_stream_recorded_logits.argtypes = [C.c_void_p, C.POINTER(f32p)]
lib.vox_transcribe_audio.restype = C.c_void_p
lib.vox_transcribe_audio.argtypes = [C.POINTER(_Ctx), f32p, C.c_int]
lib.vox_encoder_forward_incremental.restype = C.c_void_p
lib.vox_encoder_forward_incremental.argtypes = [C.POINTER(_Ctx), f32p, C.c_int, i32p]
lib.vox_adapter_forward.restype = C.c_void_p
lib.vox_adapter_forward.argtypes = [C.POINTER(_Ctx), f32p, C.c_int, i32p]
lib.vox_decoder_forward.argtypes = [C.POINTER(_Ctx), f32p, f32p]
lib.vox_decoder_prefill.argtypes = [C.POINTER(_Ctx), f32p, C.c_int]
lib.vox_load_wav.restype = C.c_void_p
lib.vox_load_wav.argtypes = [C.c_char_p, i32p]
lib.vox_mel_ctx_init.restype = C.c_void_p
lib.vox_mel_ctx_init.argtypes = [C.c_int]
lib.vox_mel_feed.argtypes = [C.c_void_p, f32p, C.c_int]
lib.vox_mel_finish.argtypes = [C.c_void_p, C.c_int]
lib.vox_mel_data.restype = C.c_void_p
lib.vox_mel_data.argtypes = [C.c_void_p, i32p]
lib.vox_mel_free.argtypes = [C.c_void_p]
# ---- prototypes: vox_hip.h (kernel-level / fused surface) ---------------------------
hip.vox_hip_device_count.restype = C.c_int
hip.vox_hip_last_error.restype = C.c_char_p
hip.vox_hip_memory_used.restype = C.c_size_t
hip.vox_hip_memory_used.argtypes = [C.c_void_p]
hip.vox_hip_linear_bf16.argtypes = [C.c_void_p, f32p, f32p, u16p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
hip.vox_hip_causal_attention.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p] + [C.c_int] * 5 + \
    [C.c_float, C.c_int, C.c_int]
hip.vox_hip_conv_stem.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int]
hip.vox_hip_conv_stem_pad_odd.argtypes = [C.c_void_p, f32p]
hip.vox_hip_reset_encoder.argtypes = [C.c_void_p]
hip.vox_hip_reset_decoder.argtypes = [C.c_void_p]
hip.vox_hip_time_decoder_step.restype = C.c_double
hip.vox_hip_time_decoder_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
hip.vox_hip_weight_format.restype = C.c_int
hip.vox_hip_weight_format.argtypes = [C.c_void_p]
hip.vox_hip_active_paths.restype = C.c_uint
hip.vox_hip_active_paths.argtypes = [C.c_void_p]
hip.vox_hip_sync.argtypes = [C.c_void_p]
hip.vox_hip_get_timing.argtypes = [C.c_void_p, C.POINTER(_Timing)]
hip.vox_hip_reset_timing.argtypes = [C.c_void_p]
hip.vox_hip_adapter_read.argtypes = [C.c_void_p, C.c_int64, C.c_int, f32p]
hip.vox_hip_adapter_append.argtypes = [C.c_void_p, f32p, C.c_int]
hip.vox_hip_adapter_rows.restype = C.c_int64
hip.vox_hip_adapter_rows.argtypes = [C.c_void_p]
hip.vox_hip_adapter_devptr.restype = C.c_void_p
hip.vox_hip_adapter_devptr.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
hip.vox_hip_stream_encode.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
hip.vox_hip_mel_frames.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int]
hip.vox_hip_decoder_prefill_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, f32p]
hip.vox_hip_decoder_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, f32p]
hip.vox_hip_sync.argtypes = [C.c_void_p]

_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _fp(a):
    return a.ctypes.data_as(f32p)


# vox_hip.h enum vox_hip_path
PATHS = {"gemm_mfma_bf16x3": 1 << 0, "gemm_mfma_f32": 1 << 1, "gemm_splitk": 1 << 2, "attn_enc_mfma": 1 << 3,
         "attn_dec_dpp": 1 << 4, "gemv3": 1 << 5, "fp8_decode": 1 << 6, "skinny_enc": 1 << 7, "dec_fused": 1 << 8,
         "gemm_planes": 1 << 9, "ffn_fused": 1 << 10, "rowsgemm": 1 << 11, "ffn_attn12": 1 << 12, "dec_stack": 1 << 13, "fp8_mfma": 1 << 14, "enc_stack": 1 << 15}
PATH_ALL_BF16 = sum(v for k, v in PATHS.items() if k not in ("fp8_decode", "fp8_mfma"))


def device_count():
    return hip.vox_hip_device_count()


def set_verbose(level):
    C.c_int.in_dll(lib, "vox_verbose").value = int(level)


def load_wav(path):
    """vox_load_wav: 16-bit PCM WAV -> mono float32 @16 kHz."""
    n = C.c_int(0)
    p = lib.vox_load_wav(os.fsencode(path), C.byref(n))
    if not p:
        raise VoxError(f"vox_load_wav failed for {path}")
    out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value,)).copy()
    _libc.free(p)
    return out


class Model:
    """vox_load / vox_free (+ the stage-level functions of voxtral.h:309-328)."""

    def __init__(self, model_dir, device=0, enc_window=0, dec_window=0, weights="bf16", devices=None):
        opts = _LoadOpts(device, enc_window, dec_window, 1 if weights == "fp8" else 0)
        if devices:
            opts.n_devices = len(devices)
            for i, dv in enumerate(devices):
                opts.devices[i] = dv
        self._ctx = lib.vox_load_ex(os.fsencode(model_dir), C.byref(opts))
        if not self._ctx:
            raise VoxError(f"vox_load failed for {model_dir}: {hip.vox_hip_last_error().decode()}")
        self.dims = self._ctx.contents.dims
        self.engine = self._ctx.contents.engine

    def close(self):
        if self._ctx:
            lib.vox_free(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def ctx(self):
        return self._ctx.contents

    def set_delay(self, delay_ms):
        lib.vox_set_delay(self._ctx, int(delay_ms))

    def active_paths(self):
        """vox_hip_active_paths as (mask, [names]): which production kernel families are live."""
        m = hip.vox_hip_active_paths(self.engine)
        return m, [k for k, v in PATHS.items() if m & v]

    def memory_used(self):
        return hip.vox_hip_memory_used(self.engine)

    # -- stage-level (host arrays in/out, GPU compute) --
    def reset_counters(self):
        """Mimic what vox_stream_init / prefill do to the public counters."""
        c = self._ctx.contents
        c.kv_cache_len = 0
        c.kv_pos_offset = 0
        c.enc_kv_cache_len = 0
        c.enc_kv_pos_offset = 0

    def encoder_forward_incremental(self, x_new):
        x = np.ascontiguousarray(x_new, np.float32)
        n = C.c_int(0)
        p = lib.vox_encoder_forward_incremental(self._ctx, _fp(x), x.shape[0], C.byref(n))
        if not p:
            raise VoxError("vox_encoder_forward_incremental failed")
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value, self.dims.enc_dim)).copy()
        _libc.free(p)
        return out

    def adapter_forward(self, enc_out):
        x = np.ascontiguousarray(enc_out, np.float32)
        n = C.c_int(0)
        p = lib.vox_adapter_forward(self._ctx, _fp(x), x.shape[0], C.byref(n))
        if not p:
            raise VoxError("vox_adapter_forward failed")
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value, self.dims.dec_dim)).copy()
        _libc.free(p)
        return out

    def decoder_prefill(self, embeds):
        x = np.ascontiguousarray(embeds, np.float32)
        lib.vox_decoder_prefill(self._ctx, _fp(x), x.shape[0])

    def decoder_forward(self, embed):
        x = np.ascontiguousarray(embed, np.float32)
        logits = np.empty(self.dims.vocab, np.float32)
        tok = lib.vox_decoder_forward(self._ctx, _fp(x), _fp(logits))
        return tok, logits

    def conv_stem(self, mel_new):
        m = np.ascontiguousarray(mel_new, np.float32)
        cap = m.shape[0] // 2 + 2
        out = np.empty((cap, self.dims.enc_dim), np.float32)
        rows = hip.vox_hip_conv_stem(self.engine, _fp(m), m.shape[0], _fp(out), cap)
        if rows < 0:
            raise VoxError("vox_hip_conv_stem failed")
        return out[:rows].copy()

    def reset_encoder(self):
        hip.vox_hip_reset_encoder(self.engine)
        c = self._ctx.contents
        c.enc_kv_cache_len = 0
        c.enc_kv_pos_offset = 0

    def mel_frames(self, padded_samples, n_frames):
        s = np.ascontiguousarray(padded_samples, np.float32)
        out = np.empty((n_frames, self.dims.mel_bins), np.float32)
        if hip.vox_hip_mel_frames(self.engine, _fp(s), n_frames, _fp(out), 0) != 0:
            raise VoxError("vox_hip_mel_frames failed")
        return out

    # -- kernel-level --
    def linear_bf16(self, x, w_bf16, bias=None, impl=0):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w_bf16, np.uint16)
        M, K = x.shape
        N = w.shape[0] // 2 if impl == 9 else w.shape[0]          # impl 9 (SwiGLU launch of k_gemm_planes): w = [w1; w3]
        y = np.empty((M, N), np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        rc = hip.vox_hip_linear_bf16(self.engine, _fp(y), _fp(x), w.ctypes.data_as(u16p),
                                     None if b is None else _fp(b), M, K, N, impl)
        if rc != 0:
            raise VoxError("vox_hip_linear_bf16 failed: " + hip.vox_hip_last_error().decode())
        return y

    def causal_attention(self, q, k, v, n_heads, n_kv_heads, head_dim, scale, window, q_offset):
        q = np.ascontiguousarray(q, np.float32)
        k = np.ascontiguousarray(k, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty_like(q)
        rc = hip.vox_hip_causal_attention(self.engine, _fp(out), _fp(q), _fp(k), _fp(v), q.shape[0], k.shape[0],
                                          n_heads, n_kv_heads, head_dim, scale, window, q_offset)
        if rc != 0:
            raise VoxError("vox_hip_causal_attention failed: " + hip.vox_hip_last_error().decode())
        return out

    def time_decoder_step(self, iters=20, kv_len=400):
        return hip.vox_hip_time_decoder_step(self.engine, iters, kv_len)

    def timing(self):
        t = _Timing()
        hip.vox_hip_get_timing(self.engine, C.byref(t))
        return dict(encode_ms=t.encode_ms, prefill_ms=t.prefill_ms, decode_ms=t.decode_ms,
                    decode_steps=t.decode_steps)

    # -- convenience --
    def stream(self, **kw):
        return Stream(self, **kw)

    def transcribe(self, samples, feed_sizes=None, interval=None, continuous=False, record_logits=0,
                   delay_ms=None, n_alt=1, alt_cutoff=0.0, force_tokens=None):
        """Drive the stream API like the reference CLI does. Returns dict(tokens, pieces, logits).
        force_tokens: teacher forcing (vox_stream_force_tokens) - `tokens` is then the engine's own
        argmax at every step given the forced history."""
        if delay_ms is not None:
            self.set_delay(delay_ms)
        s = Stream(self)
        try:
            if interval is not None:
                s.set_processing_interval(interval)
            if continuous:
                s.set_continuous(True)
            if record_logits:
                s.record_logits(record_logits)
            if n_alt > 1:
                s.set_alt(n_alt, alt_cutoff)
            if force_tokens is not None:
                s.force_tokens(force_tokens)
            x = np.ascontiguousarray(samples, np.float32)
            pieces = []
            if feed_sizes is None:
                feed_sizes = [len(x)]
            off = 0
            for n in feed_sizes:
                n = min(n, len(x) - off)
                if n <= 0:
                    break
                s.feed(x[off:off + n])
                off += n
                pieces.extend(s.get())
            s.finish()
            pieces.extend(s.get())
            return dict(tokens=s.token_ids(), pieces=pieces, logits=s.recorded_logits())
        finally:
            s.free()


class Stream:
    """vox_stream_* (voxtral.h streaming API)."""

    def __init__(self, model):
        self.model = model
        self._s = lib.vox_stream_init(model._ctx)
        if not self._s:
            raise VoxError("vox_stream_init failed")
        lib.vox_stream_record_ids(self._s, 1)
        self._forced = None

    def feed(self, samples):
        x = np.ascontiguousarray(samples, np.float32)
        return lib.vox_stream_feed(self._s, _fp(x), len(x))

    def finish(self):
        return lib.vox_stream_finish(self._s)

    def flush(self):
        return lib.vox_stream_flush(self._s)

    def get(self):
        out = []
        buf = (C.c_char_p * 64)()
        while True:
            n = lib.vox_stream_get(self._s, buf, 64)
            if n <= 0:
                break
            out.extend(buf[i].decode("utf-8", "replace") for i in range(n))
        return out

    def get_alt(self, n_alt=3):
        out = []
        buf = (C.c_char_p * (64 * n_alt))()
        while True:
            n = lib.vox_stream_get_alt(self._s, buf, 64, n_alt)
            if n <= 0:
                break
            for i in range(n):
                out.append([None if buf[i * n_alt + a] is None else buf[i * n_alt + a].decode("utf-8", "replace")
                            for a in range(n_alt)])
        return out

    def set_alt(self, n_alt, cutoff):
        lib.vox_stream_set_alt(self._s, n_alt, cutoff)

    def set_processing_interval(self, seconds):
        lib.vox_set_processing_interval(self._s, seconds)

    def set_continuous(self, enable=True):
        lib.vox_stream_set_continuous(self._s, 1 if enable else 0)

    def record_logits(self, max_rows):
        lib.vox_stream_record_logits(self._s, int(max_rows))

    def force_tokens(self, ids):
        self._forced = np.ascontiguousarray(ids, np.int32)          # must outlive the stream's use of it
        lib.vox_stream_force_tokens(self._s, self._forced.ctypes.data_as(i32p), len(self._forced))

    def token_ids(self):
        n = lib.vox_stream_token_ids(self._s, None, 0)
        out = np.zeros(n, np.int32)
        if n:
            lib.vox_stream_token_ids(self._s, out.ctypes.data_as(i32p), n)
        return out

    def recorded_logits(self):
        p = f32p()
        rows = lib.vox_stream_recorded_logits(self._s, C.byref(p))
        if not rows:
            return None
        return np.ctypeslib.as_array(p, shape=(rows, self.model.dims.vocab)).copy()

    def free(self):
        if self._s:
            lib.vox_stream_free(self._s)
            self._s = None
