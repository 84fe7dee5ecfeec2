"""ctypes binding to the *real* reference CPU path built by oracle/Makefile.

TEST INFRASTRUCTURE ONLY: imported by tests/, tools/make_golden.py and bench.py's
cpu_baseline leg. The product (voxtral_c_amd) never imports this module.

`RefLib(variant)` loads oracle/_ref/libvoxref_<variant>.so (variant = full | small |
tiny, see oracle/Makefile) and exposes the reference entry points by their own
names (voxtral.h:217-328, voxtral_audio.h:18-69, voxtral_kernels.h:18-159), plus the
token/logit recorder from oracle/ref_hooks.c.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
u16p = C.POINTER(C.c_uint16)


def ref_available(variant="tiny"):
    return os.path.exists(os.path.join(REF_DIR, f"libvoxref_{variant}.so"))


def _fp(a):
    return a.ctypes.data_as(f32p)


class RefLib:
    def __init__(self, variant="tiny", blas_threads=None):
        path = os.path.join(REF_DIR, f"libvoxref_{variant}.so")
        self.variant = variant
        self.lib = L = C.CDLL(path, mode=C.RTLD_LOCAL)
        if blas_threads is not None:
            try:
                ob = C.CDLL(None)
                L2 = C.CDLL([p for p in open(f"/proc/{os.getpid()}/maps").read().split()
                             if "libscipy_openblas" in p][0])
                L2.scipy_openblas_set_num_threads(int(blas_threads))
            except Exception:
                pass
        L.vox_load.restype = C.c_void_p
        L.vox_load.argtypes = [C.c_char_p]
        L.vox_free.argtypes = [C.c_void_p]
        L.vox_set_delay.argtypes = [C.c_void_p, C.c_int]
        L.vox_stream_init.restype = C.c_void_p
        L.vox_stream_init.argtypes = [C.c_void_p]
        L.vox_stream_feed.argtypes = [C.c_void_p, f32p, C.c_int]
        L.vox_stream_finish.argtypes = [C.c_void_p]
        L.vox_stream_flush.argtypes = [C.c_void_p]
        L.vox_stream_free.argtypes = [C.c_void_p]
        L.vox_stream_get.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
        L.vox_stream_set_continuous.argtypes = [C.c_void_p, C.c_int]
        L.vox_set_processing_interval.argtypes = [C.c_void_p, C.c_float]
        L.vox_stream_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.vox_stream_get_alt.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]
        L.vox_encoder_forward_incremental.restype = C.c_void_p
        L.vox_encoder_forward_incremental.argtypes = [C.c_void_p, f32p, C.c_int, i32p]
        L.vox_adapter_forward.restype = C.c_void_p
        L.vox_adapter_forward.argtypes = [C.c_void_p, f32p, C.c_int, i32p]
        L.vox_decoder_forward.argtypes = [C.c_void_p, f32p, f32p]
        L.vox_decoder_prefill.argtypes = [C.c_void_p, f32p, C.c_int]
        L.vox_mel_ctx_init.restype = C.c_void_p
        L.vox_mel_ctx_init.argtypes = [C.c_int]
        L.vox_mel_feed.argtypes = [C.c_void_p, f32p, C.c_int]
        L.vox_mel_finish.argtypes = [C.c_void_p, C.c_int]
        L.vox_mel_data.restype = C.c_void_p
        L.vox_mel_data.argtypes = [C.c_void_p, i32p]
        L.vox_mel_free.argtypes = [C.c_void_p]
        L.vox_load_wav.restype = C.c_void_p
        L.vox_load_wav.argtypes = [C.c_char_p, i32p]
        L.voxref_hook_reset.argtypes = [C.c_int, C.c_int]
        L.voxref_tap_config.argtypes = [i32p, C.c_int, C.c_int, C.c_int]
        L.voxref_taps.restype = f32p
        L.voxref_tap_counts.restype = i32p
        L.voxref_hook_tokens.restype = i32p
        L.voxref_hook_logits.restype = f32p
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    # ---- kernel-level (shape-generic, no weights needed) -------------------
    def linear_bf16(self, x, w_bf16, bias=None):
        """vox_linear_bf16 / vox_linear_nobias_bf16 (voxtral_kernels.c:197-240)."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w_bf16, np.uint16)
        seq, k = x.shape
        n = w.shape[0]
        y = np.empty((seq, n), np.float32)
        L = self.lib
        if bias is None:
            L.vox_linear_nobias_bf16.argtypes = [f32p, f32p, u16p, C.c_int, C.c_int, C.c_int]
            L.vox_linear_nobias_bf16(_fp(y), _fp(x), w.ctypes.data_as(u16p), seq, k, n)
        else:
            b = np.ascontiguousarray(bias, np.float32)
            L.vox_linear_bf16.argtypes = [f32p, f32p, u16p, f32p, C.c_int, C.c_int, C.c_int]
            L.vox_linear_bf16(_fp(y), _fp(x), w.ctypes.data_as(u16p), _fp(b), seq, k, n)
        return y

    def rms_norm(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.lib.vox_rms_norm.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_float]
        self.lib.vox_rms_norm(_fp(out), _fp(x), _fp(np.ascontiguousarray(w, np.float32)),
                              x.shape[0], x.shape[1], eps)
        return out

    def gelu(self, x):
        y = np.array(x, np.float32, copy=True)
        self.lib.vox_gelu.argtypes = [f32p, C.c_int]
        self.lib.vox_gelu(_fp(y), y.size)
        return y

    def silu(self, x):
        y = np.array(x, np.float32, copy=True)
        self.lib.vox_silu.argtypes = [f32p, C.c_int]
        self.lib.vox_silu(_fp(y), y.size)
        return y

    def rope_freqs(self, pos, dim, theta):
        pos = np.ascontiguousarray(pos, np.int32)
        out = np.empty((len(pos), dim // 2, 2), np.float32)
        self.lib.vox_compute_rope_freqs.argtypes = [f32p, i32p, C.c_int, C.c_int, C.c_float]
        self.lib.vox_compute_rope_freqs(_fp(out), pos.ctypes.data_as(i32p), len(pos), dim, theta)
        return out

    def apply_rope(self, x, freqs, heads, head_dim):
        y = np.array(x, np.float32, copy=True)
        self.lib.vox_apply_rope.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int]
        self.lib.vox_apply_rope(_fp(y), _fp(np.ascontiguousarray(freqs, np.float32)),
                                y.shape[0], heads, head_dim)
        return y

    def causal_attention(self, q, k, v, n_heads, n_kv_heads, head_dim, scale, window, q_offset):
        q = np.ascontiguousarray(q, np.float32)
        k = np.ascontiguousarray(k, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty_like(q)
        self.lib.vox_causal_attention.argtypes = [f32p, f32p, f32p, f32p] + [C.c_int] * 5 + \
            [C.c_float, C.c_int, C.c_int]
        self.lib.vox_causal_attention(_fp(out), _fp(q), _fp(k), _fp(v), q.shape[0], k.shape[0],
                                      n_heads, n_kv_heads, head_dim, scale, window, q_offset)
        return out

    def causal_conv1d(self, x_cl, w, b, stride):
        """vox_causal_conv1d (voxtral_kernels.c:293). x_cl: [C_in, L], w: [C_out, C_in*3]."""
        x_cl = np.ascontiguousarray(x_cl, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        cin, length = x_cl.shape
        cout = w.shape[0]
        out_len = int(np.ceil((length - 3 + (3 - stride)) / stride + 1.0))
        out = np.empty((cout, out_len), np.float32)
        self.lib.vox_causal_conv1d.argtypes = [f32p, f32p, f32p, f32p] + [C.c_int] * 5
        self.lib.vox_causal_conv1d(_fp(out), _fp(x_cl), _fp(w),
                                   _fp(np.ascontiguousarray(b, np.float32)),
                                   cin, cout, length, 3, stride)
        return out

    # ---- mel ---------------------------------------------------------------
    def mel_stream(self, samples, left_pad=32 * 1280, feeds=None, finish=True):
        """Incremental mel exactly as the stream path drives it (voxtral_audio.c:515-633)."""
        s = np.ascontiguousarray(samples, np.float32)
        ctx = self.lib.vox_mel_ctx_init(left_pad)
        if feeds is None:
            feeds = [len(s)]
        off = 0
        for n in feeds:
            if n > 0:
                self.lib.vox_mel_feed(ctx, _fp(s[off:off + n]), n)
            off += n
        if finish:
            self.lib.vox_mel_finish(ctx, 0)
        nf = C.c_int(0)
        p = self.lib.vox_mel_data(ctx, C.byref(nf))
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(nf.value, 128)).copy()
        self.lib.vox_mel_free(ctx)
        return out

    def load_wav(self, path):
        n = C.c_int(0)
        p = self.lib.vox_load_wav(path.encode(), C.byref(n))
        if not p:
            return None
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value,)).copy()
        self.libc.free(p)
        return out

    # ---- model-level -------------------------------------------------------
    def load(self, model_dir):
        ctx = self.lib.vox_load(model_dir.encode())
        if not ctx:
            raise RuntimeError(f"reference vox_load failed for {model_dir}")
        return ctx

    def free(self, ctx):
        self.lib.vox_free(ctx)

    def encoder_forward_incremental(self, ctx, x_new, dim):
        x = np.ascontiguousarray(x_new, np.float32)
        n = C.c_int(0)
        p = self.lib.vox_encoder_forward_incremental(ctx, _fp(x), x.shape[0], C.byref(n))
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value, dim)).copy()
        self.libc.free(p)
        return out

    def adapter_forward(self, ctx, enc_out, dec_dim):
        x = np.ascontiguousarray(enc_out, np.float32)
        n = C.c_int(0)
        p = self.lib.vox_adapter_forward(ctx, _fp(x), x.shape[0], C.byref(n))
        out = np.ctypeslib.as_array(C.cast(p, f32p), shape=(n.value, dec_dim)).copy()
        self.libc.free(p)
        return out

    def decoder_prefill(self, ctx, embeds):
        x = np.ascontiguousarray(embeds, np.float32)
        self.lib.vox_decoder_prefill(ctx, _fp(x), x.shape[0])

    def decoder_forward(self, ctx, embed, vocab):
        x = np.ascontiguousarray(embed, np.float32)
        logits = np.empty(vocab, np.float32)
        tok = self.lib.vox_decoder_forward(ctx, _fp(x), _fp(logits))
        return tok, logits

    def transcribe_stream(self, ctx, samples, feed_sizes=None, interval=None, continuous=False,
                          vocab=0, max_logit_rows=0, delay_ms=None, tap_steps=None, tap_hidden=0, tap_vectors=0):
        """Drive the reference stream API like main.c does and capture everything.

        Returns dict(tokens=[ids per decoder step], logits=[rows, vocab] or None,
        pieces=[strings surfaced by vox_stream_get]).
        """
        L = self.lib
        s = np.ascontiguousarray(samples, np.float32)
        if delay_ms is not None:
            L.vox_set_delay(ctx, int(delay_ms))
        L.voxref_hook_reset(int(vocab), int(max_logit_rows))
        if tap_steps is not None:      # residual-stream taps of a few decoder steps (oracle/ref_hooks.c)
            ts = np.ascontiguousarray(tap_steps, np.int32)
            L.voxref_tap_config(ts.ctypes.data_as(i32p), len(ts), int(tap_hidden), int(tap_vectors))
        st = L.vox_stream_init(ctx)
        if not st:
            raise RuntimeError("vox_stream_init failed")
        if interval is not None:
            L.vox_set_processing_interval(st, float(interval))
        if continuous:
            L.vox_stream_set_continuous(st, 1)
        pieces = []
        buf = (C.c_char_p * 64)()

        def drain():
            while True:
                n = L.vox_stream_get(st, buf, 64)
                if n <= 0:
                    break
                pieces.extend(buf[i].decode("utf-8", "replace") for i in range(n))

        if feed_sizes is None:
            feed_sizes = [len(s)]
        off = 0
        for n in feed_sizes:
            n = min(n, len(s) - off)
            if n <= 0:
                break
            L.vox_stream_feed(st, _fp(s[off:off + n]), n)
            off += n
            drain()
        L.vox_stream_finish(st)
        drain()
        ntok = L.voxref_hook_count()
        toks = np.ctypeslib.as_array(L.voxref_hook_tokens(), shape=(ntok,)).copy() if ntok else \
            np.zeros(0, np.int32)
        rows = L.voxref_hook_logit_rows()
        logits = None
        if vocab and rows:
            logits = np.ctypeslib.as_array(L.voxref_hook_logits(), shape=(rows, vocab)).copy()
        taps = None
        if tap_steps is not None:
            cnt = np.ctypeslib.as_array(L.voxref_tap_counts(), shape=(16,))[:len(tap_steps)].copy()
            assert (cnt == tap_vectors).all(), cnt
            taps = np.ctypeslib.as_array(L.voxref_taps(), shape=(len(tap_steps), tap_vectors, tap_hidden)).copy()
            L.voxref_tap_config(None, 0, 0, 0)
        L.vox_stream_free(st)
        return dict(tokens=toks, logits=logits, pieces=pieces, taps=taps)
