"""oracle/vox_oracle.py — CPU restatement (numpy, float32) of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the *checker*; never by the product (voxtral_c_amd), which has no CPU
path at all.

Every function restates one reference function and cites it (file:line into
antirez/voxtral.c).  It deliberately keeps the reference's *data-structure semantics*
(physical KV caches with compaction, conv tails, residual rows) rather than the logical
-position formulation the HIP engine uses, so that agreement between the two is evidence
and not a tautology.

Pinned (tests/test_oracle_pinning.py) against the real reference compiled from its own
sources into oracle/_ref (oracle/Makefile) and against the golden fixtures under
tests/golden/ that were generated from that build (tools/make_golden.py).
"""
import numpy as np

F = np.float32


# ------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------
class Dims:
    """Model constants (voxtral.h:19-50). Presets match tools/synth_model.c and oracle/Makefile."""

    def __init__(self, **kw):
        self.mel_bins = 128
        self.enc_dim, self.enc_layers, self.enc_heads, self.enc_head_dim = 1280, 32, 32, 64
        self.enc_hidden, self.enc_window = 5120, 750
        self.dec_dim, self.dec_layers, self.dec_heads, self.dec_kv_heads = 3072, 26, 32, 8
        self.dec_head_dim, self.dec_hidden, self.dec_window = 128, 9216, 8192
        self.vocab, self.ada_dim = 131072, 32
        self.enc_eps = self.dec_eps = 1e-5
        self.rope_theta = 1000000.0
        self.__dict__.update(kw)


PRESETS = {
    "full": Dims(),
    "small": Dims(enc_layers=2, dec_layers=2, vocab=4096),
    "tiny": Dims(enc_dim=256, enc_layers=3, enc_heads=4, enc_hidden=512, enc_window=48,
                 dec_dim=384, dec_layers=3, dec_heads=8, dec_kv_heads=2, dec_hidden=768,
                 dec_window=64, vocab=2048),
    "deep": Dims(enc_dim=256, enc_heads=4, enc_hidden=512, dec_dim=384, dec_heads=8, dec_kv_heads=2,
                 dec_hidden=768, vocab=8192),
}


# ------------------------------------------------------------------------------------
# bf16 helpers
# ------------------------------------------------------------------------------------
def bf16_to_f32(u16):
    """Exact upcast, as bf16_to_f32_buf (voxtral_kernels.c:124-128)."""
    return (np.asarray(u16, np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    """Round-to-nearest-even (used only to fabricate test weights)."""
    u = np.asarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


# ------------------------------------------------------------------------------------
# kernels (voxtral_kernels.c)
# ------------------------------------------------------------------------------------
def linear_bf16(x, w_bf16, bias=None):
    """vox_linear_bf16 / vox_linear_nobias_bf16 (voxtral_kernels.c:197-240): y = x W^T + b."""
    y = np.asarray(x, F) @ bf16_to_f32(w_bf16).T
    if bias is not None:
        y = y + np.asarray(bias, F)
    return y.astype(F)


def rms_norm(x, w, eps):
    """vox_rms_norm (voxtral_kernels.c:346-363)."""
    x = np.asarray(x, F)
    ms = (x * x).sum(-1, keepdims=True, dtype=F) / F(x.shape[-1]) + F(eps)
    inv = F(1.0) / np.sqrt(ms, dtype=F)
    return (x * inv * np.asarray(w, F)).astype(F)


def silu(x):
    """vox_silu (voxtral_kernels.c:369-374)."""
    x = np.asarray(x, F)
    return (x / (F(1.0) + np.exp(-x, dtype=F))).astype(F)


def gelu(x):
    """vox_gelu, tanh approximation (voxtral_kernels.c:376-384)."""
    x = np.asarray(x, F)
    inner = F(0.7978845608028654) * (x + F(0.044715) * x * x * x)
    return (F(0.5) * x * (F(1.0) + np.tanh(inner, dtype=F))).astype(F)


def _powf(base, e):
    """glibc powf, element-wise (numpy's float32 power is not always the same ulp)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return np.array([libm.powf(float(base), float(x)) for x in np.asarray(e, F)], F)


def rope_freqs(pos, dim, theta):
    """vox_compute_rope_freqs (voxtral_kernels.c:488-500): [seq, dim/2, (cos, sin)]."""
    pos = np.asarray(pos, np.int64).astype(F)
    d = np.arange(dim // 2, dtype=F)
    # 1/powf(theta, e) as the reference's -ffast-math build evaluates it: powf(theta, -e)
    # (bit-identical tables vs oracle/_ref; the literal form is an ulp off in places).
    freq = _powf(theta, -((F(2.0) * d) / F(dim)))
    ang = (pos[:, None] * freq[None, :]).astype(F)
    return np.stack([np.cos(ang, dtype=F), np.sin(ang, dtype=F)], -1)


def apply_rope(x, freqs, heads, head_dim):
    """vox_apply_rope, interleaved pairs (2d, 2d+1) (voxtral_kernels.c:502-526)."""
    seq = x.shape[0]
    v = np.asarray(x, F).reshape(seq, heads, head_dim // 2, 2)
    c = freqs[:, None, :, 0]
    s = freqs[:, None, :, 1]
    out = np.empty_like(v)
    out[..., 0] = v[..., 0] * c - v[..., 1] * s
    out[..., 1] = v[..., 0] * s + v[..., 1] * c
    return out.reshape(seq, heads * head_dim).astype(F)


def causal_attention(q, k, v, n_heads, n_kv_heads, head_dim, scale, window, q_offset):
    """vox_causal_attention (voxtral_kernels.c:412-482): per head and query i at global position
    g = q_offset+i, keys max(0, g-window+1) .. min(g, seq_k-1) of the *physical* K/V arrays."""
    q = np.asarray(q, F); k = np.asarray(k, F); v = np.asarray(v, F)
    seq_q, seq_k = q.shape[0], k.shape[0]
    out = np.zeros_like(q)
    hpk = n_heads // n_kv_heads
    for h in range(n_heads):
        kv = h // hpk
        qh = q[:, h * head_dim:(h + 1) * head_dim]
        kh = k[:, kv * head_dim:(kv + 1) * head_dim]
        vh = v[:, kv * head_dim:(kv + 1) * head_dim]
        s = (qh @ kh.T) * F(scale)
        g = q_offset + np.arange(seq_q)[:, None]
        j = np.arange(seq_k)[None, :]
        mask = (j <= g)
        if window > 0:
            mask &= (j >= g - window + 1)
        s = np.where(mask, s, -np.inf)
        m = s.max(-1, keepdims=True)
        m = np.where(np.isfinite(m), m, 0.0)
        p = np.exp(s - m, dtype=F)
        p = np.where(mask, p, 0).astype(F)
        den = p.sum(-1, keepdims=True, dtype=F)
        o = p @ vh
        out[:, h * head_dim:(h + 1) * head_dim] = np.where(den > 0, o / np.where(den > 0, den, 1), 0)
    return out.astype(F)


def causal_conv1d(x_cl, w, b, stride):
    """vox_causal_conv1d (voxtral_kernels.c:293-340), k=3: in [C_in, L], weight [C_out, C_in*3]
    (column ic*3+k), left pad = 3 - stride, out length ceil((L-3+pad)/stride + 1)."""
    x_cl = np.asarray(x_cl, F)
    cin, length = x_cl.shape
    pad = 3 - stride
    out_len = int(np.ceil((length - 3 + pad) / stride + 1.0))
    if out_len <= 0:
        return np.zeros((w.shape[0], 0), F)
    xp = np.zeros((cin, pad + length + 3), F)
    xp[:, pad:pad + length] = x_cl
    idx = np.arange(out_len) * stride
    cols = np.stack([xp[:, idx + kk] for kk in range(3)], 1)       # [cin, 3, out_len]
    y = np.asarray(w, F) @ cols.reshape(cin * 3, out_len)
    return (y + np.asarray(b, F)[:, None]).astype(F)


# ------------------------------------------------------------------------------------
# mel (voxtral_audio.c)
# ------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = F(f)
    if f >= F(1000.0):
        return F(15.0) + np.log(f / F(1000.0), dtype=F) * (F(27.0) / np.log(F(6.4), dtype=F))
    return F(3.0) * f / F(200.0)


def _mel_to_hz(m):
    m = F(m)
    if m >= F(15.0):
        return F(1000.0) * np.exp((np.log(F(6.4), dtype=F) / F(27.0)) * (m - F(15.0)), dtype=F)
    return F(200.0) * m / F(3.0)


def mel_filters(n_mel=128, n_freq=201, sr=16000):
    """build_mel_filters, Slaney (voxtral_audio.c:248-285)."""
    fft = (np.arange(n_freq, dtype=F) * (F(sr) / F(2.0)) / F(n_freq - 1)).astype(F)
    m0, m1 = _hz_to_mel(0.0), _hz_to_mel(sr / 2.0)
    edges = np.array([_mel_to_hz(m0 + (m1 - m0) * F(i) / F(n_mel + 1)) for i in range(n_mel + 2)], F)
    diff = (edges[1:] - edges[:-1]).astype(F)
    diff[diff == 0] = F(1e-6)
    filt = np.zeros((n_mel, n_freq), F)
    for m in range(n_mel):
        enorm = F(2.0) / (edges[m + 2] - edges[m])
        down = (fft - edges[m]) / diff[m]
        up = (edges[m + 2] - fft) / diff[m + 1]
        filt[m] = np.maximum(np.minimum(down, up), 0) * enorm
    return filt


_MEL_CACHE = {}


def _mel_tables():
    if not _MEL_CACHE:
        n = np.arange(400, dtype=F)
        _MEL_CACHE["hann"] = (F(0.5) * (F(1.0) - np.cos(F(2.0) * F(np.pi) * n / F(400.0), dtype=F))).astype(F)
        k = np.arange(201, dtype=F)[:, None]
        # The reference's -ffast-math build folds 2*pi/400 into one f32 constant and evaluates
        # angle = (c*k)*n (verified bit-for-bit against the tables inside oracle/_ref; other
        # association orders move table entries by up to 2.4e-4 because k*n reaches 8e4 rad/2pi).
        step = F(2.0) * F(np.pi) / F(400.0)
        ang = ((step * k).astype(F) * n[None, :]).astype(F)
        _MEL_CACHE["cos"] = np.cos(ang, dtype=F)
        _MEL_CACHE["sin"] = np.sin(ang, dtype=F)
        _MEL_CACHE["filt"] = mel_filters()
    return _MEL_CACHE


def mel_frames(padded, n_frames):
    """mel_compute_available (voxtral_audio.c:454-513): frame t = padded[160t : 160t+400]."""
    t = _mel_tables()
    idx = np.arange(n_frames)[:, None] * 160 + np.arange(400)[None, :]
    win = (np.asarray(padded, F)[idx] * t["hann"]).astype(F)
    re = win @ t["cos"].T
    im = win @ t["sin"].T
    power = (re * re + im * im).astype(F)
    mel = power @ t["filt"].T
    mel = np.maximum(mel, F(1e-10))
    val = np.log10(mel, dtype=F)
    val = np.maximum(val, F(1.5 - 8.0))
    return ((val + F(4.0)) / F(4.0)).astype(F)


def mel_stream(samples, left_pad=32 * 1280, right_pad_feed=0, finish=True):
    """What vox_stream_feed/flush/finish make of the audio (voxtral_audio.c:515-633,
    voxtral.c:1588-1614): zeros(200+left_pad) | samples | zeros(right_pad_feed) then, on finish,
    a 200-sample reflection of the tail, and the last frame dropped."""
    buf = np.concatenate([np.zeros(200 + left_pad, F), np.asarray(samples, F), np.zeros(right_pad_feed, F)])
    if finish:
        real_end = len(buf)
        refl = buf[real_end - 2 - np.arange(200)]
        buf = np.concatenate([buf, refl])
    n = (len(buf) - 400) // 160 + 1 if len(buf) >= 400 else 0
    if finish:
        n -= 1
    return mel_frames(buf, n) if n > 0 else np.zeros((0, 128), F)


# ------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------
ENC_PFX = "mm_streams_embeddings.embedding_module.whisper_encoder"
EMB_PFX = "mm_streams_embeddings.embedding_module"


def read_safetensors_bf16(path):
    """Minimal reader: returns {name: uint16 array (bf16 bits) with its shape}."""
    import json
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    hlen = int(np.frombuffer(raw[:8].tobytes(), np.uint64)[0])
    hdr = json.loads(raw[8:8 + hlen].tobytes())
    out = {}
    base = 8 + hlen
    for name, t in hdr.items():
        if name == "__metadata__":
            continue
        a, b = t["data_offsets"]
        out[name] = np.frombuffer(raw[base + a:base + b], np.uint16).reshape(t["shape"])
    return out


class Weights:
    """Binding of tensor names to roles (voxtral_encoder.c:50-117, voxtral_decoder.c:49-108,
    voxtral.c:102-110). Big matrices stay bf16 bits; small ones are upcast (load_f32)."""

    def __init__(self, model_dir, dims):
        import os
        self.t = read_safetensors_bf16(os.path.join(model_dir, "consolidated.safetensors"))
        self.d = dims

    def bf(self, name):
        return self.t[name]

    def f32(self, name):
        return bf16_to_f32(self.t[name])

    def enc(self, i, sfx):
        return f"{ENC_PFX}.transformer.layers.{i}.{sfx}"


# ------------------------------------------------------------------------------------
# model state + forwards
# ------------------------------------------------------------------------------------
class Oracle:
    def __init__(self, model_dir, dims):
        self.d = dims
        self.w = Weights(model_dir, dims)
        self.delay_tokens = 6
        self.reset_encoder()
        self.reset_decoder()
        self.update_time_conditioning()

    # -- time conditioning (voxtral.c:31-80) --
    def update_time_conditioning(self):
        d = self.d
        half = d.dec_dim // 2
        i = np.arange(half, dtype=F)
        inv = np.exp(-np.log(F(10000.0), dtype=F) * i / F(half), dtype=F)
        emb = (F(self.delay_tokens) * inv).astype(F)
        t_cond = np.concatenate([np.cos(emb, dtype=F), np.sin(emb, dtype=F)])
        self.ada_scale = []
        for l in range(d.dec_layers):
            down = self.w.f32(f"layers.{l}.ada_rms_norm_t_cond.0.weight")
            up = self.w.f32(f"layers.{l}.ada_rms_norm_t_cond.2.weight")
            self.ada_scale.append((up @ gelu(down @ t_cond)).astype(F))

    # -- conv stem, streaming with tails (stream_conv_stem, voxtral.c:537-715) --
    def reset_encoder(self):
        d = self.d
        self.conv_init = False
        self.mel_tail = np.zeros((2, d.mel_bins), F)
        self.conv0_tail = np.zeros((2, d.enc_dim), F)
        self.conv0_residual = None
        self.enc_k = [np.zeros((0, d.enc_heads * d.enc_head_dim), F) for _ in range(d.enc_layers)]
        self.enc_v = [np.zeros((0, d.enc_heads * d.enc_head_dim), F) for _ in range(d.enc_layers)]
        self.enc_pos_offset = 0
        self.enc_residual = np.zeros((0, d.enc_dim), F)

    def conv_stem(self, mel_new):
        d = self.d
        w0 = self.w.f32(f"{ENC_PFX}.conv_layers.0.conv.weight").reshape(d.enc_dim, -1)
        b0 = self.w.f32(f"{ENC_PFX}.conv_layers.0.conv.bias")
        w1 = self.w.f32(f"{ENC_PFX}.conv_layers.1.conv.weight").reshape(d.enc_dim, -1)
        b1 = self.w.f32(f"{ENC_PFX}.conv_layers.1.conv.bias")
        mel_new = np.asarray(mel_new, F)
        n = mel_new.shape[0]
        if n <= 0:
            return np.zeros((0, d.enc_dim), F)
        first = not self.conv_init
        if first:
            c0 = gelu(causal_conv1d(mel_new.T, w0, b0, 1)).T          # [n, enc_dim]
            self.conv_init = True
        else:
            full = gelu(causal_conv1d(np.concatenate([self.mel_tail, mel_new]).T, w0, b0, 1)).T
            c0 = full[2:]
        tail = np.zeros((2, d.mel_bins), F)
        tc = min(2, n)
        tail[2 - tc:] = mel_new[n - tc:]       # n == 1 zeroes the older slot, like the reference
        self.mel_tail = tail
        prev = 0 if self.conv0_residual is None else 1
        total = prev + n
        new_res = total & 1
        feed_from_new = n - new_res
        feed_total = prev + feed_from_new
        if feed_total <= 0:
            if new_res and n > 0:
                self.conv0_residual = c0[-1].copy()
            else:
                self.conv0_residual = None
            return np.zeros((0, d.enc_dim), F)
        feed = c0[:feed_from_new] if prev == 0 else np.concatenate([self.conv0_residual[None], c0[:feed_from_new]])
        self.conv0_residual = c0[-1].copy() if new_res else None
        if first:
            conv1_in, discard = feed, 0
        else:
            conv1_in, discard = np.concatenate([self.conv0_tail, feed]), 1
        self.conv0_tail = feed[-2:].copy()
        out = gelu(causal_conv1d(conv1_in.T, w1, b1, 2)).T
        return out[discard:].astype(F)

    # -- incremental encoder (vox_encoder_forward_incremental, voxtral_encoder.c:452-636) --
    def encoder_incremental(self, x_new):
        d = self.d
        x = np.asarray(x_new, F).copy()
        n = x.shape[0]
        if n <= 0:
            return x
        hd, heads, W = d.enc_head_dim, d.enc_heads, d.enc_window
        cache_len = self.enc_k[0].shape[0]
        if cache_len + n > W and cache_len > W:      # enc_kv_cache_compact (:388-406)
            discard = cache_len - W
            for l in range(d.enc_layers):
                self.enc_k[l] = self.enc_k[l][discard:]
                self.enc_v[l] = self.enc_v[l][discard:]
            self.enc_pos_offset += discard
            cache_len = W
        pos = self.enc_pos_offset + cache_len + np.arange(n)
        fr = rope_freqs(pos, hd, d.rope_theta)
        scale = 1.0 / np.sqrt(F(hd))
        w = self.w
        for l in range(d.enc_layers):
            xn = rms_norm(x, w.f32(w.enc(l, "attention_norm.weight")), d.enc_eps)
            q = linear_bf16(xn, w.bf(w.enc(l, "attention.wq.weight")), w.f32(w.enc(l, "attention.wq.bias")))
            k = linear_bf16(xn, w.bf(w.enc(l, "attention.wk.weight")))
            v = linear_bf16(xn, w.bf(w.enc(l, "attention.wv.weight")), w.f32(w.enc(l, "attention.wv.bias")))
            q = apply_rope(q, fr, heads, hd)
            k = apply_rope(k, fr, heads, hd)
            self.enc_k[l] = np.concatenate([self.enc_k[l], k])
            self.enc_v[l] = np.concatenate([self.enc_v[l], v])
            a = causal_attention(q, self.enc_k[l], self.enc_v[l], heads, heads, hd, scale, W, cache_len)
            x = x + linear_bf16(a, w.bf(w.enc(l, "attention.wo.weight")), w.f32(w.enc(l, "attention.wo.bias")))
            xn = rms_norm(x, w.f32(w.enc(l, "ffn_norm.weight")), d.enc_eps)
            g = silu(linear_bf16(xn, w.bf(w.enc(l, "feed_forward.w1.weight"))))
            u = linear_bf16(xn, w.bf(w.enc(l, "feed_forward.w3.weight")))
            x = x + linear_bf16(g * u, w.bf(w.enc(l, "feed_forward.w2.weight")), w.f32(w.enc(l, "feed_forward.w2.bias")))
        return rms_norm(x, w.f32(f"{ENC_PFX}.transformer.norm.weight"), d.enc_eps)

    # -- adapter (vox_adapter_forward, voxtral_encoder.c:642-674) --
    def adapter(self, enc_out):
        d = self.d
        m = enc_out.shape[0] // 4
        ds = np.asarray(enc_out, F)[:m * 4].reshape(m, 4 * d.enc_dim)
        mid = gelu(linear_bf16(ds, self.w.bf(f"{EMB_PFX}.audio_language_projection.0.weight")))
        return linear_bf16(mid, self.w.bf(f"{EMB_PFX}.audio_language_projection.2.weight"))

    # -- stream_run_encoder glue (voxtral.c:783-907): conv -> encoder -> 4x alignment -> adapter --
    def stream_encode(self, mel_new):
        x = self.conv_stem(mel_new)
        if x.shape[0] == 0:
            return np.zeros((0, self.d.dec_dim), F)
        e = self.encoder_incremental(x)
        allrows = np.concatenate([self.enc_residual, e])
        usable = (allrows.shape[0] // 4) * 4
        self.enc_residual = allrows[usable:].copy()
        if usable == 0:
            return np.zeros((0, self.d.dec_dim), F)
        return self.adapter(allrows[:usable])

    # -- decoder (voxtral_decoder.c) --
    def reset_decoder(self):
        d = self.d
        kvd = d.dec_kv_heads * d.dec_head_dim
        self.dec_k = [np.zeros((0, kvd), F) for _ in range(d.dec_layers)]
        self.dec_v = [np.zeros((0, kvd), F) for _ in range(d.dec_layers)]
        self.kv_pos_offset = 0
        self.kv_cache_max = 0

    def _dec_layers(self, x, start_pos, logical_start):
        d = self.d
        n = x.shape[0]
        hd = d.dec_head_dim
        fr = rope_freqs(logical_start + np.arange(n), hd, d.rope_theta)
        scale = 1.0 / np.sqrt(F(hd))
        for l in range(d.dec_layers):
            p = f"layers.{l}."
            xn = rms_norm(x, self.w.f32(p + "attention_norm.weight"), d.dec_eps)
            q = linear_bf16(xn, self.w.bf(p + "attention.wq.weight"))
            k = linear_bf16(xn, self.w.bf(p + "attention.wk.weight"))
            v = linear_bf16(xn, self.w.bf(p + "attention.wv.weight"))
            q = apply_rope(q, fr, d.dec_heads, hd)
            k = apply_rope(k, fr, d.dec_kv_heads, hd)
            self.dec_k[l] = np.concatenate([self.dec_k[l][:start_pos], k])
            self.dec_v[l] = np.concatenate([self.dec_v[l][:start_pos], v])
            a = causal_attention(q, self.dec_k[l], self.dec_v[l], d.dec_heads, d.dec_kv_heads, hd, scale,
                                 d.dec_window, start_pos)
            x = x + linear_bf16(a, self.w.bf(p + "attention.wo.weight"))
            xn = rms_norm(x, self.w.f32(p + "ffn_norm.weight"), d.dec_eps)
            xn = (xn * (F(1.0) + self.ada_scale[l])).astype(F)
            g = silu(linear_bf16(xn, self.w.bf(p + "feed_forward.w1.weight")))
            u = linear_bf16(xn, self.w.bf(p + "feed_forward.w3.weight"))
            x = x + linear_bf16(g * u, self.w.bf(p + "feed_forward.w2.weight"))
        return x

    def decoder_prefill(self, embeds):
        """vox_decoder_prefill (voxtral_decoder.c:410-558)."""
        x = np.asarray(embeds, F).copy()
        start = self.dec_k[0].shape[0]
        if self.kv_cache_max == 0:
            self.kv_cache_max = self.d.dec_window + x.shape[0] + 1024
        self._dec_layers(x, start, self.kv_pos_offset + start)

    def decoder_forward(self, embed):
        """vox_decoder_forward (voxtral_decoder.c:586-706) -> (token, logits)."""
        d = self.d
        if self.kv_cache_max == 0:
            self.kv_cache_max = d.dec_window + 1 + 1024
        pos = self.dec_k[0].shape[0]
        if pos >= self.kv_cache_max and pos > d.dec_window:      # kv_cache_compact (:317-347)
            discard = pos - d.dec_window
            for l in range(d.dec_layers):
                self.dec_k[l] = self.dec_k[l][discard:]
                self.dec_v[l] = self.dec_v[l][discard:]
            self.kv_pos_offset += discard
            pos = d.dec_window
        x = self._dec_layers(np.asarray(embed, F)[None, :].copy(), pos, self.kv_pos_offset + pos)
        x = rms_norm(x, self.w.f32("norm.weight"), d.dec_eps)
        logits = linear_bf16(x, self.w.bf(f"{EMB_PFX}.tok_embeddings.weight"))[0]
        return int(np.argmax(logits)), logits      # np.argmax: first maximum = lowest index (:697-704)

    def tok_embed(self, tok):
        return bf16_to_f32(self.w.bf(f"{EMB_PFX}.tok_embeddings.weight")[tok])

    # -- whole-stream transcription (vox_stream_feed/finish offline case, voxtral.c:969-1265) --
    def transcribe(self, samples, max_steps=None):
        """One feed of all samples + finish, non-continuous, n_alt = 1. Returns (tokens, logits)."""
        self.reset_encoder(); self.reset_decoder()
        n = len(samples)
        align = (1280 - n % 1280) % 1280
        right = align + ((self.delay_tokens + 1) + 10) * 1280
        # feed(): all frames that fit before the flush padding; flush(): the padded rest; finish(): tail
        mel_all = mel_stream(samples, right_pad_feed=right, finish=True)
        mel_feed = mel_stream(samples, right_pad_feed=0, finish=False)
        mel_flush = mel_stream(samples, right_pad_feed=right, finish=False)
        n1, n2 = mel_feed.shape[0], mel_flush.shape[0]
        adapter = []
        if n1 >= 312:
            adapter.append(self.stream_encode(mel_all[:n1]))
            if n2 > n1:
                adapter.append(self.stream_encode(mel_all[n1:n2]))
        else:
            adapter.append(self.stream_encode(mel_all[:n2]))
        adapter.append(self.stream_encode(mel_all[n2:]))
        A = np.concatenate(adapter)
        prompt_len = 1 + 32 + self.delay_tokens
        toks, logs = [], []
        if A.shape[0] < prompt_len:
            return np.zeros(0, np.int32), np.zeros((0, self.d.vocab), F)
        emb = np.stack([A[i] + self.tok_embed(1 if i == 0 else 32) for i in range(prompt_len)])
        self.decoder_prefill(emb[:-1])
        tok, lg = self.decoder_forward(emb[-1])
        toks.append(tok); logs.append(lg)
        pos = prompt_len
        while pos < A.shape[0] and tok != 2 and (max_steps is None or len(toks) < max_steps):
            tok, lg = self.decoder_forward(A[pos] + self.tok_embed(tok))
            toks.append(tok); logs.append(lg)
            pos += 1
        return np.array(toks, np.int32), np.stack(logs)
