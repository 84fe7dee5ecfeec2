// vox_misc.h — row-wise / elementwise kernels of the large-M path and the audio
// front-end (all HBM-bound: coalesced float4 traffic, wave-shuffle reductions).
#pragma once
#include "vox_common.h"

namespace vox {

// RMSNorm over rows: out = x * rsqrt(mean(x^2)+eps) * w [* (1+ada)]
// (voxtral_kernels.c:346-363; ada: voxtral_decoder.c:517-524). One 256-thread block per row.
__global__ __launch_bounds__(256) void k_rmsnorm_rows(float *out, int ldo, const float *x, int ldx,
                                                      const float *w, const float *ada, int D, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float *xr = x + (size_t)row * ldx;
    float *orow = out + (size_t)row * ldo;
    float ss = 0.f;
    for (int i = tid * 4; i < D; i += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(xr + i);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    for (int i = tid * 4; i < D; i += 1024) {
        float4 v = *reinterpret_cast<const float4 *>(xr + i);
        const float4 g = *reinterpret_cast<const float4 *>(w + i);
        v.x = v.x * inv * g.x; v.y = v.y * inv * g.y; v.z = v.z * inv * g.z; v.w = v.w * inv * g.w;
        if (ada) {
            const float4 s = *reinterpret_cast<const float4 *>(ada + i);
            v.x *= (1.0f + s.x); v.y *= (1.0f + s.y); v.z *= (1.0f + s.z); v.w *= (1.0f + s.w);
        }
        *reinterpret_cast<float4 *>(orow + i) = v;
    }
}

// h[m, j] = silu(gu[m, j]) * gu[m, H + j]   (merged W1;W3 GEMM output -> SwiGLU gate)
__global__ __launch_bounds__(256) void k_silu_mul(float *h, const float *gu, int M, int H) {
    const size_t total4 = (size_t)M * H / 4;
    const int h4 = H / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t m = i / h4;
        const int j = (int)(i % h4) * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gu + m * 2 * H + j);
        const float4 u = *reinterpret_cast<const float4 *>(gu + m * 2 * H + H + j);
        float4 o;
        o.x = silu(g.x) * u.x; o.y = silu(g.y) * u.y; o.z = silu(g.z) * u.z; o.w = silu(g.w) * u.w;
        *reinterpret_cast<float4 *>(h + m * H + j) = o;
    }
}

// im2col for the causal k=3 convolutions expressed on position-major activations:
//   out[m][ic*3 + k] = src[(m*stride + k)][ic]      (column order = reference weight
//   layout [C_out, C_in*3] with index ic*3+k, voxtral_kernels.c:306-319)
// src already carries its causal history rows in front (see conv stem in the engine).
__global__ __launch_bounds__(256) void k_im2col3(float *out, const float *src, int M, int C, int stride) {
    const size_t total = (size_t)M * C * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / (3 * C);
        const int col = (int)(i % (3 * C));
        const int ic = col / 3, k = col - ic * 3;
        out[i] = src[(m * stride + k) * C + ic];
    }
}

// RoPE table for positions pos0..pos0+n-1: tab[s][d] = (cos, sin)(pos * inv_freq[d]).
__global__ __launch_bounds__(256) void k_rope_table(float *tab, const float *inv_freq, int pos0, int n,
                                                    int half_dim) {
    const int total = n * half_dim;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / half_dim, d = i - s * half_dim;
        const float ang = (float)(pos0 + s) * inv_freq[d];
        tab[2 * i] = cosf(ang);
        tab[2 * i + 1] = sinf(ang);
    }
}

// In-place interleaved-pair RoPE (voxtral_kernels.c:502-526) on the first `rope_cols`
// columns of each row of a merged QKV buffer (q heads then k heads are contiguous).
__global__ __launch_bounds__(256) void k_rope_apply(float *qkv, int ld, int n, int rope_cols, int head_dim,
                                                    const float *tab) {
    const int half = head_dim / 2;
    const int pairs_per_row = rope_cols / 2;
    const size_t total = (size_t)n * pairs_per_row;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int s = (int)(i / pairs_per_row);
        const int p = (int)(i % pairs_per_row);
        const int d = p % half;
        const float c = tab[((size_t)s * half + d) * 2], sn = tab[((size_t)s * half + d) * 2 + 1];
        float2 *v = reinterpret_cast<float2 *>(qkv + (size_t)s * ld + 2 * p);
        const float2 x = *v;
        float2 o;
        o.x = x.x * c - x.y * sn;
        o.y = x.x * sn + x.y * c;
        *v = o;
    }
}

// Copy rows [row0, row0+n) of (k | v) out of a merged QKV buffer into position-indexed
// rings: ring[(pos0 + i) % cap][:] .
__global__ __launch_bounds__(256) void k_ring_append(float *kring, float *vring, int cap, int kvd,
                                                     const float *qkv, int ld, int k_off, int v_off,
                                                     int row0, int n, int pos0) {
    const int d4 = kvd / 4;
    const size_t total = (size_t)n * d4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / d4);
        const int c = (int)(i % d4) * 4;
        const int slot = (pos0 + r) % cap;
        const float *src = qkv + (size_t)(row0 + r) * ld;
        *reinterpret_cast<float4 *>(kring + (size_t)slot * kvd + c) = *reinterpret_cast<const float4 *>(src + k_off + c);
        *reinterpret_cast<float4 *>(vring + (size_t)slot * kvd + c) = *reinterpret_cast<const float4 *>(src + v_off + c);
    }
}

// Prompt embeddings: out[i] = adapter[i] + f32(tok_emb[i == 0 ? bos : pad]) (voxtral.c:993-999)
__global__ __launch_bounds__(256) void k_embed_prompt(float *out, const float *adapter, const uint16_t *tok_emb,
                                                      int n, int dim, int bos, int pad) {
    const size_t total = (size_t)n * dim;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / dim), c = (int)(i % dim);
        const int tok = r == 0 ? bos : pad;
        out[i] = adapter[i] + bf16_to_f32(tok_emb[(size_t)tok * dim + c]);
    }
}

// Log-mel frames (mel_compute_available, voxtral_audio.c:454-513): one block per frame.
//   windowed[n] = samples[t*160+n]*hann[n]; direct 400-point DFT for k = 0..200;
//   power = re^2+im^2; mel = filt . power; log10 clamp; (v+4)/4.
// Tables are passed transposed ([n][k] and [k][m]) so table reads coalesce across k / m.
constexpr int MEL_NFFT = 400, MEL_NFREQ = 201, MEL_HOP = 160;
__global__ __launch_bounds__(256) void k_mel_frames(float *mel, int mel_bins, const float *samples,
                                                    const float *hann, const float *cosT, const float *sinT,
                                                    const float *filtT) {
    __shared__ float win[MEL_NFFT];
    __shared__ float power[MEL_NFREQ + 3];
    const int t = blockIdx.x, tid = threadIdx.x;
    const float *s = samples + (size_t)t * MEL_HOP;
    for (int i = tid; i < MEL_NFFT; i += 256) win[i] = s[i] * hann[i];
    __syncthreads();
    if (tid < MEL_NFREQ) {
        float re = 0.f, im = 0.f;
        for (int n = 0; n < MEL_NFFT; n++) {
            const float w = win[n];
            re = fmaf(w, cosT[n * MEL_NFREQ + tid], re);
            im = fmaf(w, sinT[n * MEL_NFREQ + tid], im);
        }
        power[tid] = re * re + im * im;
    }
    __syncthreads();
    if (tid < mel_bins) {
        float sum = 0.f;
        for (int k = 0; k < MEL_NFREQ; k++) sum = fmaf(filtT[k * mel_bins + tid], power[k], sum);
        if (sum < 1e-10f) sum = 1e-10f;
        float val = log10f(sum);
        const float min_val = 1.5f - 8.0f;     // VOX_LOG_MEL_MAX - 8 (voxtral_audio.c:504)
        if (val < min_val) val = min_val;
        mel[(size_t)t * mel_bins + tid] = (val + 4.0f) / 4.0f;
    }
}

// Start-up self-test helper: row16_sum via DPP and via shuffles on the same data.
__global__ void k_dpp_selftest(float *out_dpp, float *out_shfl) {
    const int lane = threadIdx.x & 63;
    const float v = (float)((lane * 37 + 11) % 101) * 0.25f + (float)lane;
    out_dpp[lane] = row16_sum<true>(v);
    out_shfl[lane] = row16_sum<false>(v);
}

}  // namespace vox
