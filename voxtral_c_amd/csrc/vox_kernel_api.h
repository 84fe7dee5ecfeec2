// vox_kernel_api.h — device side of the reference's kernel-level API (voxtral_kernels.h:18-159).
//
// The reference exports its CPU math kernels (vox_linear_bf16, vox_rms_norm, vox_causal_attention,
// vox_apply_rope, ...) and SURVEY 8(b) keeps them as part of the drop-in surface: an embedder that
// calls one of them must link and get the same numbers.  They are shape-generic, so next to the
// specialised production kernels (k_gemm_mfma_*, k_gemv*, k_attn_*) this file holds plain generic
// ones: an f32 x f32 tiled GEMM with arbitrary B strides (vox_matmul / vox_matmul_t / vox_linear and,
// through an im2col, vox_conv1d / vox_causal_conv1d), elementwise ops, row softmax, a generic
// causal attention for head geometries the production kernels do not cover, and RoPE tables.
// None of this is on the streaming hot path; it is a correctness surface (tests/test_gpu_kernels_api.py
// checks every function against oracle/_ref).  Included at the end of vox_hip_engine.hip.
#pragma once
#include "vox_common.h"

namespace vox {

enum { EW_ADD = 0, EW_MUL, EW_AXPY, EW_SCALE, EW_SILU, EW_GELU };

// a[i] = op(a[i], b[i], s)   (voxtral_kernels.c:29-47, 369-384)
__global__ __launch_bounds__(256) void k_eltwise(float *a, const float *b, float s, size_t n, int op) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = a[i];
        switch (op) {
            case EW_ADD: v += b[i]; break;
            case EW_MUL: v *= b[i]; break;
            case EW_AXPY: v += s * b[i]; break;
            case EW_SCALE: v *= s; break;
            case EW_SILU: v = silu(v); break;
            default: v = gelu_tanh(v); break;
        }
        a[i] = v;
    }
}

// Row softmax in place (voxtral_kernels.c:386-406): one block per row, max / sum by wave + LDS reduce.
__global__ __launch_bounds__(256) void k_softmax_rows(float *x, int cols) {
    __shared__ float red[4];
    float *row = x + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x;
    float m = -3.0e38f;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, row[c]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += 256) { const float e = expf(row[c] - m); row[c] = e; s += e; }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = tid; c < cols; c += 256) row[c] *= inv;
}

// C[m, n] = sum_k A[m*lda + k] * B[k*sbk + n*sbn] (+ bias[n] | + rbias[m]).  f32 in, f32 FMA accumulate in k order
// within a 16-wide slice.  64 x 64 output tile, 256 threads, 4 x 4 outputs per thread, K slices of 16
// through LDS.  (sbk, sbn) = (N, 1) for B [K, N]  (vox_matmul), (1, K) for B [N, K] (vox_matmul_t, vox_linear).
__global__ __launch_bounds__(256) void k_sgemm(float *C, int ldc, const float *A, int lda, const float *B, long sbk, long sbn,
                                               int M, int N, int K, const float *bias, const float *rbias) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = tid; i < 64 * 16; i += 256) {
            {   // A tile: consecutive threads walk k (contiguous in A)
                const int kk = i & 15, mm = i >> 4;
                const int m = m0 + mm, k = k0 + kk;
                As[kk][mm] = (m < M && k < K) ? A[(size_t)m * lda + k] : 0.f;
            }
            {   // B tile: walk the contiguous dimension of B first
                int kk, nn;
                if (sbn == 1) { nn = i & 63; kk = i >> 6; } else { kk = i & 15; nn = i >> 4; }
                const int n = n0 + nn, k = k0 + kk;
                Bs[kk][nn] = (n < N && k < K) ? B[(size_t)k * sbk + (size_t)n * sbn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + tx * 4 + j;
            if (n < N) C[(size_t)m * ldc + n] = acc[i][j] + (bias ? bias[n] : 0.f) + (rbias ? rbias[m] : 0.f);
        }
    }
}

// im2col of a channel-major signal in[C_in, L] for a 1-D convolution:
//   col[(ic*ks + k), ol] = in[ic, ol*stride - pad_left + k]  (0 outside)   (voxtral_kernels.c:306-319)
__global__ __launch_bounds__(256) void k_conv_im2col(float *col, const float *in, int C_in, int L, int ks, int stride,
                                                     int pad_left, int L_out) {
    const size_t total = (size_t)C_in * ks * L_out;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ol = (int)(i % L_out);
        const int row = (int)(i / L_out);
        const int ic = row / ks, k = row - ic * ks;
        const int il = ol * stride - pad_left + k;
        col[i] = (il >= 0 && il < L) ? in[(size_t)ic * L + il] : 0.f;
    }
}

// freqs[s][d] = (cos, sin)(pos[s] * inv_freq[d])   (vox_compute_rope_freqs, voxtral_kernels.c:488-500)
__global__ __launch_bounds__(256) void k_rope_freqs(float *freqs, const int *pos, int seq, int half, const float *inv_freq) {
    const int total = seq * half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / half, d = i - s * half;
        const float ang = (float)pos[s] * inv_freq[d];
        freqs[2 * i] = cosf(ang);
        freqs[2 * i + 1] = sinf(ang);
    }
}

// Generic vox_causal_attention (voxtral_kernels.c:412-482) for any head_dim <= 256 and any GQA ratio:
// one 64-lane wave per (query, head); lanes split the head dimension (4 floats each at most), keys
// are visited in order with the reference's online-softmax recurrence.  Used only for geometries the
// production kernels (head_dim 64 MHA / head_dim 128 with 4 q heads per kv head) do not cover.
__global__ __launch_bounds__(64) void k_attn_generic(float *out, const float *Q, const float *K, const float *V, int seq_q,
                                                     int seq_k, int n_heads, int n_kv_heads, int hd, float scale, int window,
                                                     int q_offset) {
    const int qi = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int kvh = h / (n_heads / n_kv_heads);
    const int qd = n_heads * hd, kvd = n_kv_heads * hd;
    const float *q = Q + (size_t)qi * qd + (size_t)h * hd;
    float qv[4], o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; qv[j] = d < hd ? q[d] : 0.f; }
    const int gpos = q_offset + qi;
    int k0 = 0;
    if (window > 0) k0 = max(0, gpos - window + 1);
    const int k1 = min(gpos, seq_k - 1);
    float m = -1e30f, l = 0.f;
    for (int kk = k0; kk <= k1; kk++) {
        const float *kr = K + (size_t)kk * kvd + (size_t)kvh * hd;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; if (d < hd) dot = fmaf(qv[j], kr[d], dot); }
        dot = wave_sum(dot) * scale;
        const float mn = fmaxf(m, dot);
        const float corr = expf(m - mn), p = expf(dot - mn);
        const float *vr = V + (size_t)kk * kvd + (size_t)kvh * hd;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; if (d < hd) o[j] = o[j] * corr + p * vr[d]; }
        l = l * corr + p;
        m = mn;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float *orow = out + (size_t)qi * qd + (size_t)h * hd;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int d = lane + 64 * j; if (d < hd) orow[d] = o[j] * inv; }
}

}  // namespace vox

namespace vox {
// RMSNorm for any row width (the production k_rmsnorm_rows needs hidden % 4 == 0).
__global__ __launch_bounds__(256) void k_rmsnorm_generic(float *out, const float *x, const float *w, int D, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float *xr = x + (size_t)blockIdx.x * D;
    float ss = 0.f;
    for (int i = tid; i < D; i += 256) ss += xr[i] * xr[i];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    for (int i = tid; i < D; i += 256) out[(size_t)blockIdx.x * D + i] = xr[i] * inv * w[i];
}
}  // namespace vox
