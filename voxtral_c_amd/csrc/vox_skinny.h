// vox_skinny.h — transformer layer on a FEW rows (streaming encoder chunks: 25 rows at -I 0.5, 50-150 at the CLI's
// default cadence): y[n, N] = x[n, K] . W[N, K]^T for n <= 64, HBM-bound on the bf16 weights.
//
// Replaces, for small n, the large-M path of run_layer_rows (13 launches per layer: 128 x 128 MFMA tiles that are 80 %
// padding at n = 25, split-K reduce launches, separate RoPE / ring-append / SiLU / RMSNorm kernels).  Measured in round 1
// (gpurun_out/prof_stream/r1_kernel_stats.csv): ~130 us per layer for a 25-row chunk against 60.3 MB / 6.3 TB/s = 9.6 us.
// Reference: vox_encoder_forward_incremental, voxtral_encoder.c:452-636 (same arithmetic: exact bf16 weights x f32
// activations, f32 accumulation; here through the exact 3-term bf16 split of the activations on v_mfma_f32_16x16x32_bf16).
//
// One wave = one 32-row tile of W (two 16-row MFMA B operands, 16 bytes per lane straight from global memory in fragment
// layout: lane (li, kb) reads W[row0 + 16 q + li][k + 8 kb .. +7]) x all n activation rows (A operands: x[m][k + 8 kb .. +7] as
// f32 from L2, split into hi / mid / lo bf16 in registers, or pre-split planes) x a set of 64-wide K chunks.  No LDS staging and no barrier
// in the main loop: every wave streams independently with the next chunk's loads in flight under the current chunk's
// MFMAs.  The waves of a workgroup (WPB = 8) share a W tile and split K; their accumulators are added in wave order
// through LDS (deterministic), then the epilogue runs on the 32 x 32 (or 64 x 32) tile:
//   EPI_QKV     + bias, interleaved-pair RoPE from the chunk's table, q/k/v to the merged QKV buffer AND k/v straight into
//               the position-indexed rings (voxtral_encoder.c:542-567)
//   EPI_SWIGLU  two W tiles per wave (w1 rows and the matching w3 rows): h = silu(x w1^T) * (x w3^T)  (:598-612)
//   EPI_PARTIAL raw partial sums of a K split (wo, w2: only 40 row tiles, so K is split over blockIdx.y);
// k_rows_finish then adds the partials in split order + bias + residual and applies the NEXT RMSNorm in the same launch
// (voxtral_encoder.c:585-596, 614-628).  Per layer: qkv, attention (+ combine), wo, finish, w1;w3, w2, finish = 7-8
// launches instead of 13.
#pragma once
#include "vox_common.h"
#include "vox_gemm.h"

namespace vox {

enum { SK_PARTIAL = 0, SK_QKV = 1, SK_SWIGLU = 2 };
constexpr int SK_WPB = 8;                 // waves per workgroup (K splitters of one W tile)
constexpr int SK_MAXC = 3;                // 64-wide K chunks per wave at most: K <= 64 * SK_MAXC * SK_WPB * gridDim.y

struct SkinnyArgs {
    const float *X; int ldx; int n;       // [n][K] activations, n <= 32 * MT
    const uint16_t *Xp; size_t xp_plane;  // XS: the activations pre-split by their producer into bf16 planes [3][n][K] (hi, mid, lo)
    uint16_t *Yp; size_t yp_plane;        // SK_SWIGLU: h is written as bf16 planes [3][n][N] for the w2 launch (and not as f32)
    const uint16_t *W, *W2;               // [N][K] bf16; W2 = up-projection rows (SK_SWIGLU)
    int N, K;
    const float *bias;                    // [N] or null (SK_QKV)
    float *Y; int ldy;                    // SK_QKV: merged qkv [n][N]; SK_SWIGLU: h [n][N]
    float *partial;                       // SK_PARTIAL: [gridDim.y][n][N]
    int rope_cols, head_dim;              // SK_QKV: columns [0, rope_cols) are rotated (q then k), pairs (2c, 2c+1)
    const float *rope_tab;                // [n][head_dim/2][2] cos, sin of this chunk's positions
    float *kring, *vring; int ring_cap, kv_dim, pos0, q_cols;   // k columns start at q_cols, v columns at q_cols + kv_dim
    unsigned long long *tl;               // optional (tuning, VOX_HIP_ENC_TL): per-workgroup timeline, see tl_begin / tl_end
};
#define SK_MARK(k) do { if (a.tl) sk_stamp[k] = wall_clock64(); } while (0)

// x[m][k..k+7] (f32) -> three bf16x8 fragments (hi, mid, lo): exact split, vox_gemm.h split3
__device__ __forceinline__ void sk_split_frag(const float4 x0, const float4 x1, bf16x8_t &fh, bf16x8_t &fm, bf16x8_t &fl) {
    const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) split3(xv[i], h[i], m[i], l[i]);
    union { uint32_t u[4]; bf16x8_t v; } ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ph.u[i] = (h[2 * i] >> 16) | h[2 * i + 1];
        pm.u[i] = (m[2 * i] >> 16) | m[2 * i + 1];
        pl.u[i] = (l[2 * i] >> 16) | (l[2 * i + 1] & 0xffff0000u);
    }
    fh = ph.v; fm = pm.v; fl = pl.v;
}

// bf16 planes of one f32 (hi, mid, lo as 16-bit patterns)
__device__ __forceinline__ void sk_planes1(float x, uint16_t &h, uint16_t &m, uint16_t &l) {
    uint32_t hh, mm, ll;
    split3(x, hh, mm, ll);
    h = (uint16_t)(hh >> 16); m = (uint16_t)(mm >> 16); l = (uint16_t)(ll >> 16);
}

// grid = (N / 32, S); block = 64 * SK_WPB.  MT = 1 (n <= 32) or 2 (n <= 64) row tiles of the activations.
// XS: the activation fragments come pre-split (three 16-byte loads per MFMA step, no VALU work): with f32 activations every
// one of the N / 32 workgroups repeats the same hi / mid / lo split of x, ~7 VALU operations per element and MFMA step, and
// that - not HBM - bounded the qkv and w1;w3 launches (measured 15 / 22 us for 15.7 / 26.2 MB).
//
// Round 3: v_mfma_f32_16x16x32_bf16 instead of the 32x32x16 shape.  Same MFMA time (twice the instructions at half the
// passes), same register counts - but the operand layout of the smaller shape makes a load instruction take 64 bytes from
// each of 16 rows instead of 32 bytes from each of 32, for the weights and for the activations alike, and that pattern
// streams at 5.3 TB/s where the other stops at 3.9 however many loads are in flight (tools/micro/frag_bw.hip).
template <int EPI, int MT, bool XS>
__global__ __launch_bounds__(64 * SK_WPB) void k_skinny(const SkinnyArgs a) {
    constexpr int NB = (EPI == SK_SWIGLU) ? 2 : 1;              // 32-row W tiles per wave
    constexpr int MU = 2 * MT;                                   // 16-row activation tiles
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];   // [SK_WPB][NB][MT][1024] accumulators
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kb = lane >> 4;
    const int row0 = blockIdx.x * 32;
    const int nchunks = a.K / 64;
    const int kworkers = gridDim.y * SK_WPB, kw = blockIdx.y * SK_WPB + wave;
    unsigned long long sk_stamp[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tl0 = tl_begin(a.tl);

    f32x4 acc[NB][2][MU];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int u = 0; u < MU; u++) acc[b][q][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint16_t *wrow[NB][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        wrow[0][q] = a.W + (size_t)(row0 + 16 * q + li) * a.K + kb * 8;
        if constexpr (NB == 2) wrow[1][q] = a.W2 + (size_t)(row0 + 16 * q + li) * a.K + kb * 8;
    }
    const float *xrow[MU];
    const uint16_t *xprow[MU];
#pragma unroll
    for (int u = 0; u < MU; u++) {
        xrow[u] = XS ? nullptr : a.X + (size_t)min(u * 16 + li, a.n - 1) * a.ldx + kb * 8;
        xprow[u] = XS ? a.Xp + (size_t)min(u * 16 + li, a.n - 1) * a.K + kb * 8 : nullptr;
    }

    // A wave owns at most SK_MAXC chunks (the host picks the K split accordingly).  All of its WEIGHT fragments are
    // requested up front (HBM latency paid once, 16 VGPRs per chunk and tile); the activation fragments (L2 hits, 32 VGPRs
    // per chunk) are double buffered.  Register arrays are only ever indexed by compile-time constants.
    // Fragment index f = 2 * (16-row tile) + (k step of 32): 4 per 32-row tile and 64-wide chunk, as with the 32x32x16 shape.
    uint4 wq[SK_MAXC][NB][4];
    constexpr int XR = XS ? 3 : 2;            // 16-byte registers per fragment: 3 bf16 planes, or 8 f32
    float4 xq0[MT][4][XR], xq1[MT][4][XR];
    auto issue_w = [&](uint4 (&wqc)[NB][4], int c) {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                // plain (L1-allocating) loads on purpose: the two k steps of a chunk share their 128-byte lines
                for (int b = 0; b < NB; b++) {
                    wqc[b][2 * q + ks] = *reinterpret_cast<const uint4 *>(wrow[b][q] + c * 64 + ks * 32);
                }
    };
    auto issue_x = [&](float4 (&xq)[MT][4][XR], int c) {
#pragma unroll
        for (int u = 0; u < MU; u++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int t = u >> 1, f = 2 * (u & 1) + ks;
                if constexpr (XS) {
#pragma unroll
                    for (int p = 0; p < 3; p++) xq[t][f][p] = *reinterpret_cast<const float4 *>(xprow[u] + p * a.xp_plane + c * 64 + ks * 32);
                } else {
                    xq[t][f][0] = *reinterpret_cast<const float4 *>(xrow[u] + c * 64 + ks * 32);
                    xq[t][f][1] = *reinterpret_cast<const float4 *>(xrow[u] + c * 64 + ks * 32 + 4);
                }
            }
    };
    auto compute = [&](const uint4 (&wqc)[NB][4], const float4 (&xq)[MT][4][XR]) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8_t fa[MU][3];
#pragma unroll
            for (int u = 0; u < MU; u++) {
                const int t = u >> 1, f = 2 * (u & 1) + ks;
                if constexpr (XS) {
#pragma unroll
                    for (int p = 0; p < 3; p++) { union { float4 f4; bf16x8_t v; } cv; cv.f4 = xq[t][f][p]; fa[u][p] = cv.v; }
                } else {
                    sk_split_frag(xq[t][f][0], xq[t][f][1], fa[u][0], fa[u][1], fa[u][2]);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    union { uint4 u4; bf16x8_t v; } fb;
                    fb.u4 = wqc[b][2 * q + ks];
#pragma unroll
                    for (int p = 2; p >= 0; p--)                   // small terms first
#pragma unroll
                        for (int u = 0; u < MU; u++)
                            acc[b][q][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u][p], fb.v, acc[b][q][u], 0, 0, 0);
                }
        }
    };
    static_assert(SK_MAXC == 3, "the chunk schedule below is written out for three chunks");
    const int c0 = kw, c1 = kw + kworkers, c2 = kw + 2 * kworkers;
    if (c0 < nchunks) { issue_x(xq0, c0); issue_w(wq[0], c0); }
    if (c1 < nchunks) issue_w(wq[1], c1);
    if (c2 < nchunks) issue_w(wq[2], c2);
    SK_MARK(0);
    if (c0 < nchunks) {
        if (c1 < nchunks) issue_x(xq1, c1);
        compute(wq[0], xq0);
        SK_MARK(1);
        if (c1 < nchunks) {
            if (c2 < nchunks) issue_x(xq0, c2);
            compute(wq[1], xq1);
            if (c2 < nchunks) compute(wq[2], xq0);
        }
    }
    SK_MARK(2);

    // ---- add the K splitters of this workgroup in wave order.  Tiles are kept as [32 rows m][32 columns] in LDS: the 16 x 16
    // MFMA's C layout is column = lane & 15, row = 4 (lane >> 4) + r ----------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int u = 0; u < MU; u++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    sk_lds[((wave * NB + b) * MT + (u >> 1)) * 1024 + (16 * (u & 1) + 4 * kb + r) * 32 + 16 * q + li] = acc[b][q][u][r];
    __syncthreads();
    SK_MARK(3);
    float *red = sk_lds;                                            // reduced tiles overwrite wave 0's slots
    for (int e = tid; e < NB * MT * 1024; e += 64 * SK_WPB) {
        float v = sk_lds[e];
#pragma unroll
        for (int w = 1; w < SK_WPB; w++) v += sk_lds[w * NB * MT * 1024 + e];
        red[e] = v;                                                 // same index as wave 0's own slot e: no hazard
    }
    __syncthreads();
    SK_MARK(4);

    // ---- epilogue over the tile: e -> (tile t, row, column) -> row m = 32 t + ((e >> 5) & 31), col = e & 31
    for (int e = tid; e < MT * 1024; e += 64 * SK_WPB) {
        const int t = e >> 10;
        const int m = t * 32 + ((e >> 5) & 31);
        const int col = row0 + (e & 31);
        if (m >= a.n) continue;
        if constexpr (EPI == SK_PARTIAL) {
            a.partial[((size_t)blockIdx.y * a.n + m) * a.N + col] = red[e];
        } else if constexpr (EPI == SK_SWIGLU) {
            const float hv = silu(red[e]) * red[MT * 1024 + e];                         // tile 0 = gate (w1), tile 1 = up (w3)
            if (a.Yp) {
                uint16_t ph, pm, pl;
                sk_planes1(hv, ph, pm, pl);
                uint16_t *dst = a.Yp + (size_t)m * a.N + col;
                dst[0] = ph; dst[a.yp_plane] = pm; dst[2 * a.yp_plane] = pl;
            } else {
                a.Y[(size_t)m * a.ldy + col] = hv;
            }
        } else {
            float v = red[e] + (a.bias ? a.bias[col] : 0.f);
            if (col < a.rope_cols) {
                const float o = red[e ^ 1] + (a.bias ? a.bias[col ^ 1] : 0.f);          // the pair partner: adjacent column, same row
                const int d = (col % a.head_dim) >> 1;
                const float cs = a.rope_tab[((size_t)m * (a.head_dim / 2) + d) * 2], sn = a.rope_tab[((size_t)m * (a.head_dim / 2) + d) * 2 + 1];
                v = (col & 1) ? o * sn + v * cs : v * cs - o * sn;                       // (x0 c - x1 s, x0 s + x1 c)
            }
            a.Y[(size_t)m * a.ldy + col] = v;
            if (col >= a.q_cols) {                                                       // k / v also go to the ring slot of their position
                const int slot = (a.pos0 + m) % a.ring_cap;
                if (col < a.q_cols + a.kv_dim) a.kring[(size_t)slot * a.kv_dim + (col - a.q_cols)] = v;
                else a.vring[(size_t)slot * a.kv_dim + (col - a.q_cols - a.kv_dim)] = v;
            }
        }
    }
    tl_end(a.tl, tl0, sk_stamp, 5);
}

// x[m] += bias + sum_s partial[s][m]  (split order), then out_norm[m] = rmsnorm(x[m]) * w (+ada) — one block per row.
// Covers "x += wo(attn) + bo -> ffn_norm" and "x += w2(h) + b2 -> next layer's attention_norm / the final norm".
// planes (optional): the normalised row also as bf16 planes [3][n][D] for the XS skinny launches that consume it.
// block = one row, blockDim = a multiple of 64 up to 1024 (the launchers use D / 4 threads: one float4 column group per
// thread, so that all of a row's partial loads are in flight together - with 256 threads a 3072-wide row of the decoder took
// three passes of up to three load batches each: 8.7 us for 24 splits).
__global__ __launch_bounds__(1024) void k_rows_finish(float *x, int ldx, const float *partial, int nsplit, int n, int D,
                                                      const float *bias, const float *norm_w, float eps, float *out_norm, int ldo,
                                                      uint16_t *planes, const float *ada = nullptr) {
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    float *xr = x + (size_t)m * ldx;
    float ss = 0.f;
    for (int i = tid * 4; i < D; i += nth * 4) {
        float4 v = *reinterpret_cast<const float4 *>(xr + i);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // The partials were written by other CUs' previous kernel: every load is a trip to L2 / memory (~1 us).  Request a
        // batch of 16 before adding any of them (clamped index + select instead of a branch, so that the loads do not end up
        // behind a wait one by one: 10 splits took 8.8 us that way); the additions stay in split order.
        for (int z0 = 0; z0 < nsplit; z0 += 16) {
            float4 p[16];
#pragma unroll
            for (int u = 0; u < 16; u++) p[u] = *reinterpret_cast<const float4 *>(partial + ((size_t)min(z0 + u, nsplit - 1) * n + m) * D + i);
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const bool on = z0 + u < nsplit;
                s.x += on ? p[u].x : 0.f; s.y += on ? p[u].y : 0.f; s.z += on ? p[u].z : 0.f; s.w += on ? p[u].w : 0.f;
            }
        }
        if (bias) { const float4 b = *reinterpret_cast<const float4 *>(bias + i); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;                 // x + (proj + bias): the reference's order (vox_add_inplace)
        *reinterpret_cast<float4 *>(xr + i) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (!norm_w) return;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (nth >> 6); w++) tot += red[w];
    const float inv = 1.0f / sqrtf(tot / (float)D + eps);
    float *orow = out_norm ? out_norm + (size_t)m * ldo : nullptr;
    for (int i = tid * 4; i < D; i += nth * 4) {
        const float4 v = *reinterpret_cast<const float4 *>(xr + i);      // own elements, written above by this thread
        const float4 g = *reinterpret_cast<const float4 *>(norm_w + i);
        float4 o = make_float4(v.x * inv * g.x, v.y * inv * g.y, v.z * inv * g.z, v.w * inv * g.w);
        if (ada) {                                                       // decoder ffn_norm: x_norm *= 1 + ada_scale (voxtral_decoder.c:678-682)
            const float4 sc = *reinterpret_cast<const float4 *>(ada + i);
            o.x *= (1.0f + sc.x); o.y *= (1.0f + sc.y); o.z *= (1.0f + sc.z); o.w *= (1.0f + sc.w);
        }
        if (orow) *reinterpret_cast<float4 *>(orow + i) = o;
        if (planes) {
            uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            split3(o.x, h0, m0, l0); split3(o.y, h1, m1, l1); split3(o.z, h2, m2, l2); split3(o.w, h3, m3, l3);
            uint16_t *dst = planes + (size_t)m * D + i;
            const size_t ps = (size_t)n * D;
            *reinterpret_cast<uint2 *>(dst) = make_uint2((h0 >> 16) | h1, (h2 >> 16) | h3);
            *reinterpret_cast<uint2 *>(dst + ps) = make_uint2((m0 >> 16) | m1, (m2 >> 16) | m3);
            *reinterpret_cast<uint2 *>(dst + 2 * ps) = make_uint2((l0 >> 16) | (l1 & 0xffff0000u), (l2 >> 16) | (l3 & 0xffff0000u));
        }
    }
}

}  // namespace vox
