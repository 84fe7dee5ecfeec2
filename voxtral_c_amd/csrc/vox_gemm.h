// vox_gemm.h — large-M  y[M,N] = x[M,K](f32) . W[N,K]^T(bf16)  for gfx950.
//
// Replaces the reference's "convert the whole bf16 matrix to f32, then cblas_sgemm"
// path (voxtral_kernels.c:197-240, 88-116) used by the encoder chunk, decoder prefill,
// adapter and (via im2col) the conv stem (voxtral_kernels.c:293-340).
//
// Precision contract: the oracle multiplies exact-bf16 weights (upcast to f32) by f32
// activations with f32 accumulation.  k_gemm_mfma_f32 does exactly that on the matrix
// cores with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf chain,
// 157 TFLOP/s peak) — bf16 weights are upcast while being staged into LDS, activations
// are never rounded.  Roofline: MFMA(f32) for M >~ 200 rows, HBM/L2 (weight streaming)
// for the small streaming chunks.
//
// Tiling (64-wide waves): 128x128 output tile per 256-thread block, 4 waves as 2x2,
// each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  K is consumed in
// 32-wide slices staged through LDS with a +4 float row pad: every ds_read_b128 of a
// 16-lane service group lands on 16 distinct 16-byte slots (conflict-free, see the
// bank model in MI355X_MICROARCH.md §LDS).  A lane reads 4 consecutive k of its row with
// one ds_read_b128 and feeds them to 4 MFMA k-steps; A and B use the same k<->slot map,
// which is all the dot product needs.  Global loads of slice t+1 are issued before the
// MFMAs of slice t (register-staged software pipeline).
//
// Fused epilogue: + bias[n], activation (none | tanh-GELU | SiLU), + residual[m,n].
#pragma once
#include "vox_common.h"

namespace vox {

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };

constexpr int GB_M = 128, GB_N = 128, GB_K = 32, GB_LD = GB_K + 4;
constexpr int GEMM_LDS_BYTES = (GB_M + GB_N) * GB_LD * 4;

struct GemmArgs {
    const float *X; int ldx;       // [M, K]
    const uint16_t *W;             // [N, K]
    float *Y; int ldy;             // [M, N]
    int M, N, K;
    const float *bias;             // [N] or null
    const float *resid; int ldr;   // [M, N] or null (may alias Y)
    int act;
    // split-K (small grids): blockIdx.z owns K-slices [z*kper, (z+1)*kper); raw partial sums go
    // to partial[z][M][N] and k_splitk_reduce applies the epilogue in a fixed order
    // (deterministic — no atomics).
    int ksplit, kper;
    float *partial;
    // k_gemm_planes (vox_gemm_planes.h): the activations pre-split into bf16 planes [3][M][K] (hi, mid, lo), row stride ldxp
    const uint16_t *Xp; size_t xp_plane; int ldxp;
    // k_gemm_planes epilogues: GP_EPI_SWIGLU writes silu(gate) * up as bf16 planes [3][M][N] (W = [w1; w3], w3 at row N);
    // GP_EPI_ROPE applies the interleaved-pair RoPE (table [M][head_dim / 2][cos, sin]) to the first rope_cols columns
    uint16_t *Yp; size_t yp_plane;
    const float *rope_tab; int rope_cols, head_dim;
    // k_gemm_planes, split-K launches: 1-D grid in XCD-aware order (xcd_tn = N tiles, xcd_tm = M tiles; 0 = plain 3-D grid)
    int xcd_tn, xcd_tm;
    // GP_EPI_ROPE, optional (round 6): rows >= ring_row0 of the k / v columns (column >= ring_col0: k, >= ring_col0 + ring_kvd: v) ALSO go to
    // their slots of the position-indexed K / V rings - row r to slot (ring_pos0 + r - ring_row0) % ring_cap - instead of a k_ring_append pass
    float *kring, *vring; int ring_cap, ring_kvd, ring_col0, ring_row0, ring_pos0;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_GELU) return gelu_tanh(v);
    if (act == ACT_SILU) return silu(v);
    return v;
}

// Fused epilogue shared by the MFMA kernels (C/D layout of every 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <int TN = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x16 (&acc)[2][TN], int bm0, int bn0, int wm, int wn,
                                              int li, int lg, int zslice = -1) {
    const int M = a.M, N = a.N;
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    if (a.ksplit > 1) {
        float *P = a.partial + (size_t)(zslice >= 0 ? zslice : (int)blockIdx.z) * M * N;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int col = bn0 + wn * (32 * TN) + tn * 32 + li;
                if (col >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = bm0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                    if (row < M) P[(size_t)row * N + col] = acc[tm][tn][r];
                }
            }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const int col = bn0 + wn * (32 * TN) + tn * 32 + li;
            if (col >= N) continue;
            const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = bm0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                if (row < M) {
                    float v = acc[tm][tn][r];
                    if (a.bias) v += b;
                    v = apply_act(v, a.act);
                    if (a.resid) v = a.resid[(size_t)row * a.ldr + col] + v;
                    a.Y[(size_t)row * a.ldy + col] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm_mfma_f32(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [GB_M][GB_LD]
    float *Bs = smem + GB_M * GB_LD;      // [GB_N][GB_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm0 = blockIdx.y * GB_M, bn0 = blockIdx.x * GB_N;
    const int M = a.M, N = a.N, K = a.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // staging assignment
    //   A slice: 128 rows x 32 f32 = 1024 float4 -> 4 per thread (row = idx>>3, c4 = idx&7)
    //   B slice: 128 rows x 32 bf16 = 512 uint4  -> 2 per thread (row = idx>>2, c8 = idx&3)
    float4 ra[4];
    uint4 rb[2];
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            const int gm = bm0 + row;
            ra[i] = (gm < M) ? *reinterpret_cast<const float4 *>(a.X + (size_t)gm * a.ldx + k0 + c4 * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, row = idx >> 2, c8 = idx & 3;
            const int gn = bn0 + row;
            rb[i] = (gn < N) ? *reinterpret_cast<const uint4 *>(a.W + (size_t)gn * K + k0 + c8 * 8)
                             : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4 *>(As + row * GB_LD + c4 * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, row = idx >> 2, c8 = idx & 3;
            float4 lo, hi;
            lo.x = bf16_lo(rb[i].x); lo.y = bf16_hi(rb[i].x); lo.z = bf16_lo(rb[i].y); lo.w = bf16_hi(rb[i].y);
            hi.x = bf16_lo(rb[i].z); hi.y = bf16_hi(rb[i].z); hi.z = bf16_lo(rb[i].w); hi.w = bf16_hi(rb[i].w);
            *reinterpret_cast<float4 *>(Bs + row * GB_LD + c8 * 8) = lo;
            *reinterpret_cast<float4 *>(Bs + row * GB_LD + c8 * 8 + 4) = hi;
        }
    };

    const int nk_total = K / GB_K;
    const int kt0 = (a.ksplit > 1) ? blockIdx.z * a.kper : 0;
    const int kt1 = (a.ksplit > 1) ? min(nk_total, kt0 + a.kper) : nk_total;
    const int nk = kt1 - kt0;
    const int li = lane & 31, lg = lane >> 5;
    if (nk > 0) {
        load_slice(kt0 * GB_K);
        store_slice();
    }
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_slice((kt0 + kt + 1) * GB_K);
#pragma unroll
        for (int kk = 0; kk < GB_K; kk += 8) {
            float4 av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                av[t] = *reinterpret_cast<const float4 *>(As + (wm * 64 + t * 32 + li) * GB_LD + kk + lg * 4);
                bv[t] = *reinterpret_cast<const float4 *>(Bs + (wn * 64 + t * 32 + li) * GB_LD + kk + lg * 4);
            }
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_slice();
            __syncthreads();
        }
    }

    gemm_epilogue(a, acc, bm0, bn0, wm, wn, li, lg);
}

// ---------------------------------------------------------------------------------------
// k_gemm_mfma_bf16x3 — the same product on the bf16 matrix pipe, still exact in the inputs.
//
// W is bf16 already.  An f32 activation x splits EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid
// (24 significand bits = 3 x 8; every remainder is exactly representable), so
//     sum_k W[n,k] x[m,k] = sum_k W hi + sum_k W mid + sum_k W lo
// is three bf16 x bf16 MFMA passes whose products are exact and whose accumulation is f32 —
// the oracle's arithmetic up to summation order, at 3/16 of the f32-MFMA instruction cost
// (v_mfma_f32_32x32x16_bf16: 16 k per instruction at 16x the f32 rate).
//
// Tiling: 128x128 tile / 256 threads / 2x2 waves of 64x64 (2x2 MFMA tiles), K slices of 64.
// LDS per slice: three A planes + one B plane, rows of 64 bf16 padded to 72 (144 B: the 16
// lanes of a ds_read_b128 service group hit 16 distinct 16-byte slots).  A lane's fragment is
// 8 consecutive k of its row (one ds_read_b128); A and B use the same k <-> (lane>>5, element)
// map, which is all the dot product needs.  Global loads of slice t+1 are issued before the
// MFMAs of slice t; the f32 -> 3 x bf16 split happens on the way into LDS.
// ---------------------------------------------------------------------------------------
constexpr int GX_K = 64, GX_LD = GX_K + 8;                    // bf16 elements per LDS row
constexpr int GEMM_X3_LDS_BYTES = 4 * GB_M * GX_LD * 2;       // 3 A planes + B

typedef short bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
    const uint32_t xb = __float_as_uint(x);
    h = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    const uint32_t rb = __float_as_uint(r1);
    m = rb & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);                      // <= 8 significant bits: exact in the top half
}

__global__ __launch_bounds__(256) void k_gemm_mfma_bf16x3(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t *Ap = reinterpret_cast<uint16_t *>(smem);            // [3][GB_M][GX_LD]
    uint16_t *Bs = Ap + 3 * GB_M * GX_LD;                         // [GB_N][GX_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm0 = blockIdx.y * GB_M, bn0 = blockIdx.x * GB_N;
    const int M = a.M, N = a.N, K = a.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // staging: A slice 128 x 64 f32 = 2048 float4 -> 8 per thread (row = idx>>4, c4 = idx&15)
    //          B slice 128 x 64 bf16 = 1024 uint4 -> 4 per thread (row = idx>>3, c8 = idx&7)
    float4 ra[8];
    uint4 rb[4];
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            const int gm = bm0 + row;
            ra[i] = (gm < M) ? *reinterpret_cast<const float4 *>(a.X + (size_t)gm * a.ldx + k0 + c4 * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c8 = idx & 7;
            const int gn = bn0 + row;
            rb[i] = (gn < N) ? *reinterpret_cast<const uint4 *>(a.W + (size_t)gn * K + k0 + c8 * 8)
                             : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            split3(ra[i].x, h0, m0, l0); split3(ra[i].y, h1, m1, l1);
            split3(ra[i].z, h2, m2, l2); split3(ra[i].w, h3, m3, l3);
            // element k at the lower address: pack (k, k+1) as (hi16 of k) | (hi16 of k+1) << 16
            uint2 ph, pm, pl;
            ph.x = (h0 >> 16) | h1; ph.y = (h2 >> 16) | h3;
            pm.x = (m0 >> 16) | m1; pm.y = (m2 >> 16) | m3;
            pl.x = (l0 >> 16) | (l1 & 0xffff0000u); pl.y = (l2 >> 16) | (l3 & 0xffff0000u);
            uint16_t *dst = Ap + row * GX_LD + c4 * 4;
            *reinterpret_cast<uint2 *>(dst) = ph;
            *reinterpret_cast<uint2 *>(dst + GB_M * GX_LD) = pm;
            *reinterpret_cast<uint2 *>(dst + 2 * GB_M * GX_LD) = pl;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256, row = idx >> 3, c8 = idx & 7;
            *reinterpret_cast<uint4 *>(Bs + row * GX_LD + c8 * 8) = rb[i];
        }
    };

    const int nk_total = K / GX_K;
    const int kt0 = (a.ksplit > 1) ? blockIdx.z * a.kper : 0;
    const int kt1 = (a.ksplit > 1) ? min(nk_total, kt0 + a.kper) : nk_total;
    const int nk = kt1 - kt0;
    const int li = lane & 31, lg = lane >> 5;
    if (nk > 0) {
        load_slice(kt0 * GX_K);
        store_slice();
    }
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_slice((kt0 + kt + 1) * GX_K);
#pragma unroll
        for (int kk = 0; kk < GX_K; kk += 16) {
            bf16x8_t af[2][3], bf[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint16_t *ap = Ap + (wm * 64 + t * 32 + li) * GX_LD + kk + lg * 8;
#pragma unroll
                for (int p = 0; p < 3; p++) af[t][p] = *reinterpret_cast<const bf16x8_t *>(ap + p * GB_M * GX_LD);
                bf[t] = *reinterpret_cast<const bf16x8_t *>(Bs + (wn * 64 + t * 32 + li) * GX_LD + kk + lg * 8);
            }
#pragma unroll
            for (int p = 2; p >= 0; p--)                       // small terms first
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][p], bf[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_slice();
            __syncthreads();
        }
    }
    gemm_epilogue(a, acc, bm0, bn0, wm, wn, li, lg);
}

// Sum the split-K partials in split order and apply the fused epilogue.
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmArgs a) {
    const size_t total = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / a.N), n = (int)(i % a.N);
        float v = 0.f;
        for (int z = 0; z < a.ksplit; z++) v += a.partial[(size_t)z * total + i];
        if (a.bias) v += a.bias[n];
        v = apply_act(v, a.act);
        if (a.resid) v = a.resid[(size_t)m * a.ldr + n] + v;
        a.Y[(size_t)m * a.ldy + n] = v;
    }
}

// Plain fp32 FMA reference kernel (one thread per output, sequential k like the
// reference's non-BLAS loop, voxtral_kernels.c:103-114).  Used for the start-up MFMA
// layout self-test, for odd shapes (K % 32 != 0) and as a cross-check in the tests.
__global__ __launch_bounds__(256) void k_gemm_scalar(const GemmArgs a) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= a.M || n >= a.N) return;
    const float *x = a.X + (size_t)m * a.ldx;
    const uint16_t *w = a.W + (size_t)n * a.K;
    float acc = 0.f;
    for (int k = 0; k < a.K; k++) acc = fmaf(x[k], bf16_to_f32(w[k]), acc);
    if (a.bias) acc += a.bias[n];
    acc = apply_act(acc, a.act);
    if (a.resid) acc = a.resid[(size_t)m * a.ldr + n] + acc;
    a.Y[(size_t)m * a.ldy + n] = acc;
}

}  // namespace vox
