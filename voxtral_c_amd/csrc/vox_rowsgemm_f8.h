// vox_rowsgemm_f8.h — y[n, N] = x[n, K] . W8[N, K]^T on the fp8 matrix pipe (v_mfma_f32_16x16x32_fp8_fp8), n <= 64 rows.
//
// BASELINE config 5 ("fp8 weights (CDNA4 fp8 MFMA)"): in fp8 mode the decoder's seven matrices per layer exist as row-scaled e4m3
// copies (k_quant_fp8_rows, vox_gemv.h) which the decode GEMVs stream.  This is the M > 1 side of the same mode: the decoder prefill
// (38 rows, reference vox_decoder_prefill, voxtral_decoder.c:410-558) reads those copies too - half the bytes of the bf16 pass - and
// multiplies on the fp8 MFMA.  Same work split as k_rowsgemm (vox_rowsgemm.h): the waves of a workgroup own DIFFERENT 16-row weight
// tiles and the SAME K range, K is split over blockIdx.y, the raw partial sums [split][n][N] are added in split order by the finish
// kernels of vox_rowsgemm.h (k_qkv_finish, k_rows_finish, k_swiglu_finish).
//
// Operands.  The fp8 MFMA takes fp8 on both sides, so the f32 activations are split into TWO e4m3 terms on their way into LDS:
//     hi = e4m3(x * ps),   lo = e4m3((x * ps - hi) * 16)        x * ps ~= hi + lo / 16
// (8 significant bits, i.e. bf16-like activations; ps = a power of two that centres the activations in e4m3's range) and every product is two
// MFMAs into two accumulators, combined in the epilogue with the weight row's scale:  y = (acc_hi + acc_lo / 16) * s_w / ps.
// The weights' quantisation error (4 significant bits) dominates the result's error by 16x; what this kernel adds to the error of
// the fp8 decode GEMVs (which keep f32 activations) is measured in tests/test_gpu_kernels_api.py.
//
// Fragments.  A 64-column chunk of a weight row is 64 bytes: one 16-byte load per lane (lane = (row li = lane & 15, quarter kb =
// lane >> 4) -> bytes [16 kb, 16 kb + 16) of the chunk) - 64 bytes from each of 16 rows per instruction, the pattern that streams
// at 5.3 TB/s (vox_rowsgemm.h).  Its low and high 8 bytes are the B operands of two MFMA k-steps; which 32 of the chunk's 64
// columns a k-step contracts over is free as long as both operands agree: step s takes columns {16 kb + 8 s + i}, and the
// activation fragment of lane (li, kb) is the 8 bytes at [16 kb + 8 s, + 8) of row li of the chunk in LDS.
#pragma once
#include "vox_common.h"

namespace vox {

struct RowsGemmF8Args {
    const float *X; int ldx; int n;          // activations [n][K] f32
    const uint8_t *W; const float *wscale;   // [N][K] e4m3 bytes, one f32 scale per row
    int N, K;
    int cw;                                  // 64-wide K chunks per workgroup (blockIdx.y owns chunks [y cw, (y + 1) cw)), walked CPW at a time
    float prescale;                          // power of two applied to x before the split
    unsigned *clamped;                       // counts activations beyond the e4m3 range after the prescale (|x ps| > 224): the host then repeats the pass in bf16
    float *partial;                          // [gridDim.y][n][N]
};

typedef float f32x4_f8 __attribute__((ext_vector_type(4)));

// grid = (ceil(N / (32 WPB)), ceil(K / 64 / cw)); block = 64 WPB; static LDS: 2 stages x 2 planes x 64 rows x CPW x 64 B.
// MT = 16-row activation tiles (n <= 16 MT).  A wave owns 2 weight tiles (32 rows of W).
template <int WPB, int CPW, int MT>
__global__ __launch_bounds__(64 * WPB) void k_rowsgemm_f8(const RowsGemmF8Args a) {
    constexpr int NT = 64 * WPB, ROWS = 16 * MT, RB = 64 * CPW;            // bytes per LDS row and plane
    constexpr int PLANE = ROWS * RB, STAGE = 2 * PLANE;
    constexpr int SLOTS = RB / 16;                                         // 16-byte slots per row (XOR-swizzled by the row)
    constexpr int ITEMS = (ROWS * RB / 8 + NT - 1) / NT;                   // 8-column groups per thread and stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kb = lane >> 4;
    const int nchunks = a.K / 64;
    const int c_begin = blockIdx.y * a.cw, c_end = min(nchunks, c_begin + a.cw);
    const int row0 = (blockIdx.x * WPB + wave) * 32;
    const uint8_t *wrow[2];
#pragma unroll
    for (int q = 0; q < 2; q++) wrow[q] = a.W + (size_t)min(row0 + q * 16 + li, a.N - 1) * a.K + kb * 16;

    auto load_w = [&](uint4 (&wr)[2][CPW], int cb) {
#pragma unroll
        for (int c = 0; c < CPW; c++)
#pragma unroll
            for (int q = 0; q < 2; q++) wr[q][c] = *reinterpret_cast<const uint4 *>(wrow[q] + (size_t)min(cb + c, nchunks - 1) * 64);
    };
    // f32 rows of a stage into registers (requested before the MFMAs of the previous round) ...
    auto x_load = [&](float4 (&xa)[ITEMS], float4 (&xb)[ITEMS], int cb, int nc) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const int idx = min(it * NT + tid, ROWS * RB / 8 - 1);
            const int r = idx / (RB / 8), g8 = idx - r * (RB / 8);
            const int kk = min(g8 * 8, nc * 64 - 8);
            const float *src = a.X + (size_t)min(r, a.n - 1) * a.ldx + cb * 64 + kk;
            xa[it] = *reinterpret_cast<const float4 *>(src);
            xb[it] = *reinterpret_cast<const float4 *>(src + 4);
        }
    };
    // ... and the two-term e4m3 split into LDS stage st: 8 bytes per plane and item, slot (g8 / 2) ^ (r & (SLOTS - 1)) of row r
    auto x_store = [&](int st, const float4 (&xa)[ITEMS], const float4 (&xb)[ITEMS]) {
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const int idx = it * NT + tid;
            if (idx < ROWS * RB / 8) {
                const int r = idx / (RB / 8), g8 = idx - r * (RB / 8);
                const float ps = a.prescale;
                // (clamped to the range both e4m3 flavours represent: a value beyond 224 / ps loses its excess over hi + lo / 16 = 238 / ps)
                auto cl = [](float v) { return fminf(fmaxf(v, -224.0f), 224.0f); };
                const float xv[8] = {cl(xa[it].x * ps), cl(xa[it].y * ps), cl(xa[it].z * ps), cl(xa[it].w * ps), cl(xb[it].x * ps), cl(xb[it].y * ps), cl(xb[it].z * ps), cl(xb[it].w * ps)};
                // Round 6: a clamped activation is not silent any more.  The fixed prescale holds for activations up to 112 in magnitude;
                // an outlier beyond that (gated hidden rows, Wo inputs of a real checkpoint) would corrupt the prefill's K/V without a trace:
                // it is counted (workgroups of the first weight-row block only: every block sees the same rows) and the host falls back.
                if (a.clamped && blockIdx.x == 0 && r < a.n) {
                    const float lim = 224.0f / ps;
                    const bool over = fabsf(xa[it].x) > lim || fabsf(xa[it].y) > lim || fabsf(xa[it].z) > lim || fabsf(xa[it].w) > lim ||
                                      fabsf(xb[it].x) > lim || fabsf(xb[it].y) > lim || fabsf(xb[it].z) > lim || fabsf(xb[it].w) > lim;
                    if (over) atomicAdd(a.clamped, 1u);
                }
                // (the builtins want literal word selectors: written out pair by pair)
                int hw0 = 0, hw1 = 0, lw0 = 0, lw1 = 0;
                hw0 = __builtin_amdgcn_cvt_pk_fp8_f32(xv[0], xv[1], hw0, false); hw0 = __builtin_amdgcn_cvt_pk_fp8_f32(xv[2], xv[3], hw0, true);
                hw1 = __builtin_amdgcn_cvt_pk_fp8_f32(xv[4], xv[5], hw1, false); hw1 = __builtin_amdgcn_cvt_pk_fp8_f32(xv[6], xv[7], hw1, true);
                const f32x2 b0 = __builtin_amdgcn_cvt_pk_f32_fp8(hw0, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8(hw0, true);
                const f32x2 b2 = __builtin_amdgcn_cvt_pk_f32_fp8(hw1, false), b3 = __builtin_amdgcn_cvt_pk_f32_fp8(hw1, true);
                lw0 = __builtin_amdgcn_cvt_pk_fp8_f32((xv[0] - b0.x) * 16.0f, (xv[1] - b0.y) * 16.0f, lw0, false);
                lw0 = __builtin_amdgcn_cvt_pk_fp8_f32((xv[2] - b1.x) * 16.0f, (xv[3] - b1.y) * 16.0f, lw0, true);
                lw1 = __builtin_amdgcn_cvt_pk_fp8_f32((xv[4] - b2.x) * 16.0f, (xv[5] - b2.y) * 16.0f, lw1, false);
                lw1 = __builtin_amdgcn_cvt_pk_fp8_f32((xv[6] - b3.x) * 16.0f, (xv[7] - b3.y) * 16.0f, lw1, true);
                const int slot = (g8 >> 1) ^ (r & (SLOTS - 1));
                unsigned char *dst = lds + st * STAGE + r * RB + slot * 16 + (g8 & 1) * 8;
                *reinterpret_cast<uint2 *>(dst) = make_uint2((unsigned)hw0, (unsigned)hw1);
                *reinterpret_cast<uint2 *>(dst + PLANE) = make_uint2((unsigned)lw0, (unsigned)lw1);
            }
        }
    };

    f32x4_f8 acc_h[2][MT], acc_l[2][MT];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int t = 0; t < MT; t++) { acc_h[q][t] = f32x4_f8{0.f, 0.f, 0.f, 0.f}; acc_l[q][t] = f32x4_f8{0.f, 0.f, 0.f, 0.f}; }

    auto compute = [&](const uint4 (&wr)[2][CPW], int st, int nc) {
        const unsigned char *stage = lds + st * STAGE;
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            if (c < nc) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        const int r = t * 16 + li;
                        const unsigned char *src = stage + r * RB + (((c * 4 + kb) ^ (r & (SLOTS - 1))) * 16) + s * 8;
                        const long fa_h = *reinterpret_cast<const long *>(src);
                        const long fa_l = *reinterpret_cast<const long *>(src + PLANE);
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const long fb = s == 0 ? (long)(((unsigned long)wr[q][c].y << 32) | (unsigned long)wr[q][c].x)
                                                   : (long)(((unsigned long)wr[q][c].w << 32) | (unsigned long)wr[q][c].z);
                            acc_l[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fa_l, fb, acc_l[q][t], 0, 0, 0);
                            acc_h[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fa_h, fb, acc_h[q][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    uint4 wA[2][CPW], wB[2][CPW];
    float4 xa[ITEMS], xb[ITEMS];
    load_w(wA, c_begin);
    x_load(xa, xb, c_begin, min(CPW, c_end - c_begin));
    x_store(0, xa, xb);
    int st = 0, cb = c_begin;
    auto round = [&](const uint4 (&wcur)[2][CPW], uint4 (&wnext)[2][CPW]) {
        const int nc = min(CPW, c_end - cb);
        const bool more = cb + CPW < c_end;
        __syncthreads();                                         // this round's stage is written; everybody is past the MFMAs that read the other one
        if (more) {
            load_w(wnext, cb + CPW);
            x_load(xa, xb, cb + CPW, min(CPW, c_end - cb - CPW));
        }
        compute(wcur, st, nc);
        if (more) x_store(st ^ 1, xa, xb);
        cb += CPW; st ^= 1;
    };
    while (cb < c_end) {
        round(wA, wB);
        if (cb >= c_end) break;
        round(wB, wA);
    }
    // ---- raw partial sums, C layout of the 16 x 16 MFMA: column = lane & 15, row = 4 (lane >> 4) + r ----
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const float inv_ps = 1.0f / a.prescale;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int col = row0 + q * 16 + li;
        if (col < a.N) {
            const float sc = a.wscale[col] * inv_ps;
            float *P0 = a.partial + (size_t)blockIdx.y * a.n * a.N + col;
#pragma unroll
            for (int t = 0; t < MT; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = t * 16 + 4 * kb + r;
                    if (m < a.n) P0[(size_t)m * a.N] = (acc_h[q][t][r] + acc_l[q][t][r] * 0.0625f) * sc;
                }
            }
        }
    }
}

// Reference of the test surface (vox_hip_linear_bf16, impl 7): y[m][j] = sum_k x[m][k] * f32(W8[j][k]) * scale[j] (+ bias), f32 FMAs in
// k order, one wave per output column.  Not a production kernel.
__global__ __launch_bounds__(64) void k_fp8_ref_gemm(float *y, const float *x, const uint8_t *W8, const float *scale, const float *bias, int M, int N, int K) {
    const int j = blockIdx.x, lane = threadIdx.x;
    for (int m = 0; m < M; m++) {
        float acc = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const int word = *reinterpret_cast<const int *>(W8 + (size_t)j * K + k);
            const f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8(word, false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8(word, true);
            const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)m * K + k);
            acc = fmaf(p0.x, xv.x, acc); acc = fmaf(p0.y, xv.y, acc); acc = fmaf(p1.x, xv.z, acc); acc = fmaf(p1.y, xv.w, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) y[(size_t)m * N + j] = acc * scale[j] + (bias ? bias[j] : 0.f);
    }
}

}  // namespace vox
