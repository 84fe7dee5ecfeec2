// vox_rowsgemm.h — y[n, N] = x[n, K] . W[N, K]^T for 33 .. 128 rows (and fewer): the decoder prefill (38 rows, reference
// vox_decoder_prefill, voxtral_decoder.c:410-558) and the encoder's flush pass (68 rows at the end of a batch transcription,
// vox_encoder_forward_incremental, voxtral_encoder.c:452-636).  HBM-bound on the bf16 weights: every weight byte is wanted
// once, at streaming rate, by >= 200 workgroups.
//
// What the two existing paths do with such a chunk (profiles/r02_kernel_stats.csv, round-3 trace of the 30 s clip):
//   * the 128 x 128 MFMA tiles of vox_gemm.h / vox_gemm_planes.h move a full 128-row A tile per K slice through LDS for
//     38 - 68 real rows, need split-K + a reduce launch to fill the chip, and separate RoPE / ring-append / SiLU / RMSNorm
//     launches: 14 - 16 launches and 150 - 220 us per layer (prefill 5.7 ms, flush 4.8 ms of the 30 s pass);
//   * k_skinny (vox_skinny.h) lets the 8 waves of a workgroup split K of ONE 32-row weight tile and reads the activation
//     fragments straight from L2: every one of the N / 32 workgroups re-reads all of x (3 planes x n x K x 2 B: 2.3 x the
//     weight bytes at n = 25, 6 x at n = 64), and its register budget (all weight fragments up front + double-buffered
//     activation fragments) stops at one 32-row activation tile.  Measured with the loads removed one by one
//     (a debug build with the loads switched off one by one): the x loads cost as much as the weight loads (4 - 5 us of a 12 - 16 us launch).
//
// Here the roles are turned around: the waves of a workgroup own DIFFERENT weight tiles (8 x 32 = 256 rows of W, or 4 x 32)
// and the SAME K range (CPW chunks of 64), K is split over blockIdx.y.  The activation K range (3 bf16 planes x n rows x
// 64 CPW) is brought into LDS once per workgroup - by LDS-DMA from the producer's planes, or split on the fly from f32 rows
// - and every wave reads its MFMA A fragments from there; a wave's weight fragments (B operand, 16 bytes per lane in
// fragment layout straight from global memory, CPW x 4 registers) are all requested up front.  x traffic drops by the
// number of waves per workgroup (8 x), no wave waits for another except at the one barrier, the accumulators go straight
// from registers to the partial-sum buffer [split][n][N] (no LDS reduction), and up to eight 16-row activation tiles cost
// 4 accumulator registers each per weight tile.  The K splits are added in split order by whoever consumes them (k_rows_finish for the
// residual + next RMSNorm, k_qkv_finish for bias + RoPE + KV append, k_swiglu_finish for the gate): deterministic, no atomics.
// Arithmetic: exact bf16 weights x the exact 3-term bf16 split of the f32 activations on v_mfma_f32_16x16x32_bf16, f32
// accumulation = the reference's cblas_sgemm / bf16_matvec up to summation order (vox_gemm.h).
#pragma once
#include "vox_common.h"
#include "vox_gemm.h"
#include <type_traits>

namespace vox {

enum { RG_X_PLANES = 0, RG_X_F32 = 1 };

struct RowsGemmArgs {
    const uint16_t *Xp; size_t xp_plane;      // RG_X_PLANES: activations as bf16 planes [3][n][K] (hi, mid, lo), row stride K
    const float *X; int ldx;                  // RG_X_F32: activations [n][K] f32, split into planes on the way into LDS
    int n, mt;                                // rows, 32-row tiles (mt = ceil(n / 32) <= 4)
    const uint16_t *W; int N, K;              // [N][K] bf16
    int cw;                                   // 64-wide K chunks per workgroup (blockIdx.y owns chunks [y cw, (y + 1) cw)), walked CPW at a time
    float *partial;                           // [gridDim.y][n][N] raw partial sums
};

// 16-byte slot swizzle of an LDS row of P slots: 16 consecutive rows reading the same logical slot hit 16 distinct slots of the
// 256-byte bank window (rows are P x 16 bytes apart: for P % 16 == 0 the row offset vanishes mod 256 and the low 4 bits of the
// row index are XORed in; for P = 8 the offset alternates 0 / 128 and 3 bits of (row >> 1) do it).
template <int P>
__device__ __forceinline__ int rg_sw(int r) { return (P % 16 == 0) ? (r & 15) : ((r >> 1) & 7); }

// grid = (ceil(N / (32 NB WPB)), ceil(K / 64 / cw)); block = 64 WPB; dynamic LDS = 2 stages x 3 planes x 32 mt rows x CPW x 128 B.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16.  The first version used the 32x32x16 shape, whose operand layout makes a weight load
// instruction take 32 bytes from each of 32 rows (four instructions share every 128-byte line).  tools/micro/frag_bw.hip
// measures that pattern on this launch's exact split (36 x 7 workgroups, 113 MB): 3.9 TB/s however many rounds are in flight,
// against 5.3 TB/s for 64 bytes from each of 16 rows - the 16x16x32 layout - and 5.9 for whole lines; the "latency" the
// weights seemed to be bound by was the address path.  The smaller shape also pads 38 rows to 48 instead of 64 (68 to 80
// instead of 96), needs one activation fragment per 2 NB weight tiles and k step (18 ds_read_b128 per wave and round at 38 rows
// instead of 24) and 4 accumulator registers per tile pair instead of 16.
//
// NB = 32-row weight groups per wave (1, or 2 for n <= 64): a wave owns 2 NB weight tiles of 16 rows; every activation
// fragment read from LDS feeds 2 NB x 3 MFMAs.
//
// Software pipeline over the rounds (CPW chunks each) of a workgroup's K range - measured without it (gpurun_out/p6): weights
// 27 us + MFMAs 20 us + activations 7 us + skeleton 9 us ADDED UP to 75 us, because every round waited for its loads and then
// computed.  Now the activations of round r + 1 (LDS-DMA into the other LDS stage - only the row groups that hold real rows:
// the DMA path moves ~15 B/clk/CU, vox_gemm_planes.h - or f32 rows held in registers until after the MFMAs) and its weights
// (second register set) are requested before the MFMAs of round r.  (Deeper: a third LDS stage, and later three weight
// register sets, were measured and bought nothing - gpurun_out/p8, rg3.)
// MTM = accumulator budget in 16-row activation tiles (4: n <= 64, 8: n <= 128; NB = 2 goes with 4 only): accumulators of tiles
// that hold no rows still occupy registers through the whole loop.
template <int WPB, int CPW, int XMODE, int NB, int MTM>
__global__ __launch_bounds__(64 * WPB) void k_rowsgemm(const RowsGemmArgs a) {
    static_assert(MTM == 4 || (MTM == 8 && NB == 1), "accumulator budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
    constexpr int P = 8 * CPW;                                   // 16-byte slots per activation row and plane
    constexpr int NT = 64 * WPB;
    constexpr int WT = 2 * NB;                                   // 16-row weight tiles per wave
    constexpr int MTMAX = MTM;                                   // 16-row activation tiles
    // (row, slot) items of an f32 activation stage per thread, at most: rows * P / threads
    constexpr int RG_F32_ITEMS = WPB == 8 ? 3 : 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kb = lane >> 4;
    const int mt = (a.n + 15) >> 4, rows = 16 * mt;
    const int plane_bytes = rows * P * 16, stage_bytes = 3 * plane_bytes;
    const int nchunks = a.K / 64;
    const int c_begin = blockIdx.y * a.cw, c_end = min(nchunks, c_begin + a.cw);
    const int row0 = (blockIdx.x * WPB + wave) * 32 * NB;        // this wave's weight tiles: rows [row0, row0 + 32 NB)
    const uint16_t *wrow[WT];
#pragma unroll
    for (int q = 0; q < WT; q++) wrow[q] = a.W + (size_t)min(row0 + q * 16 + li, a.N - 1) * a.K + kb * 8;

    // Weight fragments of a round: plain (L1-allocating) loads, 64 bytes from each of 16 rows per instruction, the two k steps of
    // a chunk share their 128-byte lines.  Chunks past the range re-read the last valid one and are never multiplied.
    // (No run-time switches in here: a debug branch around these loads - "load, or write a constant into the same register" - made
    // the compiler wait for vmcnt(0) in front of every round's MFMAs, i.e. for the NEXT round's loads; rounds 1 and 2 of this
    // kernel's life ran without any overlap of loads and MFMAs because of it.)
    auto load_w = [&](uint4 (&wr)[WT][CPW][2], int cb) {
#pragma unroll
        for (int c = 0; c < CPW; c++)
#pragma unroll
            for (int q = 0; q < WT; q++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    wr[q][c][ks] = *reinterpret_cast<const uint4 *>(wrow[q] + (size_t)min(cb + c, nchunks - 1) * 64 + ks * 32);
                }
    };
    // Activations of chunks [cb, cb + nc) into LDS stage `st`: [plane][row][slot], slot s of row r at physical slot s ^ rg_sw(r).
    // Planes: LDS-DMA, instruction q lands 64 consecutive slots (1 KB = 64 / P rows); which global 16-byte piece a lane fetches
    // is free.  Row groups past the last real row are not fetched: their LDS rows hold whatever they held, and an MFMA's output
    // row depends on its own input row only.
    auto stage_planes = [&](int st, int cb, int nc) {
        const int ipp = (rows * P) / 64;                         // instructions per plane
        const int ninstr = 3 * ipp;
        for (int q = wave; q < ninstr; q += WPB) {
            const int pl = q / ipp, qi = q - pl * ipp;
            if (qi * (64 / P) >= a.n) continue;
            const int rem = qi * 64 + lane;
            const int r = rem / P, ls = (rem - r * P) ^ rg_sw<P>(r);
            const int kk = min(ls * 8, nc * 64 - 8);             // (pieces past a short last round are never multiplied)
            glds16(a.Xp + (size_t)pl * a.xp_plane + (size_t)min(r, a.n - 1) * a.K + cb * 64 + kk,
                   lds_addr(rg_lds) + (unsigned)(st * stage_bytes) + (unsigned)q * 1024u);
        }
    };
    // f32 rows: the loads of a stage into registers (before the MFMAs of the previous round) ...
    auto f32_load = [&](float4 (&xa)[RG_F32_ITEMS], float4 (&xb)[RG_F32_ITEMS], int cb, int nc) {
#pragma unroll
        for (int it = 0; it < RG_F32_ITEMS; it++) {
            const int idx = min(it * NT + tid, rows * P - 1);
            const int r = idx / P, ls = idx - r * P;
            const int kk = min(ls * 8, nc * 64 - 8);
            const float *src = a.X + (size_t)min(r, a.n - 1) * a.ldx + cb * 64 + kk;
            xa[it] = *reinterpret_cast<const float4 *>(src);      // (items past the stage re-read its last piece and are not stored)
            xb[it] = *reinterpret_cast<const float4 *>(src + 4);
        }
    };
    // ... and the exact 3-term split into the stage (after them)
    auto f32_store = [&](int st, const float4 (&xa)[RG_F32_ITEMS], const float4 (&xb)[RG_F32_ITEMS]) {
#pragma unroll
        for (int it = 0; it < RG_F32_ITEMS; it++) {
            const int idx = it * NT + tid;
            if (idx < rows * P) {
                const int r = idx / P, ls = idx - r * P;
                const float xv[8] = {xa[it].x, xa[it].y, xa[it].z, xa[it].w, xb[it].x, xb[it].y, xb[it].z, xb[it].w};
                uint32_t h[8], m[8], l[8];
#pragma unroll
                for (int i = 0; i < 8; i++) split3(xv[i], h[i], m[i], l[i]);
                uint4 ph, pm, pq;
                ph.x = (h[0] >> 16) | h[1]; ph.y = (h[2] >> 16) | h[3]; ph.z = (h[4] >> 16) | h[5]; ph.w = (h[6] >> 16) | h[7];
                pm.x = (m[0] >> 16) | m[1]; pm.y = (m[2] >> 16) | m[3]; pm.z = (m[4] >> 16) | m[5]; pm.w = (m[6] >> 16) | m[7];
                pq.x = (l[0] >> 16) | (l[1] & 0xffff0000u); pq.y = (l[2] >> 16) | (l[3] & 0xffff0000u);
                pq.z = (l[4] >> 16) | (l[5] & 0xffff0000u); pq.w = (l[6] >> 16) | (l[7] & 0xffff0000u);
                unsigned char *dst = rg_lds + (size_t)st * stage_bytes + (size_t)(r * P + (ls ^ rg_sw<P>(r))) * 16;
                *reinterpret_cast<uint4 *>(dst) = ph;
                *reinterpret_cast<uint4 *>(dst + plane_bytes) = pm;
                *reinterpret_cast<uint4 *>(dst + 2 * plane_bytes) = pq;
            }
        }
    };

    f32x4 acc[WT][MTMAX];
#pragma unroll
    for (int q = 0; q < WT; q++)
#pragma unroll
        for (int t = 0; t < MTMAX; t++) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // MFMAs of one round: A = activation fragment (row = lane & 15 of tile t, 8 k of quarter lane >> 4), B = weight fragment.
    // One straight-line body per tile count (switch below), NOT "if (t < mt)" around the MFMAs of each tile: with the MFMAs of
    // the later tiles branched over at run time, the epilogue read the last accumulator registers of tile 0 before the matrix
    // pipe had written them (rows 19 .. 31 of every 32 x 32 tile wrong at mt = 1, right at mt = 2: tools/rg_gemm_check2.py) -
    // the wait states between an MFMA and a read of its result are inserted per basic block.
    auto compute_mt = [&](const uint4 (&wr)[WT][CPW][2], int st, int nc, auto MTc) {
        constexpr int MT = decltype(MTc)::value;
        const unsigned char *stage = rg_lds + (size_t)st * stage_bytes;
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            if (c < nc) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int ls = c * 8 + ks * 4 + kb;
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        const int r = t * 16 + li;
                        const unsigned char *src = stage + (size_t)(r * P + (ls ^ rg_sw<P>(r))) * 16;
                        bf16x8_t fa[3];
#pragma unroll
                        for (int p = 0; p < 3; p++) fa[p] = *reinterpret_cast<const bf16x8_t *>(src + p * plane_bytes);
#pragma unroll
                        for (int q = 0; q < WT; q++) {
                            union { uint4 u; bf16x8_t v; } fb;
                            fb.u = wr[q][c][ks];
#pragma unroll
                            for (int p = 2; p >= 0; p--)         // small terms first
                                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[p], fb.v, acc[q][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    auto compute = [&](const uint4 (&wr)[WT][CPW][2], int st, int nc) {
        switch (mt) {
            case 1: compute_mt(wr, st, nc, std::integral_constant<int, 1>{}); break;
            case 2: compute_mt(wr, st, nc, std::integral_constant<int, 2>{}); break;
            case 3: compute_mt(wr, st, nc, std::integral_constant<int, 3>{}); break;
            case 4: compute_mt(wr, st, nc, std::integral_constant<int, 4>{}); break;
            default:
                if constexpr (MTM == 8) {
                    switch (mt) {
                        case 5: compute_mt(wr, st, nc, std::integral_constant<int, 5>{}); break;
                        case 6: compute_mt(wr, st, nc, std::integral_constant<int, 6>{}); break;
                        case 7: compute_mt(wr, st, nc, std::integral_constant<int, 7>{}); break;
                        default: compute_mt(wr, st, nc, std::integral_constant<int, 8>{}); break;
                    }
                }
                break;
        }
    };

    // Two weight register sets, used alternately (no copies between them: a copy "w = wn" is a read of registers with loads in
    // flight, and wherever the compiler schedules it, it waits for them there - it put those waits in front of the MFMAs).
    uint4 wA[WT][CPW][2], wB[WT][CPW][2];
    float4 xa[RG_F32_ITEMS], xb[RG_F32_ITEMS];                  // (dead in the planes variant)
    load_w(wA, c_begin);                                         // the longest latency first
    if constexpr (XMODE == RG_X_PLANES) stage_planes(0, c_begin, min(CPW, c_end - c_begin));
    else { f32_load(xa, xb, c_begin, min(CPW, c_end - c_begin)); f32_store(0, xa, xb); }
    int st = 0, cb = c_begin;
    auto round = [&](const uint4 (&wcur)[WT][CPW][2], uint4 (&wnext)[WT][CPW][2]) {
        const int nc = min(CPW, c_end - cb);                     // chunks of this round
        const bool more = cb + CPW < c_end;
        // this round's activations are in LDS stage st (DMAs landed / stores done) and its weights in wcur; everybody is past the
        // MFMAs of the previous round, so the other stage may be overwritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) {                                              // round r + 1 streams under the MFMAs of round r
            const int ncn = min(CPW, c_end - cb - CPW);
            load_w(wnext, cb + CPW);
            if constexpr (XMODE == RG_X_PLANES) stage_planes(st ^ 1, cb + CPW, ncn);
            else f32_load(xa, xb, cb + CPW, ncn);
        }
        compute(wcur, st, nc);
        if (more) {
            if constexpr (XMODE == RG_X_F32) f32_store(st ^ 1, xa, xb);
        }
        cb += CPW; st ^= 1;
    };
    while (cb < c_end) {
        round(wA, wB);
        if (cb >= c_end) break;
        round(wB, wA);
    }
    // ---- raw partial sums: C layout of the 16 x 16 MFMA: column = lane & 15, row = 4 (lane >> 4) + r ----
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // (belt and braces for the hazard described at compute_mt)
#pragma unroll
    for (int q = 0; q < WT; q++) {
        const int col = row0 + q * 16 + li;
        if (col < a.N) {
            float *P0 = a.partial + (size_t)blockIdx.y * a.n * a.N + col;
#pragma unroll
            for (int t = 0; t < MTMAX; t++) {
                if (t < mt) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int m = t * 16 + 4 * kb + r;
                        if (m < a.n) P0[(size_t)m * a.N] = acc[q][t][r];
                    }
                }
            }
        }
    }
}

// ---- consumers of the partial sums -------------------------------------------------------------------------------------
// (k_rows_finish, vox_skinny.h: x += sum + bias, next RMSNorm (+ ada) -> f32 rows and / or bf16 planes)

// qkv[m][c] = sum_s partial[s][m][c] + bias[c]; interleaved-pair RoPE (voxtral_kernels.c:502-526) on columns < rope_cols with
// the chunk's table [n][head_dim / 2][cos, sin]; the k and v columns of every row also go to the position-indexed KV rings
// (slot = (pos0 + m) % ring_cap) when kring is given (voxtral_encoder.c:542-567, voxtral_decoder.c:466-480).
// One thread = 4 consecutive columns (two RoPE pairs).  grid-stride over n * N3 / 4.
__global__ __launch_bounds__(256) void k_qkv_finish(float *qkv, int N3, const float *partial, int nsplit, int n, const float *bias,
                                                    const float *rope_tab, int rope_cols, int head_dim, float *kring, float *vring,
                                                    int ring_cap, int kv_dim, int pos0, int q_cols) {
    const size_t total4 = (size_t)n * N3 / 4, slice = (size_t)n * N3;
    const int n4 = N3 / 4, half = head_dim / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / n4), c = (int)(i - (size_t)m * n4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z0 = 0; z0 < nsplit; z0 += 8) {                 // batches of 8 loads in flight, added in split order
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = *reinterpret_cast<const float4 *>(partial + (size_t)min(z0 + u, nsplit - 1) * slice + (size_t)m * N3 + c);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool on = z0 + u < nsplit;
                v.x += on ? p[u].x : 0.f; v.y += on ? p[u].y : 0.f; v.z += on ? p[u].z : 0.f; v.w += on ? p[u].w : 0.f;
            }
        }
        if (bias) { const float4 b = *reinterpret_cast<const float4 *>(bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (c < rope_cols) {
            const int d = (c % head_dim) >> 1;                   // pairs d, d + 1
            const float4 cs = *reinterpret_cast<const float4 *>(rope_tab + ((size_t)m * half + d) * 2);     // cos0 sin0 cos1 sin1
            const float a0 = v.x * cs.x - v.y * cs.y, a1 = v.x * cs.y + v.y * cs.x;
            const float b0 = v.z * cs.z - v.w * cs.w, b1 = v.z * cs.w + v.w * cs.z;
            v = make_float4(a0, a1, b0, b1);
        }
        *reinterpret_cast<float4 *>(qkv + (size_t)m * N3 + c) = v;
        if (kring && c >= q_cols) {
            const int slot = (pos0 + m) % ring_cap;
            if (c < q_cols + kv_dim) *reinterpret_cast<float4 *>(kring + (size_t)slot * kv_dim + (c - q_cols)) = v;
            else *reinterpret_cast<float4 *>(vring + (size_t)slot * kv_dim + (c - q_cols - kv_dim)) = v;
        }
    }
}

// h[m][j] = silu(sum_s partial[s][m][j]) * sum_s partial[s][m][H + j]  (voxtral_encoder.c:598-606, voxtral_decoder.c:684-687),
// written as the bf16 planes [3][n][H] the W2 launch consumes.  partial rows are 2 H wide (W = [w1; w3]).
// hf (optional, fp8 mode): the gated rows also as f32 [n][H] for k_rowsgemm_f8, which splits them into e4m3 terms itself.
__global__ __launch_bounds__(256) void k_swiglu_finish(uint16_t *planes, size_t plane, const float *partial, int nsplit, int n, int H, float *hf = nullptr) {
    const size_t total4 = (size_t)n * H / 4, slice = (size_t)n * 2 * H;
    const int h4 = H / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / h4), j = (int)(i - (size_t)m * h4) * 4;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f), u = g;
        for (int z0 = 0; z0 < nsplit; z0 += 4) {
            float4 pg[4], pu[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float *src = partial + (size_t)min(z0 + q, nsplit - 1) * slice + (size_t)m * 2 * H + j;
                pg[q] = *reinterpret_cast<const float4 *>(src); pu[q] = *reinterpret_cast<const float4 *>(src + H);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool on = z0 + q < nsplit;
                g.x += on ? pg[q].x : 0.f; g.y += on ? pg[q].y : 0.f; g.z += on ? pg[q].z : 0.f; g.w += on ? pg[q].w : 0.f;
                u.x += on ? pu[q].x : 0.f; u.y += on ? pu[q].y : 0.f; u.z += on ? pu[q].z : 0.f; u.w += on ? pu[q].w : 0.f;
            }
        }
        const float4 o = make_float4(silu(g.x) * u.x, silu(g.y) * u.y, silu(g.z) * u.z, silu(g.w) * u.w);
        if (hf) { *reinterpret_cast<float4 *>(hf + (size_t)m * H + j) = o; continue; }
        uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3(o.x, h0, m0, l0); split3(o.y, h1, m1, l1); split3(o.z, h2, m2, l2); split3(o.w, h3, m3, l3);
        uint16_t *dst = planes + (size_t)m * H + j;
        *reinterpret_cast<uint2 *>(dst) = make_uint2((h0 >> 16) | h1, (h2 >> 16) | h3);
        *reinterpret_cast<uint2 *>(dst + plane) = make_uint2((m0 >> 16) | m1, (m2 >> 16) | m3);
        *reinterpret_cast<uint2 *>(dst + 2 * plane) = make_uint2((l0 >> 16) | (l1 & 0xffff0000u), (l2 >> 16) | (l3 & 0xffff0000u));
    }
}

}  // namespace vox
