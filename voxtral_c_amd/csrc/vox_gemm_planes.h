// vox_gemm_planes.h — the large-M GEMM on pre-split activations, for gfx950.
//
//     y[M,N] = sum_p Xp[p][M,K] (bf16 planes hi, mid, lo) . W[N,K]^T (bf16)        p = 0..2
//
// Same arithmetic as k_gemm_mfma_bf16x3 (vox_gemm.h: an f32 activation is EXACTLY hi + mid + lo in bf16, three
// v_mfma_f32_32x32x16_bf16 passes with exact products and f32 accumulation = the oracle's cblas_sgemm up to summation
// order, voxtral_kernels.c:197-240) and the same tiling (128 x 128 per 256-thread workgroup, 2 x 2 waves of 64 x 64, so
// the epilogue and the split-K reduce are shared), but the f32 -> 3 x bf16 split is done ONCE by whoever produces the
// activations (RMSNorm, attention output, SwiGLU gate: k_*_planes below) instead of by every one of the N / 128 tiles that
// consume them, on the way into LDS, in the tile loop.  Measured on k_gemm_mfma_bf16x3 (profiles/r02_pmc_encoder_mfma.json):
// 23-28 % MFMA utilisation - per K slice the split costs about as many VALU cycles as the slice's MFMAs, sits between two
// barriers, and the register-staged global loads are one slice ahead only.
//
// Here a K slice (32 wide: 3 x 8 KB of A planes + 8 or 16 KB of B) goes from global memory straight into LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no VALU), NSTAGE slices deep, one barrier per slice.  A DMA instruction lands 1 KB
// linearly (16 rows x 64 B); which 16-byte global chunk a lane fetches is free, so chunk c of row r is stored at slot
// c ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 service group (16 consecutive rows, same k chunk) then hit 16 distinct
// 16-byte slots of the 256-byte bank window - conflict-free without padding.
//
// Measured (1664-row encoder chunk, w1;w3: N = 10240, K = 1280, 1040 tiles; rocprofv3 averages):
//   k_gemm_mfma_bf16x3 205 us  ->  this kernel 178 us (736 TF/s on the bf16 pipe = 29 % of the dense peak).
//   With the MFMAs removed the kernel still takes 147 us: 1.33 GB of tiles in 147 us = 9 TB/s = ~15 B/clk/CU through
//   global_load_lds, independent of request contiguity (whole 1 KB runs: same) and of pipeline depth (3 or 4 stages at one
//   workgroup per CU: slower); with the DMAs removed 114 us.  Staging the same tiles through registers (16-byte loads +
//   ds_write_b128): 633 us.  TN = 4 (128 x 256 tiles, 1.6 x fewer bytes per MFMA): no faster at M = 1664 (520 tiles on 512
//   slots), -5 .. -8 % on the encoder pass of the 300 s clip (M = 16946; round 5: the launchers take it from 600 wide tiles on,
//   profiles/r05_wide_tiles_ab.txt).  The 128 x 128 / one-barrier-per-slice structure is transport-bound at ~30 %; going further
//   needs the 256 x 256 deep-pipelined structure, whose tile count (7 x 40) does not fill the chip at the 30 s clip's M.
#pragma once
#include "vox_gemm.h"

namespace vox {

constexpr int GP_K = 32;                              // k per stage
constexpr int GP_ROWB = GP_K * 2;                     // bytes per row per stage (64)
constexpr int GP_PLANE_BYTES = 128 * GP_ROWB;         // one 128-row A plane tile: 8 KB
constexpr int gp_stage_bytes(int TN) { return 3 * GP_PLANE_BYTES + 64 * TN * GP_ROWB; }   // 3 A planes + B (64 TN rows)

// TN = MFMA tiles per wave along N: the workgroup's tile is 128 x (64 TN), a wave's 64 x (32 TN).  Per 16-wide k step a wave
// reads 6 A fragments (2 row tiles x 3 planes) and TN B fragments for 6 TN MFMAs: with TN = 2 the LDS (fragment reads + the
// DMA writes) is as busy as the matrix pipe (measured: 31 % MFMA utilisation, barely better than the kernel this replaces);
// TN = 4 shares every A fragment between four B tiles (10 reads per 24 MFMAs).
enum { GP_EPI_STD = 0, GP_EPI_SWIGLU = 1, GP_EPI_ROPE = 2 };

// SwiGLU launches: row `row` (0 .. 64 TN - 1) of workgroup bx's B tile -> row of [w1; w3] (N hidden columns each)
template <int TN>
__device__ __forceinline__ int gp_swiglu_row(int row, int bx, int N) {
    const int wn = row / (32 * TN), t = (row >> 5) % TN;
    return (t & 1) * N + min(bx * (32 * TN) + wn * (16 * TN) + (t >> 1) * 32 + (row & 31), N - 1);
}

// BD ("B direct", round 3): the weight fragments do not go through LDS at all - a lane's B operand (W[col][k .. k + 7], 16 bytes)
// is loaded straight from global memory into the register the MFMA reads, one slice ahead, with plain (L1-allocating) loads
// (fragment layout: an instruction takes 32 bytes from each of 32 rows, the two k steps of a slice and the next slice share the
// 128-byte line).  The kernel is bound by the LDS-DMA transport (~15 B/clk/CU, header): this takes the B quarter of every stage
// (8 of 32 KB at TN = 2) off that path and a quarter of the fragment reads off the LDS.
template <int NSTAGE, int TN, int EPI = GP_EPI_STD, bool BD = false>
__global__ __launch_bounds__(256, 2) void k_gemm_planes(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    constexpr int STAGE = BD ? 3 * GP_PLANE_BYTES : gp_stage_bytes(TN);
    constexpr int NINSTR = STAGE / 1024, IPW = NINSTR / 4;          // DMA instructions (1 KB each) per stage, per wave
    static_assert(!BD || NSTAGE == 2, "the direct-B pipeline is written for two stages (everything of slice t is waited for at the top of iteration t)");
    constexpr int BN = 64 * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Tile of this workgroup.  Split-K launches (wo, w2: 10 N tiles x 13 M tiles x 3 K slices at the 30 s clip) come as a 1-D grid in
    // XCD-aware order: workgroups go to the eight XCDs round robin in launch order, and the (M tile, K slice) groups - the 10 workgroups
    // that read the same 3 x 128 x K/3 slab of activation planes - are dealt to the XCDs whole, so that a slab is fetched into ONE L2
    // instead of up to eight (with the plain 3-D grid, whose x extent is not a multiple of 8, every L2 ended up reading most of A).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_tn > 0) {
        const int lid = blockIdx.x, k = lid >> 3, gi = (lid & 7) + 8 * (k / a.xcd_tn);
        if (gi >= a.xcd_tm * a.ksplit) return;                       // padding workgroups of the last round of groups
        bx = k % a.xcd_tn; by = gi % a.xcd_tm; bz = gi / a.xcd_tm;
    }
    const int bm0 = by * GB_M, bn0 = bx * BN;
    const int M = a.M, N = a.N, K = a.K;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- DMA sources of this wave: instruction q of a stage (q = wave * IPW + i) lands at stage + 1 KB * q and brings 16 rows:
    //      q < 24: rows 16 (q & 7) .. of A plane q >> 3;  q >= 24: rows 16 (q - 24) .. of B ----
    const int drow = lane >> 2, dslot = lane & 3;
    const int dchunk = dslot ^ ((drow >> 2) & 3);                    // (16 i + drow) >> 2 & 3 == drow >> 2 & 3
    const unsigned char *src[IPW];
#pragma unroll
    for (int i = 0; i < IPW; i++) {
        const int q = wave * IPW + i;
        if (q < 24) {
            const int row = 16 * (q & 7) + drow;
            src[i] = reinterpret_cast<const unsigned char *>(a.Xp + (size_t)(q >> 3) * a.xp_plane + (size_t)min(bm0 + row, M - 1) * a.ldxp) + dchunk * 16;
        } else {
            const int row = 16 * (q - 24) + drow;
            int wrow = min(bn0 + row, N - 1);
            if constexpr (EPI == GP_EPI_SWIGLU) {
                // a workgroup covers 32 TN hidden columns; tile row r = 32 TN wn + 32 tn + li holds gate (tn even: w1) or up (tn odd: w3)
                // of column 32 TN blockIdx.x + 16 TN wn + 32 (tn >> 1) + li, so that a lane's accumulator tiles 2 j, 2 j + 1 are the pair the gate needs
                static_assert(EPI != GP_EPI_SWIGLU || (TN & 1) == 0, "the SwiGLU epilogue pairs the N tiles of a wave");
                wrow = gp_swiglu_row<TN>(row, bx, N);
            }
            src[i] = reinterpret_cast<const unsigned char *>(a.W + (size_t)wrow * K) + dchunk * 16;
        }
    }
    const unsigned lds_wave = lds_addr(gp_smem) + (unsigned)(wave * IPW) * 1024u;
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < IPW; i++) glds16(src[i] + (size_t)kt * GP_ROWB, lds_wave + (unsigned)buf * STAGE + (unsigned)i * 1024u);
    };

    // ---- fragment addresses (bytes inside a stage): row r of a tile at r * 64, k chunk c at slot c ^ ((r >> 2) & 3) ----
    const int li = lane & 31, lg = lane >> 5;
    int aoff[2], boff[TN], asw[2], bsw[TN];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int ra = wm * 64 + t * 32 + li;
        aoff[t] = ra * GP_ROWB; asw[t] = (ra >> 2) & 3;
    }
#pragma unroll
    for (int t = 0; t < TN; t++) {
        const int rb = wn * (32 * TN) + t * 32 + li;
        boff[t] = 3 * GP_PLANE_BYTES + rb * GP_ROWB; bsw[t] = (rb >> 2) & 3;
    }

    const int nk_total = K / GP_K;
    const int kt0 = (a.ksplit > 1) ? bz * a.kper : 0;
    const int kt1 = (a.ksplit > 1) ? min(nk_total, kt0 + a.kper) : nk_total;
    const int nk = kt1 - kt0;

    // BD: this lane's weight rows (one per N tile of the wave) and the loader of a slice's fragments
    const uint16_t *bsrc[TN];
#pragma unroll
    for (int tt = 0; tt < TN; tt++) {
        const int row = wn * (32 * TN) + tt * 32 + li;               // row of the workgroup's B tile
        int wrow = min(bn0 + row, N - 1);
        if constexpr (EPI == GP_EPI_SWIGLU) wrow = gp_swiglu_row<TN>(row, bx, N);
        bsrc[tt] = a.W + (size_t)wrow * K + lg * 8;
    }
    uint4 bcur[TN][2], bnxt[TN][2];
    auto load_b = [&](uint4 (&b)[TN][2], int kt) {
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int tt = 0; tt < TN; tt++) b[tt][k2] = *reinterpret_cast<const uint4 *>(bsrc[tt] + (size_t)kt * GP_K + k2 * 16);
    };

    auto compute = [&](const unsigned char *st) {
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const int c = 2 * k2 + lg;
            bf16x8_t af[2][3], bf[TN];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int p = 0; p < 3; p++)
                    af[tt][p] = *reinterpret_cast<const bf16x8_t *>(st + p * GP_PLANE_BYTES + aoff[tt] + ((c ^ asw[tt]) << 4));
#pragma unroll
            for (int tt = 0; tt < TN; tt++) {
                if constexpr (BD) { union { uint4 u; bf16x8_t v; } cv; cv.u = bcur[tt][k2]; bf[tt] = cv.v; }
                else bf[tt] = *reinterpret_cast<const bf16x8_t *>(st + boff[tt] + ((c ^ bsw[tt]) << 4));
            }
#pragma unroll
            for (int p = 2; p >= 0; p--)                       // small terms first
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < TN; tn++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][p], bf[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    if constexpr (BD) {
        if (nk > 0) { load_b(bcur, kt0); issue(kt0, 0); }
        for (int t = 0; t < nk; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slice t: this wave's DMAs (issued an iteration ago) and its B fragments
            __syncthreads();                               // everybody's part of slice t is in; everybody is done reading slice t - 1
            if (t + 1 < nk) { load_b(bnxt, kt0 + t + 1); issue(kt0 + t + 1, (t + 1) & 1); }
            compute(gp_smem + (size_t)(t & 1) * STAGE);
            if (t + 1 < nk) {
#pragma unroll
                for (int tt = 0; tt < TN; tt++) { bcur[tt][0] = bnxt[tt][0]; bcur[tt][1] = bnxt[tt][1]; }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; s++)
            if (s < nk) issue(kt0 + s, s);
        for (int t = 0; t < nk; t++) {
            // this wave's DMAs of slice t have landed once only the younger slices' (IPW instructions each) are outstanding
            if (NSTAGE >= 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // everybody's part of slice t is in; everybody is done reading slice t - 1
            if (t + NSTAGE - 1 < nk) issue(kt0 + t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
            compute(gp_smem + (size_t)(t % NSTAGE) * STAGE);
        }
    }
    if constexpr (EPI == GP_EPI_SWIGLU) {
        // h = silu(gate) * up (voxtral_encoder.c:598-606), written as the bf16 planes the W2 launch consumes
#pragma unroll
        for (int j = 0; j < TN / 2; j++) {
            const int col = bx * (32 * TN) + wn * (16 * TN) + j * 32 + li;
            if (col < N) {
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = bm0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                        if (row < M) {
                            const float hv = silu(acc[tm][2 * j][r]) * acc[tm][2 * j + 1][r];
                            uint32_t ph, pm, pl;
                            split3(hv, ph, pm, pl);
                            uint16_t *dst = a.Yp + (size_t)row * N + col;
                            dst[0] = (uint16_t)(ph >> 16); dst[a.yp_plane] = (uint16_t)(pm >> 16); dst[2 * a.yp_plane] = (uint16_t)(pl >> 16);
                        }
                    }
            }
        }
    } else if constexpr (EPI == GP_EPI_ROPE) {
        // + bias, then the interleaved-pair RoPE (voxtral_kernels.c:502-526) on columns < rope_cols: the partner of column c is
        // c ^ 1 = the neighbouring lane (C layout: column = lane & 31), fetched with one DPP quad permute
        const int half = a.head_dim >> 1;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int col = bn0 + wn * (32 * TN) + tn * 32 + li;
                const float b = (a.bias && col < N) ? a.bias[col] : 0.f;
                const bool rot = col < a.rope_cols;
                const int pd = (col % a.head_dim) >> 1;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = bm0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                    float v = acc[tm][tn][r] + b;
                    const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
                    if (rot && row < M) {
                        const float2 cs = *reinterpret_cast<const float2 *>(a.rope_tab + ((size_t)row * half + pd) * 2);
                        v = (col & 1) ? other * cs.y + v * cs.x : v * cs.x - other * cs.y;
                    }
                    if (row < M && col < N) {
                        a.Y[(size_t)row * a.ldy + col] = v;
                        if (a.kring && col >= a.ring_col0 && row >= a.ring_row0) {
                            const int kc = col - a.ring_col0;
                            const int slot = (a.ring_pos0 + row - a.ring_row0) % a.ring_cap;
                            if (kc < a.ring_kvd) a.kring[(size_t)slot * a.ring_kvd + kc] = v;
                            else a.vring[(size_t)slot * a.ring_kvd + kc - a.ring_kvd] = v;
                        }
                    }
                }
            }
    } else {
        gemm_epilogue<TN>(a, acc, bm0, bn0, wm, wn, li, lg, bz);
    }
}

// ---- producers of the planes --------------------------------------------------------------------------------------
// 4 consecutive f32 -> 4 bf16 in each of the three planes (element k at the lower address)
__device__ __forceinline__ void planes_store4(uint16_t *p0, size_t plane, const float4 v) {
    uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(v.x, h0, m0, l0); split3(v.y, h1, m1, l1); split3(v.z, h2, m2, l2); split3(v.w, h3, m3, l3);
    uint2 ph, pm, pl;
    ph.x = (h0 >> 16) | h1; ph.y = (h2 >> 16) | h3;
    pm.x = (m0 >> 16) | m1; pm.y = (m2 >> 16) | m3;
    pl.x = (l0 >> 16) | (l1 & 0xffff0000u); pl.y = (l2 >> 16) | (l3 & 0xffff0000u);
    *reinterpret_cast<uint2 *>(p0) = ph;
    *reinterpret_cast<uint2 *>(p0 + plane) = pm;
    *reinterpret_cast<uint2 *>(p0 + 2 * plane) = pl;
}

// RMSNorm over rows (k_rmsnorm_rows, voxtral_kernels.c:346-363) with the result written as planes [3][n][D].
__global__ __launch_bounds__(256) void k_rmsnorm_planes(uint16_t *planes, size_t plane, const float *x, int ldx, const float *w,
                                                        const float *ada, int D, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float *xr = x + (size_t)row * ldx;
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
        planes_store4(planes + (size_t)row * D + i, plane, v);
    }
}

// Split-K reduce + epilogue (k_splitk_reduce: partials in split order, bias, residual) of a launch whose output is the residual stream, and the
// RMSNorm that follows it (k_rmsnorm_planes) in ONE pass over the row: block = row (round 6: two ~5 us launches less per layer of a large chunk).
__global__ __launch_bounds__(256) void k_splitk_reduce_norm_planes(const GemmArgs a, uint16_t *planes, size_t plane, const float *w, const float *ada, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, N = a.N;
    const size_t total = (size_t)a.M * N;
    float *yr = a.Y + (size_t)row * a.ldy;
    float ss = 0.f;
    for (int i = tid * 4; i < N; i += 1024) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < a.ksplit; z++) {
            const float4 p = *reinterpret_cast<const float4 *>(a.partial + (size_t)z * total + (size_t)row * N + i);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (a.bias) { const float4 b = *reinterpret_cast<const float4 *>(a.bias + i); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (a.resid) {
            const float4 r = *reinterpret_cast<const float4 *>(a.resid + (size_t)row * a.ldr + i);
            v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
        }
        *reinterpret_cast<float4 *>(yr + i) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
    for (int i = tid * 4; i < N; i += 1024) {
        float4 v = *reinterpret_cast<const float4 *>(yr + i);              // (this thread's own stores)
        const float4 g = *reinterpret_cast<const float4 *>(w + i);
        v.x = v.x * inv * g.x; v.y = v.y * inv * g.y; v.z = v.z * inv * g.z; v.w = v.w * inv * g.w;
        if (ada) {
            const float4 s = *reinterpret_cast<const float4 *>(ada + i);
            v.x *= (1.0f + s.x); v.y *= (1.0f + s.y); v.z *= (1.0f + s.z); v.w *= (1.0f + s.w);
        }
        planes_store4(planes + (size_t)row * N + i, plane, v);
    }
}

// x[M][K] (f32, row stride ldx) -> planes [3][M][K]
__global__ __launch_bounds__(256) void k_split_planes(uint16_t *planes, size_t plane, const float *x, int ldx, int M, int K) {
    const size_t total4 = (size_t)M * K / 4;
    const int k4 = K / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t m = i / k4;
        const int j = (int)(i % k4) * 4;
        planes_store4(planes + m * K + j, plane, *reinterpret_cast<const float4 *>(x + m * ldx + j));
    }
}

// h = silu(gu[:, :H]) * gu[:, H:]   (k_silu_mul) written as planes [3][M][H]
__global__ __launch_bounds__(256) void k_silu_mul_planes(uint16_t *planes, size_t plane, const float *gu, int M, int H) {
    const size_t total4 = (size_t)M * H / 4;
    const int h4 = H / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t m = i / h4;
        const int j = (int)(i % h4) * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gu + m * 2 * H + j);
        const float4 u = *reinterpret_cast<const float4 *>(gu + m * 2 * H + H + j);
        float4 o;
        o.x = silu(g.x) * u.x; o.y = silu(g.y) * u.y; o.z = silu(g.z) * u.z; o.w = silu(g.w) * u.w;
        planes_store4(planes + m * H + j, plane, o);
    }
}

}  // namespace vox
