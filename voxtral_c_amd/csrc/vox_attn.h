// vox_attn.h — causal sliding-window attention kernels (vox_causal_attention,
// voxtral_kernels.c:412-482) for the two shapes the model uses.
//
//  * k_attn_rows<64>  — encoder / large-M: MHA, head_dim 64, window 750.  One thread owns
//    one query row (q and the output accumulator live in registers, 64+64 VGPRs), a block
//    of 128 consecutive queries of one head walks the union of their key windows in
//    32-key tiles staged through LDS (all lanes read the same K/V row => LDS broadcast,
//    no bank conflicts).  Keys are visited in increasing position with the reference's own
//    online-softmax recurrence, so the arithmetic order per query matches the oracle.
//    FLOP-bound on the fp32 VALU (2*2*64 FLOP per (q,k) pair).
//
//  * k_attn_dec<128,4> — decoder (prefill rows and the M=1 step): GQA with 4 query heads per
//    KV head, head_dim 128, fp32 KV ring.  HBM-bound on the KV read, so one wave streams
//    every K/V row once for all 4 query heads: 16 lanes x 32 bytes cover one 512-byte head
//    row, a wave-instruction covers 4 keys; the 16-lane dot products are reduced with DPP
//    row operations; each 16-lane group keeps its own online-softmax state which is merged
//    once at the end (lane groups -> waves through LDS -> optional split-K partials).
//
// Key addressing is by logical position: keys with position < posB0 live in a ring
// (slot = pos % cap, filled by previous chunks), keys >= posB0 in a linear array (the
// current chunk's merged QKV buffer, or the caller's K/V for the kernel-level API).
// "Attend to physical rows max(0,p-W+1)..p of the compacted cache" in the reference is
// exactly "logical positions max(0,P-W+1)..P" here.
#pragma once
#include "vox_common.h"
#include "vox_gemm.h"

namespace vox {

struct AttnArgs {
    float *out; int ldo;            // [n_q, n_heads*HD]
    const float *q; int ldq;        // [n_q, ...] head h at column h*HD
    int n_q;
    int qpos0;                      // logical position of query row 0
    const float *kB, *vB; int ldB;  // linear segment: row r <-> position posB0 + r
    int posB0;                      // keys with position < posB0 come from the ring
    int last_key;                   // last key position that exists (k_end = min(gpos+1, seq_k))
    const float *kA, *vA;           // ring segment (positions < posB0): [capA][ldA]
    int capA, ldA;
    int n_heads, n_kv_heads;
    float scale;
    int window;
    const DecState *st;             // decoder step: qpos0 = st->pos (when non-null)
    // split-K (decoder)
    unsigned *arrive;               // k_attn_small: [n_heads] arrival counters (zero between launches): the last slice of a head to arrive merges it
    uint16_t *out_planes; size_t out_plane;   // k_attn_enc_bf16, one key slice: the output ALSO as bf16 planes [3][n_q][ldo] (the Wo launch's operand: no k_split_planes pass)
    int xcd_map;                    // k_attn_enc_bf16: remap (tile, head) so that a head's query tiles share an XCD (see there)
    int split_keys;                 // keys per blockIdx.y
    float *part_o, *part_ml;        // [n_q][n_heads][nsplit][HD], [..][2]
    int force_partials;             // write partials even when nsplit == 1 (merged by the Wo GEMV prologue)
};

template <int HD>
__global__ __launch_bounds__(128) void k_attn_rows(const AttnArgs a) {
    constexpr int TK = 32;
    __shared__ __attribute__((aligned(16))) float Ks[TK][HD];
    __shared__ __attribute__((aligned(16))) float Vs[TK][HD];
    const int tid = threadIdx.x;
    const int h = blockIdx.y;
    const int kvh = h / (a.n_heads / a.n_kv_heads);
    const int qi = blockIdx.x * 128 + tid;
    const bool valid = qi < a.n_q;
    const int P = a.qpos0 + qi;
    const int last_key = a.last_key;
    int lo_i = P - a.window + 1; if (lo_i < 0) lo_i = 0;
    int hi_i = P < last_key ? P : last_key;
    if (!valid) { lo_i = 1; hi_i = 0; }

    // block-wide key range
    const int q_first = blockIdx.x * 128;
    const int q_last = min(q_first + 127, a.n_q - 1);
    int blo = a.qpos0 + q_first - a.window + 1; if (blo < 0) blo = 0;
    int bhi = a.qpos0 + q_last; if (bhi > last_key) bhi = last_key;
    // optional split of the key range over blockIdx.z (small chunks: too few query tiles to
    // fill the chip); partial (m, l, o) go to part_* and k_attn_combine<HD> merges them
    const int nsplit = gridDim.z;
    if (nsplit > 1) {
        const int tiles = (bhi - blo + TK) / TK;
        const int per = (tiles + nsplit - 1) / nsplit;
        const int lo2 = blo + (int)blockIdx.z * per * TK;
        const int hi2 = lo2 + per * TK - 1;
        blo = lo2;
        if (hi2 < bhi) bhi = hi2;
    }

    float qv[HD], o[HD];
    if (valid) {
        const float *qp = a.q + (size_t)qi * a.ldq + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(qp + d);
            qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int d = 0; d < HD; d++) qv[d] = 0.f;
    }
#pragma unroll
    for (int d = 0; d < HD; d++) o[d] = 0.f;
    float m = -1e30f, l = 0.f;

    for (int t0 = blo; t0 <= bhi; t0 += TK) {
        __syncthreads();
        // cooperative tile load: TK rows x HD floats for K and V
        for (int i = tid; i < TK * (HD / 4); i += 128) {
            const int r = i / (HD / 4), c = (i % (HD / 4)) * 4;
            const int pos = t0 + r;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (pos <= bhi) {
                if (pos >= a.posB0) {
                    const size_t off = (size_t)(pos - a.posB0) * a.ldB + kvh * HD + c;
                    kk = *reinterpret_cast<const float4 *>(a.kB + off);
                    vv = *reinterpret_cast<const float4 *>(a.vB + off);
                } else {
                    const size_t off = (size_t)(pos % a.capA) * a.ldA + kvh * HD + c;
                    kk = *reinterpret_cast<const float4 *>(a.kA + off);
                    vv = *reinterpret_cast<const float4 *>(a.vA + off);
                }
            }
            *reinterpret_cast<float4 *>(&Ks[r][c]) = kk;
            *reinterpret_cast<float4 *>(&Vs[r][c]) = vv;
        }
        __syncthreads();
        const bool any_here = (t0 + TK - 1 >= lo_i) && (t0 <= hi_i);
        if (!__any(any_here)) continue;
#pragma unroll 1
        for (int j = 0; j < TK; j++) {
            const int pos = t0 + j;
            const bool in = (pos >= lo_i) && (pos <= hi_i);
            // four independent accumulation chains (a single 64-long fmaf chain is latency bound)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kk = *reinterpret_cast<const float4 *>(&Ks[j][d]);
                s0 = fmaf(qv[d], kk.x, s0); s1 = fmaf(qv[d + 1], kk.y, s1);
                s2 = fmaf(qv[d + 2], kk.z, s2); s3 = fmaf(qv[d + 3], kk.w, s3);
            }
            float s = ((s0 + s1) + (s2 + s3)) * a.scale;
            if (in) {
                // reference recurrence (voxtral_kernels.c:456-470), branch-free form
                const float mn = fmaxf(m, s);
                const float corr = expf(m - mn);
                const float p = expf(s - mn);
                l = l * corr + p;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4 *>(&Vs[j][d]);
                    o[d] = o[d] * corr + p * vv.x; o[d + 1] = o[d + 1] * corr + p * vv.y;
                    o[d + 2] = o[d + 2] * corr + p * vv.z; o[d + 3] = o[d + 3] * corr + p * vv.w;
                }
                m = mn;
            }
        }
    }
    if (valid && nsplit > 1) {
        const size_t pidx = ((size_t)qi * a.n_heads + h) * nsplit + blockIdx.z;
        float *po = a.part_o + pidx * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4 *>(po + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
        a.part_ml[pidx * 2] = m;
        a.part_ml[pidx * 2 + 1] = l;
    } else if (valid) {
        float *op = a.out + (size_t)qi * a.ldo + h * HD;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            float4 t;
            t.x = o[d] * inv; t.y = o[d + 1] * inv; t.z = o[d + 2] * inv; t.w = o[d + 3] * inv;
            *reinterpret_cast<float4 *>(op + d) = t;
        }
    }
}

// ---------------------------------------------------------------------------------
// k_attn_enc_bf16 (round 4) - encoder attention (MHA, head_dim 64) on the bf16 matrix pipe.
//
// A wave owns 32 consecutive queries of one head; a block = 4 waves = 128 queries, sharing 32-key K/V tiles staged in LDS.  Both
// products are computed TRANSPOSED so that a lane always owns one query (col = lane & 31 of the MFMA result) and the online softmax
// needs no cross-lane traffic except one exchange between the two 32-lane halves:
//     S^T[key][query] = K[key][:] . Q[query][:]      A = K tile (LDS), B = Q (registers)
//     O^T[dim][query] += V^T[dim][key] . P^T[key][query]
// P^T comes out of the first product in the C layout, which is what the second product wants as its B operand; V is stored
// transposed in LDS so that the matching A operand is one LDS read.  grid = (ceil(n_q / 128), heads, key splits); partials as in
// k_attn_rows.  (Rounds 1 - 3 ran this formulation on v_mfma_f32_32x32x2_f32 - exact f32 x f32 products, but that instruction runs
// at the f32 VECTOR rate, 1/16 of the bf16 MFMA rate: 64 of them per 32 x 32 (query, key) tile and wave = 4096 cycles, 34.8 % of
// that pipe's peak in the 30 s clip's big pass and the largest single item of the encoder pass, 5.8 of 21.7 ms.  That kernel,
// k_attn_enc_mfma, was removed in round 5; a failed start-up self-test of this one falls back to k_attn_rows.)
// Here every f32 operand is split EXACTLY into three bf16 terms (hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid:
// 24 = 3 x 8 significand bits, vox_gemm.h) and a product a . b is the six bf16 MFMAs hh + hm + mh + hl + lh + mm; the dropped
// terms (ml, lm, ll) are below 2^-24 of the product.  Per tile and wave: 4 k-steps x 6 for S^T = K . Q^T and 2 k-steps x 2 halves
// x 6 for O^T += V^T . P^T = 48 v_mfma_f32_32x32x16_bf16 = 1536 cycles.
// Same transposed formulation as above (a lane owns one query: column lane & 31 of both results), same online softmax, same
// staging (next tile in registers under this tile's math), same partial outputs.  Layout differences:
//   * K is kept as three bf16 planes [plane][key][64 dims] with 144-byte rows (16 lanes of a ds_read_b128 hit 16 distinct
//     16-byte slots), V as three transposed planes [plane][dim][32 keys] with 72-byte rows; the split happens on the way into LDS;
//   * the C layout of S^T hands a lane the keys (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the second product contracts over keys, so
//     any key order will do as long as both operands use the same one: k-step s takes registers 8 s .. 8 s + 7 as they are
//     (keys 16 s + 4 lg + {0..3} and 16 s + 8 + 4 lg + {0..3}) and reads V^T at exactly those keys (two 8-byte reads per plane).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void ab_split8(const float (&x)[8], bf16x8_t &h, bf16x8_t &m, bf16x8_t &l) {
    uint32_t hh[8], mm[8], ll[8];
#pragma unroll
    for (int i = 0; i < 8; i++) split3(x[i], hh[i], mm[i], ll[i]);
    union { uint32_t u[4]; bf16x8_t v; } ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ph.u[i] = (hh[2 * i] >> 16) | hh[2 * i + 1];
        pm.u[i] = (mm[2 * i] >> 16) | mm[2 * i + 1];
        pl.u[i] = (ll[2 * i] >> 16) | (ll[2 * i + 1] & 0xffff0000u);
    }
    h = ph.v; m = pm.v; l = pl.v;
}
// c += a . b with a = (ah, am, al), b = (bh, bm, bl): the six terms, small ones first
__device__ __forceinline__ f32x16 ab_mfma6(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

// (256, 3): 168 registers instead of 182 + 32 accumulation registers, no spills, a third workgroup per CU: nothing at the 30 s clip's
// 416 workgroups (20.9 -> 20.8 ms per encoder pass), -3.3 % on the 300 s clip's pass (140.0 -> 135.4 ms; profiles/r05_attn_occ_ab.txt)
__global__ __launch_bounds__(256, 3) void k_attn_enc_bf16(const AttnArgs a) {
    constexpr int HD = 64, TK = 32;
    constexpr int KROW = 144, KPL = TK * KROW;          // bytes per K row / plane
    constexpr int VROW = 72, VPL = HD * VROW;           // bytes per V^T row / plane
    __shared__ __attribute__((aligned(16))) unsigned char Kp[3 * KPL];
    __shared__ __attribute__((aligned(16))) unsigned char Vp[3 * VPL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lg = lane >> 5;
    int h = blockIdx.y, qtile = blockIdx.x;
    if (a.xcd_map && (gridDim.y & 7) == 0) {           // all query tiles of a head on one XCD (a head's K/V rows then stay in one L2)
        const int lid = blockIdx.x + gridDim.x * blockIdx.y, k = lid >> 3, hpx = gridDim.y >> 3;
        h = (lid & 7) + 8 * (k % hpx);
        qtile = k / hpx;
    }
    const int q_first = qtile * 128;
    const int qi = q_first + wave * 32 + li;            // this lane's query
    const bool qvalid = qi < a.n_q;
    const int P = a.qpos0 + qi;
    const int last_key = a.last_key;
    const int q_last = min(q_first + 127, a.n_q - 1);
    int blo = a.qpos0 + q_first - a.window + 1; if (blo < 0) blo = 0;
    int bhi = a.qpos0 + q_last; if (bhi > last_key) bhi = last_key;
    const int nsplit = gridDim.z;
    if (nsplit > 1) {
        const int tiles = (bhi - blo + TK) / TK;
        const int per = (tiles + nsplit - 1) / nsplit;
        const int lo2 = blo + (int)blockIdx.z * per * TK;
        const int hi2 = lo2 + per * TK - 1;
        blo = lo2;
        if (hi2 < bhi) bhi = hi2;
    }
    int lo_i = P - a.window + 1; if (lo_i < blo) lo_i = blo;
    int hi_i = P < bhi ? P : bhi;
    if (!qvalid) { lo_i = 1; hi_i = 0; }
    const int wq0 = a.qpos0 + q_first + wave * 32;
    const int w_lo = max(wq0 - a.window + 1, blo), w_hi = min(wq0 + 31, bhi);

    // Q planes of this lane's query: k-step t covers dims 16 t + 8 lg .. + 7 (B operand of S^T = K . Q^T)
    bf16x8_t qp[4][3];
    {
        const float *qsrc = a.q + (size_t)(qvalid ? qi : 0) * a.ldq + h * HD;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float4 v0 = *reinterpret_cast<const float4 *>(qsrc + 16 * t + 8 * lg);
            const float4 v1 = *reinterpret_cast<const float4 *>(qsrc + 16 * t + 8 * lg + 4);
            float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (!qvalid) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = 0.f;
            }
            ab_split8(x, qp[t][0], qp[t][1], qp[t][2]);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; r++) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -1e30f, l = 0.f;

    // staging: 32 rows x 16 float4 for K and for V -> 2 + 2 per thread 
    float4 rk[2], rv[2];
    auto load_tile = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, r = idx >> 4, c = (idx & 15) * 4;
            const int pos = t0 + r;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (pos <= bhi) {
                if (pos >= a.posB0) {
                    const size_t off = (size_t)(pos - a.posB0) * a.ldB + h * HD + c;
                    kk = *reinterpret_cast<const float4 *>(a.kB + off);
                    vv = *reinterpret_cast<const float4 *>(a.vB + off);
                } else {
                    const size_t off = (size_t)(pos % a.capA) * a.ldA + h * HD + c;
                    kk = *reinterpret_cast<const float4 *>(a.kA + off);
                    vv = *reinterpret_cast<const float4 *>(a.vA + off);
                }
            }
            rk[i] = kk; rv[i] = vv;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 256, r = idx >> 4, c = (idx & 15) * 4;
            uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            split3(rk[i].x, h0, m0, l0); split3(rk[i].y, h1, m1, l1); split3(rk[i].z, h2, m2, l2); split3(rk[i].w, h3, m3, l3);
            unsigned char *kd = Kp + r * KROW + c * 2;
            *reinterpret_cast<uint2 *>(kd) = make_uint2((h0 >> 16) | h1, (h2 >> 16) | h3);
            *reinterpret_cast<uint2 *>(kd + KPL) = make_uint2((m0 >> 16) | m1, (m2 >> 16) | m3);
            *reinterpret_cast<uint2 *>(kd + 2 * KPL) = make_uint2((l0 >> 16) | (l1 & 0xffff0000u), (l2 >> 16) | (l3 & 0xffff0000u));
            const float vx[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {                 // V transposed: [dim c + e][key r]
                uint32_t vh, vm, vl;
                split3(vx[e], vh, vm, vl);
                unsigned char *vd = Vp + (c + e) * VROW + r * 2;
                *reinterpret_cast<uint16_t *>(vd) = (uint16_t)(vh >> 16);
                *reinterpret_cast<uint16_t *>(vd + VPL) = (uint16_t)(vm >> 16);
                *reinterpret_cast<uint16_t *>(vd + 2 * VPL) = (uint16_t)(vl >> 16);
            }
        }
    };

    if (blo <= bhi) {
        load_tile(blo);
        store_tile();
    }
    __syncthreads();
    for (int t0 = blo; t0 <= bhi; t0 += TK) {
        const bool more = t0 + TK <= bhi;
        if (more) load_tile(t0 + TK);
        if (t0 + TK - 1 >= w_lo && t0 <= w_hi) {             // wave-uniform
            // ---- S^T = K . Q^T: A = K planes (row = key li, dims 16 t + 8 lg ..), B = Q planes ------------------------------
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; r++) sc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                bf16x8_t kf[3];
#pragma unroll
                for (int p = 0; p < 3; p++) kf[p] = *reinterpret_cast<const bf16x8_t *>(Kp + p * KPL + li * KROW + (16 * t + 8 * lg) * 2);
                sc = ab_mfma6(kf, qp[t], sc);
            }
            // ---- online softmax for this lane's query  ---------------------
            float mt = -1e30f;
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int kp = t0 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                ok[r] = (kp >= lo_i) && (kp <= hi_i);
                sc[r] = sc[r] * a.scale;
                if (ok[r]) mt = fmaxf(mt, sc[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(m, mt);
            const float corr = expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float pv = ok[r] ? expf(sc[r] - mn) : 0.f;
                sc[r] = pv;
                ps += pv;
            }
            l = l * corr + ps;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; r++) { o0[r] *= corr; o1[r] *= corr; }
            // ---- O^T += V^T . P^T: k-step s = registers 8 s .. 8 s + 7 = keys 16 s + 4 lg + {0..3}, 16 s + 8 + 4 lg + {0..3} ----
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                float px[8];
#pragma unroll
                for (int e = 0; e < 8; e++) px[e] = sc[8 * s2 + e];
                bf16x8_t pf[3];
                ab_split8(px, pf[0], pf[1], pf[2]);
                bf16x8_t v0[3], v1[3];
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const unsigned char *b0 = Vp + p * VPL + li * VROW + (16 * s2 + 4 * lg) * 2;
                    const unsigned char *b1 = Vp + p * VPL + (32 + li) * VROW + (16 * s2 + 4 * lg) * 2;
                    union { uint2 u[2]; bf16x8_t v; } c0, c1;
                    c0.u[0] = *reinterpret_cast<const uint2 *>(b0); c0.u[1] = *reinterpret_cast<const uint2 *>(b0 + 16);
                    c1.u[0] = *reinterpret_cast<const uint2 *>(b1); c1.u[1] = *reinterpret_cast<const uint2 *>(b1 + 16);
                    v0[p] = c0.v; v1[p] = c1.v;
                }
                o0 = ab_mfma6(v0, pf, o0);
                o1 = ab_mfma6(v1, pf, o1);
            }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- finish: the two lane halves hold disjoint keys (same m); O^T rows: dim (r & 3) + 8 (r >> 2) + 4 lg of each half ----
    l += __shfl_xor(l, 32, 64);
    if (!qvalid) return;
    if (nsplit > 1) {
        const size_t pidx = ((size_t)qi * a.n_heads + h) * nsplit + blockIdx.z;
        float *po = a.part_o + pidx * HD;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int d = 8 * q4 + 4 * lg;
            *reinterpret_cast<float4 *>(po + d) = make_float4(o0[4 * q4], o0[4 * q4 + 1], o0[4 * q4 + 2], o0[4 * q4 + 3]);
            *reinterpret_cast<float4 *>(po + 32 + d) = make_float4(o1[4 * q4], o1[4 * q4 + 1], o1[4 * q4 + 2], o1[4 * q4 + 3]);
        }
        if (lg == 0) { a.part_ml[pidx * 2] = m; a.part_ml[pidx * 2 + 1] = l; }
    } else {
        float *op = a.out + (size_t)qi * a.ldo + h * HD;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int d = 8 * q4 + 4 * lg;
            const float4 v0 = make_float4(o0[4 * q4] * inv, o0[4 * q4 + 1] * inv, o0[4 * q4 + 2] * inv, o0[4 * q4 + 3] * inv);
            const float4 v1 = make_float4(o1[4 * q4] * inv, o1[4 * q4 + 1] * inv, o1[4 * q4 + 2] * inv, o1[4 * q4 + 3] * inv);
            if (a.out_planes) {          // (round 6) straight into the Wo launch's operand layout
                uint16_t *pp = a.out_planes + (size_t)qi * a.ldo + h * HD;
                planes_store4(pp + d, a.out_plane, v0);
                planes_store4(pp + 32 + d, a.out_plane, v1);
            } else {
                *reinterpret_cast<float4 *>(op + d) = v0;
                *reinterpret_cast<float4 *>(op + 32 + d) = v1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Decoder attention.  grid = (n_kv_heads, nsplit, n_q); block = 256 (4 waves).
// ---------------------------------------------------------------------------------
template <int HD, int HPK, bool USE_DPP>
__global__ __launch_bounds__(256) void k_attn_dec(const AttnArgs a, const int nsplit) {
    static_assert(HD == 128, "16 lanes x 8 dims");
    __shared__ float sm_m[4][HPK], sm_l[4][HPK];
    __shared__ __attribute__((aligned(16))) float sm_o[4][HPK][HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = lane >> 4, dc = lane & 15;
    const int kvh = blockIdx.x, split = blockIdx.y, qi = blockIdx.z;
    static_assert(HPK == 4, "one wave per query head in the cross-wave merge");
    const int P = (a.st ? a.st->pos : a.qpos0) + qi;
    const int last_key = a.st ? P : a.last_key;
    int lo = P - a.window + 1; if (lo < 0) lo = 0;
    int hi = P < last_key ? P : last_key;
    // this block's slice, then this wave's quarter of it
    const int s_lo = lo + split * a.split_keys;
    int s_hi = s_lo + a.split_keys - 1; if (s_hi > hi) s_hi = hi;
    const int per_wave = a.split_keys / 4;
    const int w_lo = s_lo + wave * per_wave;
    int w_hi = w_lo + per_wave - 1; if (w_hi > s_hi) w_hi = s_hi;

    float qv[HPK][8], o[HPK][8], m[HPK], l[HPK];
#pragma unroll
    for (int h = 0; h < HPK; h++) {
        const float *qp = a.q + (size_t)qi * a.ldq + (kvh * HPK + h) * HD + dc * 8;
        const float4 t0 = *reinterpret_cast<const float4 *>(qp);
        const float4 t1 = *reinterpret_cast<const float4 *>(qp + 4);
        qv[h][0] = t0.x; qv[h][1] = t0.y; qv[h][2] = t0.z; qv[h][3] = t0.w;
        qv[h][4] = t1.x; qv[h][5] = t1.y; qv[h][6] = t1.z; qv[h][7] = t1.w;
        m[h] = -1e30f; l[h] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; d++) o[h][d] = 0.f;
    }

    // 16 keys per trip: the K/V loads of four consecutive 4-key groups are issued together so
    // that one trip costs one memory latency instead of four (the loop is latency-bound at the
    // short contexts of real-time decoding).  Out-of-range slots are clamped to the last valid
    // key for the load and masked with s = -inf (p = 0) for the arithmetic.
    constexpr int UNR = 4;
    for (int t = w_lo + ks; t <= w_hi; t += 4 * UNR) {
        float4 kq0[UNR], kq1[UNR], vq0[UNR], vq1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            int tt = t + 4 * u;
            if (tt > w_hi) tt = w_hi;
            const float *kp_, *vp_;
            if (tt >= a.posB0) {
                const size_t off = (size_t)(tt - a.posB0) * a.ldB + kvh * HD + dc * 8;
                kp_ = a.kB + off; vp_ = a.vB + off;
            } else {
                const size_t off = (size_t)(tt % a.capA) * a.ldA + kvh * HD + dc * 8;
                kp_ = a.kA + off; vp_ = a.vA + off;
            }
            kq0[u] = *reinterpret_cast<const float4 *>(kp_);
            kq1[u] = *reinterpret_cast<const float4 *>(kp_ + 4);
            vq0[u] = *reinterpret_cast<const float4 *>(vp_);
            vq1[u] = *reinterpret_cast<const float4 *>(vp_ + 4);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const bool ok = (t + 4 * u) <= w_hi;
            const float4 k0 = kq0[u], k1 = kq1[u], v0 = vq0[u], v1 = vq1[u];
#pragma unroll
            for (int h = 0; h < HPK; h++) {
                float s = qv[h][0] * k0.x;
                s = fmaf(qv[h][1], k0.y, s); s = fmaf(qv[h][2], k0.z, s); s = fmaf(qv[h][3], k0.w, s);
                s = fmaf(qv[h][4], k1.x, s); s = fmaf(qv[h][5], k1.y, s); s = fmaf(qv[h][6], k1.z, s);
                s = fmaf(qv[h][7], k1.w, s);
                s = row16_sum<USE_DPP>(s) * a.scale;
                if (!ok) s = -INFINITY;
                const float mn = fmaxf(m[h], s);
                const float corr = expf(m[h] - mn);
                const float p = expf(s - mn);
                l[h] = l[h] * corr + p;
                o[h][0] = o[h][0] * corr + p * v0.x; o[h][1] = o[h][1] * corr + p * v0.y;
                o[h][2] = o[h][2] * corr + p * v0.z; o[h][3] = o[h][3] * corr + p * v0.w;
                o[h][4] = o[h][4] * corr + p * v1.x; o[h][5] = o[h][5] * corr + p * v1.y;
                o[h][6] = o[h][6] * corr + p * v1.z; o[h][7] = o[h][7] * corr + p * v1.w;
                m[h] = mn;
            }
        }
    }

    // merge the 4 lane groups of the wave (same dims dc, disjoint key subsets)
#pragma unroll
    for (int h = 0; h < HPK; h++) {
        float mm = fmaxf(m[h], __shfl_xor(m[h], 16, 64));
        mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
        const float f = expf(m[h] - mm);
        float ll = l[h] * f;
        ll += __shfl_xor(ll, 16, 64);
        ll += __shfl_xor(ll, 32, 64);
#pragma unroll
        for (int d = 0; d < 8; d++) {
            float ov = o[h][d] * f;
            ov += __shfl_xor(ov, 16, 64);
            ov += __shfl_xor(ov, 32, 64);
            o[h][d] = ov;
        }
        m[h] = mm; l[h] = ll;
        if (ks == 0) {
#pragma unroll
            for (int d = 0; d < 8; d++) sm_o[wave][h][dc * 8 + d] = o[h][d];
            if (dc == 0) { sm_m[wave][h] = mm; sm_l[wave][h] = ll; }
        }
    }
    __syncthreads();
    // merge the 4 waves: thread -> (head, 2 dims)
    {
        const int h = tid >> 6;          // HPK == 4 -> one wave per head here
        const int d0 = (tid & 63) * 2;
        if (h < HPK) {
            float mm = sm_m[0][h];
#pragma unroll
            for (int w = 1; w < 4; w++) mm = fmaxf(mm, sm_m[w][h]);
            float ll = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float f = expf(sm_m[w][h] - mm);
                ll += sm_l[w][h] * f;
                o0 += sm_o[w][h][d0] * f;
                o1 += sm_o[w][h][d0 + 1] * f;
            }
            const int head = kvh * HPK + h;
            auto put = [&](float *p, float v) {
                *p = v;
            };
            if (nsplit == 1 && !a.force_partials) {
                const float inv = ll > 0.f ? 1.0f / ll : 0.f;
                float *op = a.out + (size_t)qi * a.ldo + head * HD + d0;
                put(op, o0 * inv); put(op + 1, o1 * inv);
            } else {
                const size_t pidx = ((size_t)qi * a.n_heads + head) * nsplit + split;
                put(a.part_o + pidx * HD + d0, o0);
                put(a.part_o + pidx * HD + d0 + 1, o1);
                if (d0 == 0) { put(a.part_ml + pidx * 2, mm); put(a.part_ml + pidx * 2 + 1, ll); }
            }
        }
    }
}

// Encoder attention for streaming-size chunks (n_q <= 32 queries, head_dim 64, no GQA): grid = (heads, 64-key slices of the
// window), 256 threads; partials (o, m, l) in the layout k_attn_combine<64> merges.  k_attn_enc_bf16 tiles 128 queries per
// workgroup and spent 17.5 us per layer on a 25-row chunk, most of it on padding; this is plain f32 FMA (vox_causal_attention,
// voxtral_kernels.c:412-482, up to summation order): lane = key for the scores (K row in registers, q broadcast from LDS),
// lane = dim for P.V.
// Device-coherent float store / load without fences: write-through (sc1) stores and L1-bypassing loads, as the decode step's
// granules use (vox_decfuse.h).  (A device-scope __threadfence() is the wrong tool on this part: its release writes back and its
// acquire invalidates the XCD's whole L2 - the few-rows encoder layer took 130 us instead of 68 with one in k_attn_small.)
__device__ __forceinline__ void st_dev(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_dev(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float as_dpp_sum(float v) {          // wave-wide sum: 4 DPP row steps + the 4 row sums via readlane
    v = row16_sum<true>(v);
    const int iv = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}
__device__ __forceinline__ float as_dpp_max(float v) {
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

// NW waves per workgroup, 32 / NW query rows per wave (NW = 8: the score and P.V phases are serial chains of ~650 / ~250 cycles per
// row; with 8 rows per wave they were 2.2 + 1.7 us of a 12.9 us launch).
template <bool USE_DPP, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_attn_small(const AttnArgs a, int key_lo) {
    constexpr int RPW = 32 / NW, NTH = 64 * NW;
    __shared__ __attribute__((aligned(16))) float qs[32][64];
    __shared__ float ks[64][65];
    __shared__ __attribute__((aligned(16))) float vs[64][64];
    __shared__ __attribute__((aligned(16))) float ps[32][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, z = blockIdx.y, nz = gridDim.y;
    const int n = a.n_q, t0 = key_lo + z * 64;
    for (int i = tid; i < 512; i += NTH) {
        const int r = i >> 4, c = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = *reinterpret_cast<const float4 *>(a.q + (size_t)r * a.ldq + h * 64 + c);
        *reinterpret_cast<float4 *>(&qs[r][c]) = v;
    }
    for (int i = tid; i < 1024; i += NTH) {
        const int key = i >> 4, c = (i & 15) * 4;
        const int pos = t0 + key;
        float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
        if (pos <= a.last_key) {
            if (pos >= a.posB0) {
                const size_t off = (size_t)(pos - a.posB0) * a.ldB + h * 64 + c;
                kk = *reinterpret_cast<const float4 *>(a.kB + off); vv = *reinterpret_cast<const float4 *>(a.vB + off);
            } else {
                const size_t off = (size_t)(pos % a.capA) * a.ldA + h * 64 + c;
                kk = *reinterpret_cast<const float4 *>(a.kA + off); vv = *reinterpret_cast<const float4 *>(a.vA + off);
            }
        }
        ks[key][c] = kk.x; ks[key][c + 1] = kk.y; ks[key][c + 2] = kk.z; ks[key][c + 3] = kk.w;
        *reinterpret_cast<float4 *>(&vs[key][c]) = vv;
    }
    __syncthreads();
    {   // scores and the slice's softmax statistics: wave = RPW query rows, lane = key
        float kr[64];
#pragma unroll
        for (int d = 0; d < 64; d++) kr[d] = ks[lane][d];
        const int t = t0 + lane;
        // one row at a time (unrolling the 8 rows makes the compiler hoist all 512 q values into registers: 256 VGPRs + spills,
        // 23 us); the reductions are DPP row steps + readlane, not ds_bpermute butterflies
#pragma unroll 1
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r, P = a.qpos0 + row;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                const float4 q4 = *reinterpret_cast<const float4 *>(&qs[row][d]);
                s = fmaf(q4.x, kr[d], s); s = fmaf(q4.y, kr[d + 1], s); s = fmaf(q4.z, kr[d + 2], s); s = fmaf(q4.w, kr[d + 3], s);
            }
            const bool ok = row < n && t <= a.last_key && t <= P && t >= P - a.window + 1;
            s = ok ? s * a.scale : -1e30f;
            const float m = USE_DPP ? as_dpp_max(s) : wave_max(s);
            const float pe = (ok && m > -1e29f) ? expf(s - m) : 0.f;
            const float l = USE_DPP ? as_dpp_sum(pe) : wave_sum(pe);
            ps[row][lane] = pe;
            if (lane == 0 && row < n) {
                float *ml = a.part_ml + (((size_t)row * a.n_heads + h) * nz + z) * 2;
                st_dev(ml, m > -1e29f ? m : -1e30f); st_dev(ml + 1, l);
            }
        }
    }
    __syncthreads();
    {   // P.V: wave = RPW query rows, lane = dim
        float acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[r] = 0.f;
#pragma unroll 4
        for (int k4 = 0; k4 < 64; k4 += 4) {
            const float v0 = vs[k4][lane], v1 = vs[k4 + 1][lane], v2 = vs[k4 + 2][lane], v3 = vs[k4 + 3][lane];
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const float4 p4 = *reinterpret_cast<const float4 *>(&ps[wave * RPW + r][k4]);
                acc[r] = fmaf(p4.x, v0, acc[r]); acc[r] = fmaf(p4.y, v1, acc[r]); acc[r] = fmaf(p4.z, v2, acc[r]); acc[r] = fmaf(p4.w, v3, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = wave * RPW + r;
            if (row < n) st_dev(a.part_o + (((size_t)row * a.n_heads + h) * nz + z) * 64 + lane, acc[r]);
        }
    }
    // Memory-model note (advisor, round 5): the publication below is relaxed agent-scope stores + a relaxed fetch_add + relaxed loads, with
    // no release / acquire pair - formally a data race under the HIP memory model.  What makes it correct on gfx942 / gfx950 is the hardware
    // behaviour this engine relies on everywhere (vox_decfuse.h, vox_encstack.h; MI355X guide, "drained sc1" hand-off): agent-scope (sc1)
    // stores are written through and the s_waitcnt vmcnt(0) + barrier in front of the fetch_add means they have been acknowledged by
    // memory before the counter moves; the reader's sc1 loads bypass its L1.  The engine is gfx950-only (vox_hip_engine_create refuses
    // anything else), and vox_hip_reset_encoder clears the counters, so an aborted launch cannot leave a head unmerged for ever.
    // ---- merge of the head's key slices by whichever of its workgroups arrives last (round 5: this was a launch of its own,
    // k_attn_combine: ~5 us of launch floor per encoder layer for 200 KB of partials).  The partials are written through (sc1);
    // a workgroup counts itself in only when all of its stores have been acknowledged (vmcnt(0), barrier), and the last arriver
    // reads the partials with L1-bypassing loads.  The arithmetic and its order are k_attn_combine's.
    if (!a.arrive) return;
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(a.arrive + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == (unsigned)nz - 1u;
        if (s_last) __hip_atomic_store(a.arrive + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nobody else touches the counter until the next launch)
    }
    __syncthreads();
    if (!s_last) return;
    for (int row = wave; row < n; row += NW) {
        const size_t base = ((size_t)row * a.n_heads + h) * nz;
        float mm = -1e30f, ll = 0.f, ov = 0.f;
        for (int s0 = 0; s0 < nz; s0 += 16) {      // batches of 16 slices in flight (a 750-position window is 12 - 13 slices)
            float2 ml[16]; float o[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int sc = min(s0 + u, nz - 1);
                ml[u].x = ld_dev(a.part_ml + (base + sc) * 2); ml[u].y = ld_dev(a.part_ml + (base + sc) * 2 + 1);
                o[u] = ld_dev(a.part_o + (base + sc) * 64 + lane);
            }
            float mn = mm;
#pragma unroll
            for (int u = 0; u < 16; u++) if (s0 + u < nz) mn = fmaxf(mn, ml[u].x);
            const float c0 = expf(mm - mn);
            ll *= c0; ov *= c0; mm = mn;
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (s0 + u < nz) {
                    const float f = expf(ml[u].x - mm);
                    ll += ml[u].y * f;
                    ov += o[u] * f;
                }
        }
        a.out[(size_t)row * a.ldo + h * 64 + lane] = ll > 0.f ? ov * (1.0f / ll) : 0.f;
    }
}

// Merge split-K partials.  grid = (n_heads, n_q), block = HD threads.
// The partials come from other CUs' previous kernel (every load a trip to L2 / memory): all (max, sum) pairs and a batch of
// 16 output values are requested before anything is computed on them; the arithmetic and its order are unchanged.
template <int HD>
__global__ __launch_bounds__(HD) void k_attn_combine(float *out, int ldo, const float *part_o,
                                                     const float *part_ml, int n_heads, int nsplit) {
    const int head = blockIdx.x, qi = blockIdx.y, d = threadIdx.x;
    const size_t base = ((size_t)qi * n_heads + head) * nsplit;
    float mm = -1e30f, ll = 0.f, ov = 0.f;
    if (nsplit <= 16) {
        float2 ml[16]; float o[16];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int sc = min(s, nsplit - 1);
            ml[s] = *reinterpret_cast<const float2 *>(part_ml + (base + sc) * 2);
            o[s] = part_o[(base + sc) * HD + d];
        }
#pragma unroll
        for (int s = 0; s < 16; s++) if (s < nsplit) mm = fmaxf(mm, ml[s].x);
#pragma unroll
        for (int s = 0; s < 16; s++)
            if (s < nsplit) {
                const float f = expf(ml[s].x - mm);
                ll += ml[s].y * f;
                ov += o[s] * f;
            }
    } else {
        for (int s = 0; s < nsplit; s++) mm = fmaxf(mm, part_ml[(base + s) * 2]);
        for (int s = 0; s < nsplit; s++) {
            const float f = expf(part_ml[(base + s) * 2] - mm);
            ll += part_ml[(base + s) * 2 + 1] * f;
            ov += part_o[(base + s) * HD + d] * f;
        }
    }
    out[(size_t)qi * ldo + head * HD + d] = ll > 0.f ? ov * (1.0f / ll) : 0.f;
}

}  // namespace vox
