// vox_persist.h — persistent decode kernel for the Voxtral-4B decoder shapes (gfx950).
//
// Why: with one launch per GEMV (vox_gemv.h) a decoder token is ~130 launches of 5-25 us.
// rocprofv3 shows each of them streaming at ~6.3 TB/s *while it streams*, but paying ~5-6 us
// of start-up / prologue / reduction / drain during which HBM idles: ~40 % of the step.
// This kernel runs the whole greedy loop (all steps of a vox_hip_decoder_run batch: 26
// layers, logits, argmax per step) in ONE cooperative launch of 256 workgroups (one per CU)
// and replaces launch boundaries by software grid barriers.  The point is not the barrier
// (it costs about as much as a launch boundary) but what can be done across it: before a
// workgroup arrives at a barrier its streaming waves have already issued the weight loads of
// their NEXT phase (24-36 x 16 B per lane, ~37 MB chip-wide), so HBM keeps streaming while
// the grid synchronises, the new activation vector is staged and results are reduced.
//
// Roles inside a workgroup (256 threads = one wave per SIMD, so each wave may use the full
// 512-entry register file: the weight pieces in flight live in VGPRs):
//   waves 0-2  "streaming": issue weight loads (non-temporal, all up front), FMA them against
//              the activation vector in LDS, wave-reduce, drop results into an LDS outbox.
//              They never read activations from global memory and never store to it — loads
//              return in order, so any such access would queue behind the prefetched weights.
//   wave 3     "control": has no prefetch in flight.  Stages the activation vector of each
//              phase (RMSNorm, embedding, attention merge) from global memory into LDS, writes
//              the outbox to global memory (RoPE, residual adds, logits, partials), and runs
//              the grid barrier: stores -> release fence -> vmcnt(0) -> relaxed arrive ->
//              relaxed poll (bounded) -> acquire fence (cdna_hip_programming.md §6 G16).
//
// Work split (NB = 256 blocks x 3 streaming waves = 768 waves):
//   P1 qkv    6144 x 3072 : 8 rows / wave  (2 half-batches of 4 rows x 6 pieces = 48 loads/lane)
//   P2 attn   blocks < 8*nsplit run one (kv head, key slice) each
//   P3 wo     3072 x 4096 : 4 rows / wave  (2 half-batches of 4 rows x 4 pieces)
//   P4 w1;w3  2 x 9216 x 3072 : 12 row pairs / wave (6 half-batches, ring of two)
//   P5 w2     3072 x 9216 : 4 rows / wave  (3 half-batches of 4 rows x 6 pieces)
//   PL logits 131072 x 3072 : 512 rows / block in 64 groups of 8, dealt round-robin to the waves
// A "piece" is the 16 bytes a lane loads (64 lanes x 16 B = 1 KiB contiguous per row).
//
// Barriers are two-level (8 group counters + 1 top counter; the counters are zeroed by the
// host before every launch) and every spin is bounded: on a timeout an error word is set,
// all later barriers fall through, and the host discards the batch and re-runs it on the
// multi-launch path.  Arithmetic is that of the multi-launch kernels (same per-row piece
// order, same wave reduction, same epilogues), so the two paths produce identical bits.
#pragma once
#include "vox_common.h"

namespace vox {

struct PersistLayer {
    const uint16_t *wqkv, *wo, *w13, *w2;
    const float *n1, *n2, *ada;
    float *kring, *vring;
};

struct PersistArgs {
    const PersistLayer *layers;     // [n_layers] in device memory
    int n_layers;
    const uint16_t *tok_emb;        // [131072][3072]
    const float *final_norm, *inv_freq, *adapter;
    DecState *st;
    float *x, *q, *h, *part_o, *part_ml, *logits, *blk_val;
    int *blk_idx, *tokens_out;
    unsigned *bar;                  // [0..7] group counters, [8] top counter, [9] error word
    int n_steps, eos, kv_cap, window;
    float eps;
    long long logits_stride;        // 0: every step overwrites `logits`; else step i -> logits + i*stride
    unsigned long long spin_limit;  // wall_clock64 ticks (100 MHz) a barrier may wait
    unsigned long long *trace;      // optional [2][4096] timestamps of blocks 0 and 131 (VOX_HIP_PERSIST_TRACE)
};

namespace pk {
constexpr int NB = 256;             // workgroups (= CUs of an MI355X)
constexpr int SW = 3;               // streaming waves per workgroup
constexpr int D = 3072, DQ = 4096, DKV = 1024, DH = 9216, HD = 128, VOCAB = 131072;
constexpr int THREADS = 64 * (SW + 1);
constexpr int LDS_FLOATS = DH + 16 + HD + 256 + 16 + 16 + 4 * 4 * HD + 512 + 512 + 16;
static_assert(THREADS == 256, "one wave per SIMD");

template <int NR, int NC>
struct HB { uint4 w[NR][NC]; };     // NR rows x NC pieces per lane

// Weight loads go through a buffer descriptor held in SGPRs: address = base(SGPR) + row/piece
// offset (SGPR, wave-uniform) + lane*16 (the only VGPR).  With flat 64-bit addresses hipcc
// hoists one VGPR pair per (row, piece) out of the layer loop and spills hundreds of them.
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, 0x7fffffff, 0x00020000);
}
template <int NR, int NC>
__device__ __forceinline__ void hb_load(HB<NR, NC> &b, const uint16_t *W, int K, int row0, int piece0, int lane) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(W);
    const int voff = lane * 16;
    // pin the issue point: without this the scheduler hoists later batches above earlier dot
    // products (more "ILP") and the extra live weight registers spill to scratch
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int soff = ((row0 + r) * K + piece0 * 512) * 2;      // bytes; wave-uniform
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + c * 1024, /*nt*/ 2);
            b.w[r][c] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int NR, int NC>
__device__ __forceinline__ void hb_dot(const HB<NR, NC> &b, const float *xs, int piece0, int lane, float *acc) {
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float *xp = xs + ((piece0 + c) * 64 + lane) * 8;
        const float4 x0 = *reinterpret_cast<const float4 *>(xp);
        const float4 x1 = *reinterpret_cast<const float4 *>(xp + 4);
#pragma unroll
        for (int r = 0; r < NR; r++) acc[r] = dot8_bf16(b.w[r][c], x0, x1, acc[r]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Control wave: arrive at / wait for grid barrier number `epoch` (1-based since launch).
// Its own global stores (the phase results) precede this call in program order.
struct Trace {
    unsigned long long *p; int n;
    __device__ __forceinline__ void mark() { if (p && n < 4096) p[n++] = wall_clock64(); }
};

__device__ __forceinline__ void ctrl_barrier(const PersistArgs &a, unsigned epoch, int lane, Trace &tr) {
    if (lane == 0) {
        unsigned *grp = a.bar + (blockIdx.x & 7);
        unsigned *top = a.bar + 8;
        unsigned *err = a.bar + 9;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr.mark();                                                                  // released
        const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((old + 1u) % (NB / 8)) == 0u) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tr.mark();                                                                  // arrived
        const unsigned target = epoch * 8u;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > a.spin_limit) {
                    __hip_atomic_store(err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            }
        }
        tr.mark();                                                                  // everyone arrived
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        tr.mark();                                                                  // acquired
    }
}

// Control wave: src (global, n floats) -> xs (LDS), RMS-normalised with w and optional (1+ada).
__device__ __forceinline__ void ctrl_stage_rms(float *xs, const float *src, const float *w, const float *ada,
                                               float eps, int lane) {
    float ss = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        *reinterpret_cast<float4 *>(xs + i) = v;
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)D + eps);
    for (int i = lane * 4; i < D; i += 256) {
        float4 v = *reinterpret_cast<float4 *>(xs + i);
        const float4 g = *reinterpret_cast<const float4 *>(w + i);
        v.x = v.x * inv * g.x; v.y = v.y * inv * g.y; v.z = v.z * inv * g.z; v.w = v.w * inv * g.w;
        if (ada) {
            const float4 sc = *reinterpret_cast<const float4 *>(ada + i);
            v.x *= (1.0f + sc.x); v.y *= (1.0f + sc.y); v.z *= (1.0f + sc.z); v.w *= (1.0f + sc.w);
        }
        *reinterpret_cast<float4 *>(xs + i) = v;
    }
}

__device__ __forceinline__ int persist_split_keys(int kv_len) {     // nsplit <= 8 for windows <= 8192
    return kv_len <= 512 ? 64 : kv_len <= 1024 ? 128 : kv_len <= 2048 ? 256 : kv_len <= 4096 ? 512 : 1024;
}
}  // namespace pk

// LDS carve-up shared by the two roles
struct PersistLds {
    float *xs, *ropet, *scl, *at_m, *at_l, *at_o, *qs, *outbox, *ctl;
    __device__ explicit PersistLds(float *smem) {
        xs = smem;                        // [DH] activation vector of the current phase
        ropet = xs + pk::DH + 16;         // [HD] cos,sin pairs of this step
        scl = ropet + pk::HD;             // [32*8] attention merge scales
        at_m = scl + 256;                 // [SW waves][4 heads]
        at_l = at_m + 16;
        at_o = at_l + 16;                 // [SW][4][HD]
        qs = at_o + 4 * 4 * pk::HD;       // [4*HD] q of this block's head group (attention blocks)
        outbox = qs + 512;                // [512] phase results of the streaming waves
        ctl = outbox + 512;               // [16] token broadcast
    }
};

// The two roles execute the same sequence of __syncthreads(); each S<n> below is matched by
// the S<n> of the other role.  They are separate functions so that the streaming waves' weight
// registers are plain straight-line values (no role-dependent phis keeping them live everywhere).

// ------------------------------------------------------------------------------------------
// streaming waves (0 .. SW-1)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void persist_streaming(const PersistArgs &a, const PersistLds &m, int wave, int lane) {
    using namespace pk;
    const int blk = blockIdx.x;
    const int gw = blk * SW + wave;
    int pos = a.st->pos, token = a.st->token, stop = a.st->stop;
    (void)token;
    HB<4, 6> A46, B46;
    if (!stop && a.n_steps > 0) {
        hb_load(A46, a.layers[0].wqkv, D, gw * 8, 0, lane);
        hb_load(B46, a.layers[0].wqkv, D, gw * 8 + 4, 0, lane);
    }
    for (int step = 0; step < a.n_steps && !stop; step++) {
        const int kv_len = min(pos + 1, a.window);
        const int split_keys = persist_split_keys(kv_len);
        const int nsplit = (kv_len + split_keys - 1) / split_keys;
        const bool attn_blk = blk < 8 * nsplit;
        for (int l = 0; l < a.n_layers; l++) {
            const PersistLayer L = a.layers[l];
            const bool last_layer = (l + 1 == a.n_layers);
            HB<4, 4> A44, B44;
            // ---- P1 -------------------------------------------------------------------------
            __syncthreads();                                                       // S1: xs staged
            {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                hb_dot(A46, m.xs, 0, lane, acc);
                hb_dot(B46, m.xs, 0, lane, acc + 4);
#pragma unroll
                for (int r = 0; r < 8; r++) acc[r] = wave_sum(acc[r]);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 8; r++) m.outbox[wave * 8 + r] = acc[r];
                }
            }
            if (!attn_blk) {
                // everybody but the attention blocks asks for its share of Wo before the barrier
                hb_load(A44, L.wo, DQ, gw * 4, 0, lane);
                hb_load(B44, L.wo, DQ, gw * 4, 4, lane);
                __syncthreads();                                                   // S2: outbox ready
                __syncthreads();                                                   // S3: grid barrier 1 passed
            } else {
                __syncthreads();                                                   // S2
                __syncthreads();                                                   // S3
                // ---- P2: one (kv head, key slice) per attention block ------------------------
                const int kvh = blk & 7, split = blk >> 3;
                __syncthreads();                                                   // S4a: q staged
                const int ks = lane >> 4, dc = lane & 15;
                int lo = pos - a.window + 1; if (lo < 0) lo = 0;
                const int s_lo = lo + split * split_keys;
                int s_hi = s_lo + split_keys - 1; if (s_hi > pos) s_hi = pos;
                const int per_wave = (split_keys + SW - 1) / SW;
                const int w_lo = s_lo + wave * per_wave;
                int w_hi = w_lo + per_wave - 1; if (w_hi > s_hi) w_hi = s_hi;
                const float scale = 1.0f / sqrtf((float)HD);
                float qv[4][8], o[4][8], mx[4], lsum[4];
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const float4 t0 = *reinterpret_cast<const float4 *>(m.qs + h * HD + dc * 8);
                    const float4 t1 = *reinterpret_cast<const float4 *>(m.qs + h * HD + dc * 8 + 4);
                    qv[h][0] = t0.x; qv[h][1] = t0.y; qv[h][2] = t0.z; qv[h][3] = t0.w;
                    qv[h][4] = t1.x; qv[h][5] = t1.y; qv[h][6] = t1.z; qv[h][7] = t1.w;
                    mx[h] = -1e30f; lsum[h] = 0.f;
#pragma unroll
                    for (int d = 0; d < 8; d++) o[h][d] = 0.f;
                }
                constexpr int UNR = 4;
                for (int t = w_lo + ks; t <= w_hi; t += 4 * UNR) {
                    float4 kq0[UNR], kq1[UNR], vq0[UNR], vq1[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        int tt = t + 4 * u; if (tt > w_hi) tt = w_hi;
                        const size_t off = (size_t)(tt % a.kv_cap) * DKV + kvh * HD + dc * 8;
                        kq0[u] = *reinterpret_cast<const float4 *>(L.kring + off);
                        kq1[u] = *reinterpret_cast<const float4 *>(L.kring + off + 4);
                        vq0[u] = *reinterpret_cast<const float4 *>(L.vring + off);
                        vq1[u] = *reinterpret_cast<const float4 *>(L.vring + off + 4);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const bool ok = (t + 4 * u) <= w_hi;
                        const float4 k0 = kq0[u], k1 = kq1[u], v0 = vq0[u], v1 = vq1[u];
#pragma unroll
                        for (int h = 0; h < 4; h++) {
                            float s = qv[h][0] * k0.x;
                            s = fmaf(qv[h][1], k0.y, s); s = fmaf(qv[h][2], k0.z, s); s = fmaf(qv[h][3], k0.w, s);
                            s = fmaf(qv[h][4], k1.x, s); s = fmaf(qv[h][5], k1.y, s); s = fmaf(qv[h][6], k1.z, s);
                            s = fmaf(qv[h][7], k1.w, s);
                            s = row16_sum<true>(s) * scale;
                            if (!ok) s = -INFINITY;
                            const float mn = fmaxf(mx[h], s);
                            const float corr = expf(mx[h] - mn);
                            const float p = expf(s - mn);
                            lsum[h] = lsum[h] * corr + p;
                            o[h][0] = o[h][0] * corr + p * v0.x; o[h][1] = o[h][1] * corr + p * v0.y;
                            o[h][2] = o[h][2] * corr + p * v0.z; o[h][3] = o[h][3] * corr + p * v0.w;
                            o[h][4] = o[h][4] * corr + p * v1.x; o[h][5] = o[h][5] * corr + p * v1.y;
                            o[h][6] = o[h][6] * corr + p * v1.z; o[h][7] = o[h][7] * corr + p * v1.w;
                            mx[h] = mn;
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    float mm = fmaxf(mx[h], __shfl_xor(mx[h], 16, 64));
                    mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
                    const float f = expf(mx[h] - mm);
                    float ll = lsum[h] * f;
                    ll += __shfl_xor(ll, 16, 64);
                    ll += __shfl_xor(ll, 32, 64);
#pragma unroll
                    for (int d = 0; d < 8; d++) {
                        float ov = o[h][d] * f;
                        ov += __shfl_xor(ov, 16, 64);
                        ov += __shfl_xor(ov, 32, 64);
                        o[h][d] = ov;
                    }
                    if (ks == 0) {
#pragma unroll
                        for (int d = 0; d < 8; d++) m.at_o[(wave * 4 + h) * HD + dc * 8 + d] = o[h][d];
                        if (dc == 0) { m.at_m[wave * 4 + h] = mm; m.at_l[wave * 4 + h] = ll; }
                    }
                }
                // now this block's share of Wo
                hb_load(A44, L.wo, DQ, gw * 4, 0, lane);
                hb_load(B44, L.wo, DQ, gw * 4, 4, lane);
                __syncthreads();                                                   // S4b: slice results in LDS
            }
            __syncthreads();                                                       // S5: grid barrier 2 passed
            // ---- P3 -------------------------------------------------------------------------
            __syncthreads();                                                       // S6: merged attention staged
            {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                hb_dot(A44, m.xs, 0, lane, acc);
                hb_dot(B44, m.xs, 4, lane, acc);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = wave_sum(acc[r]);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) m.outbox[wave * 4 + r] = acc[r];
                }
            }
            hb_load(A46, L.w13, D, gw * 12, 0, lane);                             // w1 rows 0..3
            hb_load(B46, L.w13 + (size_t)DH * D, D, gw * 12, 0, lane);            // w3 rows 0..3
            __syncthreads();                                                       // S7
            __syncthreads();                                                       // S8: grid barrier 3 passed
            // ---- P4 -------------------------------------------------------------------------
            __syncthreads();                                                       // S9: xs staged
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float g[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
                hb_dot(A46, m.xs, 0, lane, g);
                if (j < 2) hb_load(A46, L.w13, D, gw * 12 + 4 * (j + 1), 0, lane);
                else hb_load(A46, L.w2, DH, gw * 4, 0, lane);                     // P5 pieces 0..5
                hb_dot(B46, m.xs, 0, lane, u);
                if (j < 2) hb_load(B46, L.w13 + (size_t)DH * D, D, gw * 12 + 4 * (j + 1), 0, lane);
                else hb_load(B46, L.w2, DH, gw * 4, 6, lane);                     // P5 pieces 6..11
#pragma unroll
                for (int r = 0; r < 4; r++) { g[r] = wave_sum(g[r]); u[r] = wave_sum(u[r]); }
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) m.outbox[wave * 12 + 4 * j + r] = silu(g[r]) * u[r];
                }
            }
            __syncthreads();                                                       // S10
            __syncthreads();                                                       // S11: grid barrier 4 passed
            // ---- P5 -------------------------------------------------------------------------
            __syncthreads();                                                       // S12: h staged
            {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                hb_dot(A46, m.xs, 0, lane, acc);
                hb_load(A46, L.w2, DH, gw * 4, 12, lane);                         // pieces 12..17
                hb_dot(B46, m.xs, 6, lane, acc);
                if (!last_layer) hb_load(B46, a.layers[l + 1].wqkv, D, gw * 8 + 4, 0, lane);   // next P1 rows 4..7
                else hb_load(B46, a.tok_emb, D, blk * 512 + wave * 8 + 4, 0, lane);            // PL group `wave`
                hb_dot(A46, m.xs, 12, lane, acc);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = wave_sum(acc[r]);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) m.outbox[wave * 4 + r] = acc[r];
                }
                if (!last_layer) hb_load(A46, a.layers[l + 1].wqkv, D, gw * 8, 0, lane);       // next P1 rows 0..3
                else hb_load(A46, a.tok_emb, D, blk * 512 + wave * 8, 0, lane);
            }
            __syncthreads();                                                       // S13
            __syncthreads();                                                       // S14: grid barrier 5 passed
        }
        // ---- PL ---------------------------------------------------------------------------------
        __syncthreads();                                                           // S15: xs staged
#pragma unroll 1
        for (int g = wave; g < 64; g += SW) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f}, acc2[4] = {0.f, 0.f, 0.f, 0.f};
            const int gn = g + SW;
            hb_dot(A46, m.xs, 0, lane, acc);
            if (gn < 64) hb_load(A46, a.tok_emb, D, blk * 512 + gn * 8, 0, lane);
            hb_dot(B46, m.xs, 0, lane, acc2);
            if (gn < 64) hb_load(B46, a.tok_emb, D, blk * 512 + gn * 8 + 4, 0, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) { acc[r] = wave_sum(acc[r]); acc2[r] = wave_sum(acc2[r]); }
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) { m.outbox[g * 8 + r] = acc[r]; m.outbox[g * 8 + 4 + r] = acc2[r]; }
            }
        }
        if (step + 1 < a.n_steps) {       // next step's first weights do not depend on the token
            hb_load(A46, a.layers[0].wqkv, D, gw * 8, 0, lane);
            hb_load(B46, a.layers[0].wqkv, D, gw * 8 + 4, 0, lane);
        }
        __syncthreads();                                                           // S16: logits in outbox
        __syncthreads();                                                           // S17: token in ctl
        token = reinterpret_cast<const int *>(m.ctl)[0];
        pos += 1;
        if (token == a.eos) stop = 1;
        __syncthreads();                                                           // S18
    }
}

// ------------------------------------------------------------------------------------------
// control wave (wave SW)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void persist_control(const PersistArgs &a, const PersistLds &m, int lane) {
    using namespace pk;
    const int blk = blockIdx.x;
    unsigned epoch = 0;
    Trace tr{nullptr, 0};
    if (a.trace && lane == 0 && (blk == 0 || blk == 131)) tr.p = a.trace + (blk == 0 ? 0 : 4096);
    int pos = a.st->pos, token = a.st->token, stop = a.st->stop;
    long long arow = a.st->adapter_row;
    for (int step = 0; step < a.n_steps && !stop; step++) {
        const int kv_len = min(pos + 1, a.window);
        const int split_keys = persist_split_keys(kv_len);
        const int nsplit = (kv_len + split_keys - 1) / split_keys;
        const bool attn_blk = blk < 8 * nsplit;
        float *logits = a.logits + (size_t)step * a.logits_stride;
        for (int l = 0; l < a.n_layers; l++) {
            const PersistLayer L = a.layers[l];
            // ---- P1: stage RMSNorm(x) (layer 0: build the step embedding first) ---------------
            if (l == 0) {
                float *raw = m.xs + DH - D;              // park the raw row in the tail of xs
                const float *ar = a.adapter + (size_t)arow * D;
                const uint16_t *er = a.tok_emb + (size_t)token * D;
                for (int i = lane * 4; i < D; i += 256) {
                    float4 v = *reinterpret_cast<const float4 *>(ar + i);
                    const uint2 eb = *reinterpret_cast<const uint2 *>(er + i);
                    v.x += bf16_lo(eb.x); v.y += bf16_hi(eb.x); v.z += bf16_lo(eb.y); v.w += bf16_hi(eb.y);
                    *reinterpret_cast<float4 *>(raw + i) = v;
                    if (blk == 0) *reinterpret_cast<float4 *>(a.x + i) = v;
                }
                const float ang = (float)pos * a.inv_freq[lane];          // HD/2 == 64 lanes
                m.ropet[2 * lane] = cosf(ang);
                m.ropet[2 * lane + 1] = sinf(ang);
                ctrl_stage_rms(m.xs, raw, L.n1, nullptr, a.eps, lane);
            } else {
                ctrl_stage_rms(m.xs, a.x, L.n1, nullptr, a.eps, lane);
            }
            tr.mark();
            __syncthreads();                                                       // S1
            __syncthreads();                                                       // S2
            tr.mark();
            if (lane < 12) {              // the block's 24 rows = 12 (even, odd) RoPE pairs
                const int row = blk * 24 + 2 * lane;
                float o0 = m.outbox[2 * lane], o1 = m.outbox[2 * lane + 1];
                if (row < DQ + DKV) {
                    const int d = (row % HD) >> 1;
                    const float c = m.ropet[2 * d], sn = m.ropet[2 * d + 1];
                    const float x0 = o0, x1 = o1;
                    o0 = x0 * c - x1 * sn;
                    o1 = x0 * sn + x1 * c;
                }
                const int slot = pos % a.kv_cap;
                float *dst;
                if (row < DQ) dst = a.q + row;
                else if (row < DQ + DKV) dst = L.kring + (size_t)slot * DKV + (row - DQ);
                else dst = L.vring + (size_t)slot * DKV + (row - DQ - DKV);
                dst[0] = o0;
                dst[1] = o1;
            }
            ctrl_barrier(a, ++epoch, lane, tr);
            __syncthreads();                                                       // S3
            // ---- P2 ---------------------------------------------------------------------------
            if (attn_blk) {
                const int kvh = blk & 7, split = blk >> 3;
                for (int i = lane * 4; i < 4 * HD; i += 256)
                    *reinterpret_cast<float4 *>(m.qs + i) = *reinterpret_cast<const float4 *>(a.q + kvh * 4 * HD + i);
                tr.mark();
                __syncthreads();                                                   // S4a
                __syncthreads();                                                   // S4b
                tr.mark();
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    float mm = m.at_m[h];
#pragma unroll
                    for (int w_ = 1; w_ < SW; w_++) mm = fmaxf(mm, m.at_m[w_ * 4 + h]);
                    float ll = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < SW; w_++) {
                        const float f = expf(m.at_m[w_ * 4 + h] - mm);
                        ll += m.at_l[w_ * 4 + h] * f;
                        o0 += m.at_o[(w_ * 4 + h) * HD + lane * 2] * f;
                        o1 += m.at_o[(w_ * 4 + h) * HD + lane * 2 + 1] * f;
                    }
                    const size_t pidx = (size_t)(kvh * 4 + h) * nsplit + split;
                    a.part_o[pidx * HD + lane * 2] = o0;
                    a.part_o[pidx * HD + lane * 2 + 1] = o1;
                    if (lane == 0) { a.part_ml[pidx * 2] = mm; a.part_ml[pidx * 2 + 1] = ll; }
                }
            }
            ctrl_barrier(a, ++epoch, lane, tr);
            __syncthreads();                                                       // S5
            // ---- P3: merge the attention partials into xs --------------------------------------
            if (lane < 32) {
                const int hh = lane;
                float mm = -1e30f;
                for (int s_ = 0; s_ < nsplit; s_++) mm = fmaxf(mm, a.part_ml[(hh * nsplit + s_) * 2]);
                float ll = 0.f;
                for (int s_ = 0; s_ < nsplit; s_++) ll += a.part_ml[(hh * nsplit + s_) * 2 + 1] * expf(a.part_ml[(hh * nsplit + s_) * 2] - mm);
                const float inv = ll > 0.f ? 1.0f / ll : 0.f;
                for (int s_ = 0; s_ < nsplit; s_++) m.scl[hh * 8 + s_] = expf(a.part_ml[(hh * nsplit + s_) * 2] - mm) * inv;
            }
            for (int i = lane * 4; i < DQ; i += 256) {
                const int hh = i / HD, d = i - hh * HD;
                float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s_ = 0; s_ < nsplit; s_++) {
                    const float f = m.scl[hh * 8 + s_];
                    const float4 ov = *reinterpret_cast<const float4 *>(a.part_o + ((size_t)hh * nsplit + s_) * HD + d);
                    acc4.x += ov.x * f; acc4.y += ov.y * f; acc4.z += ov.z * f; acc4.w += ov.w * f;
                }
                *reinterpret_cast<float4 *>(m.xs + i) = acc4;
            }
            tr.mark();
            __syncthreads();                                                       // S6
            __syncthreads();                                                       // S7
            tr.mark();
            if (lane < 12) a.x[blk * 12 + lane] = a.x[blk * 12 + lane] + m.outbox[lane];      // x += proj (voxtral_decoder.c:676)
            ctrl_barrier(a, ++epoch, lane, tr);
            __syncthreads();                                                       // S8
            // ---- P4 ---------------------------------------------------------------------------
            ctrl_stage_rms(m.xs, a.x, L.n2, L.ada, a.eps, lane);
            tr.mark();
            __syncthreads();                                                       // S9
            __syncthreads();                                                       // S10
            tr.mark();
            if (lane < 36) a.h[blk * 36 + lane] = m.outbox[lane];
            ctrl_barrier(a, ++epoch, lane, tr);
            __syncthreads();                                                       // S11
            // ---- P5 ---------------------------------------------------------------------------
            for (int i = lane * 4; i < DH; i += 256)
                *reinterpret_cast<float4 *>(m.xs + i) = *reinterpret_cast<const float4 *>(a.h + i);
            tr.mark();
            __syncthreads();                                                       // S12
            __syncthreads();                                                       // S13
            tr.mark();
            if (lane < 12) a.x[blk * 12 + lane] = a.x[blk * 12 + lane] + m.outbox[lane];      // x += ffn (voxtral_decoder.c:689)
            ctrl_barrier(a, ++epoch, lane, tr);
            __syncthreads();                                                       // S14
        }
        // ---- PL ---------------------------------------------------------------------------------
        ctrl_stage_rms(m.xs, a.x, a.final_norm, nullptr, a.eps, lane);
        __syncthreads();                                                           // S15
        __syncthreads();                                                           // S16
        float bv = -3.0e38f; int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = lane + 64 * k;
            const float v = m.outbox[idx];
            const int row = blk * 512 + idx;
            logits[row] = v;
            if (v > bv || (v == bv && row < bi)) { bv = v; bi = row; }          // strict '>' => lowest index wins
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { a.blk_val[blk] = bv; a.blk_idx[blk] = bi; }
        ctrl_barrier(a, ++epoch, lane, tr);
        // every block reduces the 256 partials itself (same order everywhere => same token)
        bv = -3.0e38f; bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < NB / 64; k++) {
            const float v = a.blk_val[lane + 64 * k];
            const int ix = a.blk_idx[lane + 64 * k];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        token = (bi < 0 || bi == 0x7fffffff) ? 0 : bi;      // no winner (all-NaN logits): index 0 like the reference's scan
        if (lane == 0) reinterpret_cast<int *>(m.ctl)[0] = token;
        __syncthreads();                                                           // S17
        pos += 1; arow += 1;
        if (token == a.eos) stop = 1;
        if (blk == 0 && lane == 0) {
            a.tokens_out[step] = token;
            a.st->n_out = step + 1; a.st->token = token; a.st->pos = pos; a.st->adapter_row = arow; a.st->stop = stop;
        }
        __syncthreads();                                                           // S18
    }
}

__global__ __launch_bounds__(pk::THREADS, 1) void k_decode_persist(const PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PersistLds m(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction; readfirstlane makes that provable (row offsets stay in SGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave == pk::SW) persist_control(a, m, lane);
    else persist_streaming(a, m, wave, lane);
}

}  // namespace vox
