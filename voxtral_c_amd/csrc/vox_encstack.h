// vox_encstack.h — the encoder transformer on a FEW rows (streaming chunks: 25 rows at -I 0.5) as ONE persistent launch per chunk
// (round 6).  Replaces the 8 launches per layer of encoder_rows_skinny (vox_skinny.h: 69.6 us per layer for 60.3 MB of weights
// = 0.108 of the HBM roofline; every launch has a ~4.9 us floor in a dependent chain and pulls 3 - 4x its weight bytes as
// activation fragments).  Reference: vox_encoder_forward_incremental, voxtral_encoder.c:452-636 (per layer: attention_norm ->
// wq/wk/wv + bias -> RoPE -> KV append -> sliding-window attention -> wo + bias + residual -> ffn_norm -> silu(w1 x) * (w3 x) ->
// w2 + bias + residual); precedent for "a whole chunk as one submission": voxtral_metal.m:2717.
//
// 256 workgroups x 512 threads (one per CU, all co-resident), a loop over the layers; seven phases per layer, every phase spread
// over all 256 workgroups, phases separated by an all-to-all hand-off through memory:
//   P1  q;k;v = (x g1) Wqkv^T * inv_rms + bias, RoPE; q -> qbuf, k / v -> the position-indexed rings            24 W rows per workgroup, K split over the 8 waves
//   P2  attention partials of (head, key slice): 32 heads x 8 slices (<= 98 keys), online softmax, f32 FMAs      = k_attn_small's arithmetic
//   P3  wo as K-split partials: workgroup (head, 160-column group) merges its head's 8 slices on the way in       K = 64 per workgroup
//   F3  x' = x + bo + sum of the 32 partials; per (row, 160-column block): sum of squares; (x' g2) as bf16 planes
//   P4  h = silu(gate) * up, gate / up = (x' g2) W1^T / W3^T * inv_rms                                         20 hidden units per workgroup
//   P5  w2 as K-split partials: (320-wide K range, 80-row group) per workgroup
//   F5  x = x' + b2 + sum of the 16 partials; sums of squares; (x g1[l + 1]) as planes
// The RMSNorm's 1 / rms is applied AFTER the product that consumes the normalised row (it is a per-row scalar and the product is
// linear): the finish phases then need no row-wide reduction of their own, i.e. no extra hand-off - they publish the per-block sums
// of squares next to the planes and the consumer's epilogue adds them up.  ((x g) W^T) inv instead of ((x inv) g) W^T: the same
// value up to f32 rounding of single terms.
//
// Hand-offs.  Payload: write-through (sc1) stores, L1-bypassing (sc1) loads - no fences (a device-scope fence writes back /
// invalidates the XCD's whole L2: DESIGN 8.6).  Completion: workgroup w publishes "I have finished phase g" in flags[w] once its
// stores are acknowledged (s_waitcnt vmcnt(0), barrier, one 4-byte sc1 store: the drained-sc1 form of the guide's hand-off table);
// a phase begins when wave 0 has seen the flags it depends on (one 16-byte sc1 load per lane = all 256 flags per poll), bounded
// by ACTIVE spin time (DfSpin, vox_common.h); a time-out flags the chunk and the host repeats it on the 8-launch path.  The flag
// values are epoch + phase index with a per-launch epoch, so nothing is ever cleared.
// Measured and not kept (profiles/NOTES.md, round 6): three polls in flight instead of one (the flag words become a hot spot: every hand-off
// and every body got slower, 62.3 -> 66.4 us per layer); the attention phase on packed f32 FMAs (v_pk_fma_f32 with transposed q / p tiles: the
// compiler serialises one LDS read + one dependent FMA pair, 9.6 -> 15.1 us).
// Every phase's WEIGHT fragments are requested before the wait for the previous phase (waves 1 - 7; wave 0 polls first: a poll queued
// behind 100 KB of weight loads would return when they land), so the weight stream runs under the hand-offs.
//
// Activation operand layout ("fragment-major planes"): the exact 3-term bf16 split (hi, mid, lo: vox_gemm.h split3) of an f32 row
// block [32 rows][K], stored so that one wave-wide 16-byte-per-lane load IS one MFMA operand: piece (k-step ks of 32, plane p, row
// tile u of 16) = 1 KiB, lane (kb, li) holds row 16 u + li, columns 32 ks + 8 kb .. + 7.  v_mfma_f32_16x16x32_bf16, activations as
// the A operand, 16 weight rows as the B operand: C[row 4 kb + r][weight row li] (as in vox_skinny.h).
#pragma once
#include "vox_common.h"
#include "vox_gemm.h"
#include "vox_attn.h"

namespace vox {

constexpr int ES_D = 1280, ES_QD = 2048, ES_N3 = 6144, ES_H = 5120, ES_HEADS = 32, ES_HD = 64;
constexpr int ES_WGS = 256, ES_THREADS = 512;
constexpr int ES_NSL = 8;                 // key slices per head
constexpr int ES_CB = 8;                  // column blocks of a row in the finish phases
constexpr int ES_CBW = ES_D / ES_CB;      // 160 columns
constexpr int ES_KS_D = ES_D / 32;        // 40 k-steps of 32
constexpr int ES_KS_H = ES_H / 32;        // 160
constexpr int ES_KG5 = 16, ES_NG5 = 16;   // P5: K groups (320 wide) x row groups (80 rows)
constexpr int ES_NG3 = 8;                 // P3: column groups (160 rows of Wo) per head
constexpr int ES_PHASES = 7;
constexpr int ES_TL_STRIDE = 32;          // timeline words per workgroup (VOX_HIP_ENC_TL)
constexpr int ES_LDS_BYTES = 96 * 1024;
constexpr int ES_P2_LD = 68, ES_P2_LDS = 132;     // attention phase: LDS row strides (floats) of the q / K / V tiles and of the score tile

struct EncStackLayer {
    const uint16_t *wqkv;                 // [6144][1280] q rows, k rows, v rows
    const uint16_t *wo;                   // [1280][2048]
    const uint16_t *w1, *w3;              // [5120][1280] each
    const uint16_t *w2;                   // [1280][5120]
    const float *bqkv, *bo, *b2;          // [6144] (zeros where the reference has no bias), [1280], [1280]
    const float *n1, *n2;                 // attention_norm, ffn_norm weights [1280]
    float *kring, *vring;                 // [ring_cap][2048]
};

struct EncStackArgs {
    const EncStackLayer *layers; int n_layers;
    int n, pos0, ring_cap, window;
    float eps, scale;
    const float *x_in;                    // [n][1280] the chunk's rows (conv stem output); not modified
    const float *rope_tab;                // [n][32][2] cos, sin of the chunk's positions
    float *xa, *xb;                       // [32][1280] residual stream (layer input / after the attention block); the stack's output is xa
    uint16_t *aplanes;                    // fragment-major planes of a [32][1280] block: F0 / F5 -> P1, F3 -> P4       [40][3][2][64][8]
    float *ssq;                           // [2][32][ES_CB] sums of squares per (row, column block): [0] from F0 / F5, [1] from F3
    float *qbuf;                          // [32][2048]
    float *part_o;                        // [32 heads][8 slices][32 rows][64]
    float *part_ml;                       // [32 heads][8 slices][32 rows][2]
    float *wo_part;                       // [32 heads][32 rows][1280]
    uint16_t *hplanes;                    // fragment-major planes of h [32][5120]                                       [160][3][2][64][8]
    float *w2_part;                       // [16][32][1280]
    unsigned *flags;                      // [256]
    unsigned epoch;                       // phase g (0-based, over the whole launch) publishes epoch + g + 1
    unsigned *err; unsigned long long spin_limit;
    unsigned long long *tl; int tl_layer; // VOX_HIP_ENC_TL: per-workgroup stamps of layer tl_layer's phases
};

typedef unsigned es_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned es_u32x2 __attribute__((ext_vector_type(2)));
// Device-coherent accesses: buffer loads / stores with sc1 (aux = 16), tracked by the compiler's waitcnt insertion.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t es_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ es_u32x4 es_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ es_u32x2 es_ld8(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16); }
__device__ __forceinline__ float es_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 16)); }
__device__ __forceinline__ void es_st16(__amdgpu_buffer_rsrc_t r, unsigned off, es_u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
__device__ __forceinline__ void es_st8(__amdgpu_buffer_rsrc_t r, unsigned off, es_u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 16); }
__device__ __forceinline__ void es_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 16); }
__device__ __forceinline__ es_u32x4 es_f4(float x, float y, float z, float w) {
    return es_u32x4{__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
}

// weights: plain (default cache policy) 16-byte buffer loads - scalar base, 32-bit per-lane byte offset, counted by vmcnt only (a pointer
// fetched from the layer table is a generic pointer to the compiler: flat loads, which also tick lgkmcnt)
__device__ __forceinline__ uint4 es_ldw(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const es_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// the layer's table entry through the scalar (constant) path: twelve 8-byte words
__device__ __forceinline__ void es_load_layer(const EncStackLayer *p, EncStackLayer &L) {
    typedef const unsigned long long __attribute__((address_space(4))) *cptr;
    const cptr q = (cptr)(unsigned long long)p;
    unsigned long long w[12];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = q[i];
    L.wqkv = (const uint16_t *)w[0]; L.wo = (const uint16_t *)w[1]; L.w1 = (const uint16_t *)w[2]; L.w3 = (const uint16_t *)w[3];
    L.w2 = (const uint16_t *)w[4]; L.bqkv = (const float *)w[5]; L.bo = (const float *)w[6]; L.b2 = (const float *)w[7];
    L.n1 = (const float *)w[8]; L.n2 = (const float *)w[9]; L.kring = (float *)w[10]; L.vring = (float *)w[11];
}
static_assert(sizeof(EncStackLayer) == 12 * 8, "es_load_layer reads twelve pointers");

// byte offset of the 16-byte piece (k-step, plane, row tile, lane) of a fragment-major plane buffer
__device__ __forceinline__ unsigned es_frag_off(int ks, int p, int u, int lane) { return ((((unsigned)ks * 3 + p) * 2 + u) * 64 + lane) * 16; }
// byte offset of element (row m, column k) of plane p
__device__ __forceinline__ unsigned es_elem_off(int m, int k, int p) {
    return es_frag_off(k >> 5, p, m >> 4, ((k & 31) >> 3) * 16 + (m & 15)) + (k & 7) * 2;
}
// four consecutive columns (k0 % 4 == 0) of one row as the three planes' 8-byte pieces
__device__ __forceinline__ void es_split4(float x0, float x1, float x2, float x3, es_u32x2 &ph, es_u32x2 &pm, es_u32x2 &pl) {
    uint32_t h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(x0, h0, m0, l0); split3(x1, h1, m1, l1); split3(x2, h2, m2, l2); split3(x3, h3, m3, l3);
    ph = es_u32x2{(h0 >> 16) | h1, (h2 >> 16) | h3};
    pm = es_u32x2{(m0 >> 16) | m1, (m2 >> 16) | m3};
    pl = es_u32x2{(l0 >> 16) | (l1 & 0xffff0000u), (l2 >> 16) | (l3 & 0xffff0000u)};
}

// The thread / block index as values the compiler cannot see through (as df_tid in vox_decfuse.h): the phases run inside the loop over the
// layers, and with a plain threadIdx.x every address they derive from it is loop-invariant, hoisted and kept live across the loop -
// hundreds of spilled registers.  Laundered at the top of every phase, the arithmetic stays where it is written.
__device__ __forceinline__ int es_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int es_bid() { int b = blockIdx.x; asm volatile("" : "+s"(b)); return b; }

// ---- hand-off completion ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void es_arrive(const EncStackArgs &a, unsigned value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every store of this wave is acknowledged (write-through: it is in memory)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(a.flags + blockIdx.x, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wave 0 polls the flags of the workgroups this phase depends on until each is at or past `target` (es_poll: one 16-byte load per
// lane = all 256 flags per poll; `need(f)` says whether workgroup f is a producer of this workgroup's inputs); the other waves go
// straight on to request their weight fragments and meet wave 0 at the barrier (es_join).  Wave 0 requests its own fragments AFTER
// the poll: a poll queued behind ~10 KB of weight loads returns when they have landed.  es_join: false = the chunk is flagged (a
// time-out here or anywhere else) and the caller leaves the kernel.
// Dependencies narrower than "everybody" shorten the tails (a consumer starts when ITS producers are done); what keeps buffer reuse
// safe is that the waits in front of P4 and P1 stay global: every buffer is rewritten only behind a global wait that follows its
// last reader's phase (and the narrow ones are transitively complete: e.g. F3 needs all 32 heads' P3, each needs its head's P2, each
// needs its head's P1 - all 256 workgroups of P1).
struct EsNeedAll { __device__ __forceinline__ bool operator()(int) const { return true; } };
struct EsNeedNone { __device__ __forceinline__ bool operator()(int) const { return false; } };
struct EsNeedHead { int xcd, h4; __device__ __forceinline__ bool operator()(int f) const { return (f & 7) == xcd && ((f >> 3) & 3) == h4; } };     // the 8 workgroups of a head
struct EsNeedShift { int sh, v; __device__ __forceinline__ bool operator()(int f) const { return (f >> sh) == v; } };                             // workgroups f with f >> sh == v
template <typename Need>
__device__ __forceinline__ void es_poll(const EncStackArgs &a, unsigned target, int *s_ok, const Need need) {
    if (threadIdx.x < 64) {
        const __amdgpu_buffer_rsrc_t fr = es_rsrc(a.flags);
        const int f0 = threadIdx.x * 4;
        const bool n0 = need(f0), n1 = need(f0 + 1), n2 = need(f0 + 2), n3 = need(f0 + 3);
        const bool any = n0 || n1 || n2 || n3;         // a lane none of whose four flags matters does not load (narrow waits touch 1 - 8 lines, not the whole KB)
        int res = 1;
        DfSpin sp;
        for (unsigned it = 0;; it++) {
            es_u32x4 f = es_u32x4{target, target, target, target};
            if (any) f = es_ld16(fr, threadIdx.x * 16);
            const bool ok = (!n0 || (int)(f.x - target) >= 0) && (!n1 || (int)(f.y - target) >= 0) && (!n2 || (int)(f.z - target) >= 0) && (!n3 || (int)(f.w - target) >= 0);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            if (it == 0) df_spin_begin(sp);
            else if (df_spin_expired(sp, a.err, a.spin_limit, 11u, target)) { res = 0; break; }
            if ((it & 7u) == 7u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { res = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) *s_ok = res;
    }
}
__device__ __forceinline__ bool es_join(const int *s_ok) {
    __syncthreads();
    return *s_ok != 0;
}
template <typename Need>
__device__ __forceinline__ bool es_wait(const EncStackArgs &a, unsigned target, int *s_ok, const Need need) {
    es_poll(a, target, s_ok, need);
    return es_join(s_ok);
}

// 1 / rms of the chunk's rows from the per-block sums of squares (ssq[m][ES_CB], published by a finish phase) -> s_inv[32]
__device__ __forceinline__ void es_load_inv(const float *ssq, int n, float eps, float *s_inv) {
    if (threadIdx.x < 32) {
        const __amdgpu_buffer_rsrc_t r = es_rsrc(ssq);
        const int m = threadIdx.x;
        float tot = 0.f;
        if (m < n) {
            const es_u32x4 a0 = es_ld16(r, (unsigned)m * ES_CB * 4), a1 = es_ld16(r, (unsigned)m * ES_CB * 4 + 16);
            tot = __uint_as_float(a0.x); tot += __uint_as_float(a0.y); tot += __uint_as_float(a0.z); tot += __uint_as_float(a0.w);
            tot += __uint_as_float(a1.x); tot += __uint_as_float(a1.y); tot += __uint_as_float(a1.z); tot += __uint_as_float(a1.w);
        }
        s_inv[m] = 1.0f / sqrtf(tot / (float)ES_D + eps);
    }
}

// ---- finish phase: dst[m] = src[m] + bias + sum_s part[s][m] over one (row, 160-column block); sum of squares; (dst g) as planes ----------
// One workgroup per (row m, column block cb): task = m * ES_CB + cb < n * ES_CB.  Thread t: column group c4 = t % 40 (4 columns),
// partial subset sg = t / 40 (12 subsets; threads 480 .. 511 idle); subsets are added in order through LDS.
__device__ __forceinline__ void es_finish(const EncStackArgs &a, int task, const float *src, float *dst, const float *part, int nsplit,
                                          size_t part_stride /* floats between partials */, const float *bias, const float *norm_w,
                                          float *ssq, bool planes, unsigned char *lds) {
    const int tid = es_tid();
    const int m = task / ES_CB, cb = task - m * ES_CB;
    float4 *red = reinterpret_cast<float4 *>(lds);                       // [12][40]
    float *wsum = reinterpret_cast<float *>(lds + 12 * 40 * 16);         // [1]
    const int c4 = tid % 40, sg = tid / 40;
    const int col = cb * ES_CBW + c4 * 4;
    if (sg < 12) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nsplit > 0) {
            const __amdgpu_buffer_rsrc_t pr = es_rsrc(part);
            es_u32x4 v[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int sp = min(sg + 12 * j, nsplit - 1);
                v[j] = es_ld16(pr, (unsigned)(((size_t)sp * part_stride + (size_t)m * ES_D + col) * 4));
            }
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (sg + 12 * j < nsplit) {
                    s.x += __uint_as_float(v[j].x); s.y += __uint_as_float(v[j].y); s.z += __uint_as_float(v[j].z); s.w += __uint_as_float(v[j].w);
                }
        }
        red[sg * 40 + c4] = s;
    }
    __syncthreads();
    if (tid < 64) {
        float ss = 0.f;
        if (tid < 40) {
            float4 s = red[c4];
#pragma unroll
            for (int g = 1; g < 12; g++) { const float4 t = red[g * 40 + c4]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            if (bias) { const float4 b = *reinterpret_cast<const float4 *>(bias + col); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
            float4 x;
            if (nsplit > 0) {
                const es_u32x4 xv = es_ld16(es_rsrc(src), (unsigned)(((size_t)m * ES_D + col) * 4));
                x = make_float4(__uint_as_float(xv.x), __uint_as_float(xv.y), __uint_as_float(xv.z), __uint_as_float(xv.w));
            } else {
                x = *reinterpret_cast<const float4 *>(src + (size_t)m * ES_D + col);          // the chunk's input rows (written before the launch)
            }
            x.x += s.x; x.y += s.y; x.z += s.z; x.w += s.w;                                   // x + (proj + bias): the reference's order
            es_st16(es_rsrc(dst), (unsigned)(((size_t)m * ES_D + col) * 4), es_f4(x.x, x.y, x.z, x.w));
            ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            if (planes) {
                const float4 g = *reinterpret_cast<const float4 *>(norm_w + col);
                es_u32x2 ph, pm, pl;
                es_split4(x.x * g.x, x.y * g.y, x.z * g.z, x.w * g.w, ph, pm, pl);
                const __amdgpu_buffer_rsrc_t ar = es_rsrc(a.aplanes);
                es_st8(ar, es_elem_off(m, col, 0), ph); es_st8(ar, es_elem_off(m, col, 1), pm); es_st8(ar, es_elem_off(m, col, 2), pl);
            }
        }
        ss = wave_sum(ss);
        if (tid == 0) es_st4(es_rsrc(ssq), (unsigned)((m * ES_CB + cb) * 4), ss);
    }
    (void)wsum;
}

// (the matrix pipe's results are read by VALU / LDS instructions right behind conditionally executed MFMA blocks: DESIGN 6, "an MFMA hazard")
#define ES_MFMA_SETTLE() asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory")
#define ES_MARK(k) do { if (TL && tlon) es_stamp[k] = wall_clock64(); } while (0)

// MU = 16-row activation tiles in use (1: n <= 16, 2: n <= 32); TL = the instrumented build (VOX_HIP_ENC_TL)
template <int MU, bool TL>
__global__ __launch_bounds__(ES_THREADS) void k_enc_stack(const EncStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char es_lds[];
    __shared__ int s_ok;
    __shared__ float s_inv[32];
    const int n = a.n;
#define ES_IDS() const int tid = es_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, kb = lane >> 4, \
                           w = es_bid(), xcd = w & 7, widx = w >> 3; (void)li; (void)kb; (void)xcd; (void)widx; (void)wv; (void)lane
    unsigned long long es_stamp[2 * ES_PHASES + 1];
    const unsigned long long tl0 = TL ? wall_clock64() : 0ull;
    unsigned g = 0;                                   // phases finished so far
    // (xcd, widx) = the split of the workgroup id: workgroups w, w + 8, .. run on the same XCD (observed placement; speed only)

    // ---- F0: the chunk's rows as layer 0's un-normalised planes + sums of squares ---------------------------------------------------
    {
        ES_IDS();
        if (w < n * ES_CB)
            es_finish(a, w, a.x_in, a.xa, nullptr, 0, 0, nullptr, a.n_layers > 0 ? a.layers[0].n1 : nullptr, a.ssq, a.n_layers > 0, es_lds);
        es_arrive(a, a.epoch + (++g));
    }

    for (int l = 0; l < a.n_layers; l++) {
        EncStackLayer L; es_load_layer(a.layers + l, L);
        const bool tlon = TL && a.tl && l == a.tl_layer;
        const __amdgpu_buffer_rsrc_t apl = es_rsrc(a.aplanes);

        // =========================== P1: q;k;v ===============================================================================
        // workgroup (head, j): columns 8 j .. 8 j + 7 of the head's q, of its k and of its v (24 W rows; the 8 workgroups of a head share an
        // XCD and are all the attention phase of that head waits for).  W tile 0 = the q rows + the k rows, tile 1 = the v rows (+ 8 idle lanes).
        {
            ES_IDS();
            const int head = xcd * 4 + (widx & 3), c0 = head * 64 + 8 * (widx >> 2);
            uint4 wq[2][5];
            const __amdgpu_buffer_rsrc_t wr = es_rsrc(L.wqkv);
            auto issue = [&]() {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int row = t == 0 ? (li < 8 ? c0 + li : ES_QD + c0 + li - 8) : 2 * ES_QD + c0 + min(li, 7);
                    const unsigned wo_ = (unsigned)((row * ES_D + (5 * wv) * 32 + kb * 8) * 2);
#pragma unroll
                    for (int j = 0; j < 5; j++) wq[t][j] = es_ldw(wr, wo_ + j * 64);
                }
            };
            es_poll(a, a.epoch + g, &s_ok, EsNeedAll());            // (wave 0 only)
            issue();
            if (!es_join(&s_ok)) return;
            ES_MARK(0);
            es_load_inv(a.ssq, n, a.eps, s_inv);
            f32x4 acc[2][MU];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < MU; u++) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            es_u32x4 af[5][MU][3];                        // the wave's whole K slice (30 KiB per wave in flight: these loads are latency-bound)
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int p = 0; p < 3; p++) af[j][u][p] = es_ld16(apl, es_frag_off(5 * wv + j, p, u, lane));
#pragma unroll
            for (int j = 0; j < 5; j++) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    union { uint4 u4; bf16x8_t v; } fb; fb.u4 = wq[t][j];
#pragma unroll
                    for (int p = 2; p >= 0; p--)
#pragma unroll
                        for (int u = 0; u < MU; u++) {
                            union { es_u32x4 u4; bf16x8_t v; } fa; fa.u4 = af[j][u][p];
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[t][u], 0, 0, 0);
                        }
                }
            }
            // add the 8 K-splitters in wave order: red[wave][row 32][col 32 (24 used: q 0 .. 7, k 8 .. 15, v 16 .. 23)], row stride 33
            ES_MFMA_SETTLE();
            float *red = reinterpret_cast<float *>(es_lds);
            float *outt = red + 8 * 32 * 33;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[(wv * 32 + 16 * u + 4 * kb + r) * 33 + 16 * t + li] = acc[t][u][r];
            __syncthreads();
            for (int e = tid; e < 16 * MU * 24; e += ES_THREADS) {
                const int m = e / 24, c = e - m * 24;
                float v = red[m * 33 + c];
#pragma unroll
                for (int k = 1; k < 8; k++) v += red[(k * 32 + m) * 33 + c];
                outt[m * 33 + c] = v;
            }
            __syncthreads();
            // epilogue: thread = (row m, 4 columns of q, k or v): * inv + bias, RoPE pairs (q, k), destinations
            if (tid < 32 * 6) {
                const int m = tid / 6, c = (tid - m * 6) * 4;
                if (m < n) {
                    const int which = c >> 3, col = c0 + (c & 7);               // 0 q, 1 k, 2 v; column inside the [2048] block
                    const float inv = s_inv[m];
                    const float4 b = *reinterpret_cast<const float4 *>(L.bqkv + which * ES_QD + col);
                    float v0 = outt[m * 33 + c] * inv + b.x, v1 = outt[m * 33 + c + 1] * inv + b.y;
                    float v2 = outt[m * 33 + c + 2] * inv + b.z, v3 = outt[m * 33 + c + 3] * inv + b.w;
                    if (which < 2) {
                        const int d = (col & 63) >> 1;
                        const float4 cs = *reinterpret_cast<const float4 *>(a.rope_tab + ((size_t)m * 32 + d) * 2);      // cos d, sin d, cos d+1, sin d+1
                        const float q0 = v0 * cs.x - v1 * cs.y, q1 = v0 * cs.y + v1 * cs.x;
                        const float q2 = v2 * cs.z - v3 * cs.w, q3 = v2 * cs.w + v3 * cs.z;
                        v0 = q0; v1 = q1; v2 = q2; v3 = q3;
                    }
                    const es_u32x4 ov = es_f4(v0, v1, v2, v3);
                    if (which == 0) es_st16(es_rsrc(a.qbuf), (unsigned)(((size_t)m * ES_QD + col) * 4), ov);
                    else {
                        const int slot = (a.pos0 + m) % a.ring_cap;
                        es_st16(es_rsrc(which == 1 ? L.kring : L.vring), (unsigned)(((size_t)slot * ES_QD + col) * 4), ov);
                    }
                }
            }
            ES_MARK(1);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== P2: attention partials of (head, key slice) ============================================
        // vox_causal_attention's arithmetic (voxtral_kernels.c:412-482, up to summation order) on a slice of up to 128 keys, resident in LDS
        // as f32; both products on the f32 matrix pipe (round 6: the FMA form - lane = key for the scores, lane = dim for P.V - was 9.6 us of
        // VALU issue at two waves per SIMD), the softmax over the whole slice in between (k_attn_small merges 64-key tiles instead).  The
        // K / V rows of OLD positions do not depend on this layer's P1: they are requested before the wait; rows of this chunk's own
        // positions (the last slice or two) are fetched behind it.
        {
            ES_IDS();
            const int head = xcd * 4 + (widx & 3);
            const int slice = widx >> 2;
            const int lo = max(0, a.pos0 - a.window + 1), hi = a.pos0 + n - 1;
            const int nkeys = hi - lo + 1, KS = (nkeys + ES_NSL - 1) / ES_NSL;
            const int t0 = lo + slice * KS, t1 = min(t0 + KS - 1, hi);        // this slice's keys [t0, t1] (empty if t0 > t1)
            // LDS (floats): q [32][68] | K [128][68] (later: the four key quarters' partial outputs) | V [128][68] | scores / probabilities [32][132] |
            // slice max [32] | slice sum [32].  Row strides of 68 / 132 floats: the 16 lanes of a ds_read_b128 service group (16 rows, same
            // column) hit 16 distinct 16-byte slots.
            float *qs = reinterpret_cast<float *>(es_lds);                    // [32][68]
            float *ks = qs + 32 * ES_P2_LD;                                   // [128][68]
            float *vs = ks + 128 * ES_P2_LD;                                  // [128][68]
            float *ss = vs + 128 * ES_P2_LD;                                  // [32][132]
            float *sm = ss + 32 * ES_P2_LDS;                                  // [32] slice max
            float *sl = sm + 32;                                              // [32] slice sum
            const __amdgpu_buffer_rsrc_t qr = es_rsrc(a.qbuf), kr_ = es_rsrc(L.kring), vr_ = es_rsrc(L.vring);
            // thread -> (key, 4 dims) x 2 per tile
            es_u32x4 kk[2][2], vv[2][2];
            auto kv_off = [&](int tile, int it, int &pos) -> unsigned {
                const int i = tid + it * ES_THREADS, key = i >> 4, c = (i & 15) * 4;
                pos = t0 + 64 * tile + key;
                return (unsigned)(((size_t)(pos % a.ring_cap) * ES_QD + head * 64 + c) * 4);
            };
#pragma unroll
            for (int tile = 0; tile < 2; tile++)
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    int pos; const unsigned off = kv_off(tile, it, pos);
                    kk[tile][it] = es_u32x4{0u, 0u, 0u, 0u}; vv[tile][it] = kk[tile][it];
                    if (pos <= t1 && pos < a.pos0) { kk[tile][it] = es_ld16(kr_, off); vv[tile][it] = es_ld16(vr_, off); }
                }
            if (!es_wait(a, a.epoch + g, &s_ok, EsNeedHead{xcd, widx & 3})) return;
            ES_MARK(2);
            {
                const int r = tid >> 4, c = (tid & 15) * 4;                    // 512 threads = 32 rows x 16 column groups
                es_u32x4 v = es_u32x4{0u, 0u, 0u, 0u};
                if (r < n) v = es_ld16(qr, (unsigned)(((size_t)r * ES_QD + head * 64 + c) * 4));
#pragma unroll
                for (int tile = 0; tile < 2; tile++)
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        int pos; const unsigned off = kv_off(tile, it, pos);
                        if (pos <= t1 && pos >= a.pos0) { kk[tile][it] = es_ld16(kr_, off); vv[tile][it] = es_ld16(vr_, off); }      // this chunk's own rows
                    }
                *reinterpret_cast<es_u32x4 *>(&qs[r * ES_P2_LD + c]) = v;
#pragma unroll
                for (int tile = 0; tile < 2; tile++)
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int i = tid + it * ES_THREADS, key = tile * 64 + (i >> 4), cc = (i & 15) * 4;
                        *reinterpret_cast<es_u32x4 *>(&ks[key * ES_P2_LD + cc]) = kk[tile][it];
                        *reinterpret_cast<es_u32x4 *>(&vs[key * ES_P2_LD + cc]) = vv[tile][it];
                    }
            }
            __syncthreads();
            // Both products on the f32 matrix pipe (v_mfma_f32_32x32x2_f32: f32 products, f32 accumulation = the FMA chains of k_attn_small /
            // vox_causal_attention, voxtral_kernels.c:412-482, up to summation order).  The 32x32x2 operand map: lane (i = lane & 31, g = lane >> 5)
            // supplies A[i][k] and B[i][k] of the step's two k values; which k a (step, g) pair stands for is free as long as A and B agree:
            // g = the half of the reduction range, so a lane reads CONSECUTIVE floats.  C: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 g.
            const int i32 = lane & 31, g2 = lane >> 5;
            const int nk = t1 - t0 + 1;                                       // keys of this slice (<= 0: empty)
            // scores S[32 rows][key]: waves 0 .. 3 take 32 keys each (reduction over the 64 dims: 32 steps)
            if (wv < 4 && 32 * wv < nk) {                                     // (wave-uniform)
                float4 qa[8], kb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    qa[j] = *reinterpret_cast<const float4 *>(&qs[i32 * ES_P2_LD + 32 * g2 + 4 * j]);
                    kb[j] = *reinterpret_cast<const float4 *>(&ks[(32 * wv + i32) * ES_P2_LD + 32 * g2 + 4 * j]);
                }
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j].x, kb[j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j].y, kb[j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j].z, kb[j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j].w, kb[j].w, acc, 0, 0, 0);
                }
                ES_MFMA_SETTLE();
#pragma unroll
                for (int r = 0; r < 16; r++) ss[((r & 3) + 8 * (r >> 2) + 4 * g2) * ES_P2_LDS + 32 * wv + i32] = acc[r];
            }
            __syncthreads();
            // softmax numerators over the slice's keys: wave = 4 query rows, lane = keys lane and lane + 64 (columns no wave computed are masked:
            // they are read through a select, never through arithmetic)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = wv * 4 + r, P = a.pos0 + row;
                float sv[2]; bool ok[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int key = lane + 64 * h, t = t0 + key;
                    ok[h] = row < n && t <= t1 && t <= P && t >= P - a.window + 1;
                    sv[h] = ok[h] ? ss[row * ES_P2_LDS + key] * a.scale : -1e30f;
                }
                const float mx = as_dpp_max(fmaxf(sv[0], sv[1]));
                const float p0 = (ok[0] && mx > -1e29f) ? expf(sv[0] - mx) : 0.f, p1 = (ok[1] && mx > -1e29f) ? expf(sv[1] - mx) : 0.f;
                const float lsum = as_dpp_sum(p0 + p1);
                ss[row * ES_P2_LDS + lane] = p0; ss[row * ES_P2_LDS + lane + 64] = p1;
                if (lane == 0) { sm[row] = mx > -1e29f ? mx : -1e30f; sl[row] = lsum; }
            }
            __syncthreads();
            // P.V: wave -> (32-dim half dt, 32-key quarter q4): O_q4[32 rows][32 dims], reduction over the quarter's keys in 16 steps; the four
            // quarters meet in LDS (where K was)
            {
                const int dt = wv & 1, q4 = wv >> 1;
                float4 pa[4];
#pragma unroll
                for (int j = 0; j < 4; j++) pa[j] = *reinterpret_cast<const float4 *>(&ss[i32 * ES_P2_LDS + 32 * q4 + 16 * g2 + 4 * j]);
                float vb[16];
#pragma unroll
                for (int st = 0; st < 16; st++) vb[st] = vs[(32 * q4 + 16 * g2 + st) * ES_P2_LD + 32 * dt + i32];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j].x, vb[4 * j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j].y, vb[4 * j + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j].z, vb[4 * j + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j].w, vb[4 * j + 3], acc, 0, 0, 0);
                }
                ES_MFMA_SETTLE();
                float *red = ks;                                              // [4 quarters][32 rows][68]
#pragma unroll
                for (int r = 0; r < 16; r++) red[(q4 * 32 + (r & 3) + 8 * (r >> 2) + 4 * g2) * ES_P2_LD + 32 * dt + i32] = acc[r];
            }
            __syncthreads();
            {
                const __amdgpu_buffer_rsrc_t por = es_rsrc(a.part_o), pmr = es_rsrc(a.part_ml);
                const size_t base = ((size_t)head * ES_NSL + slice) * 32;
                const int row = tid >> 4, d4 = (tid & 15) * 4;
                if (row < n) {
                    float4 o = *reinterpret_cast<const float4 *>(&ks[row * ES_P2_LD + d4]);
#pragma unroll
                    for (int q4 = 1; q4 < 4; q4++) {
                        const float4 t = *reinterpret_cast<const float4 *>(&ks[(q4 * 32 + row) * ES_P2_LD + d4]);
                        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                    }
                    es_st16(por, (unsigned)(((base + row) * 64 + d4) * 4), es_f4(o.x, o.y, o.z, o.w));
                    if ((tid & 15) == 0) es_st8(pmr, (unsigned)((base + row) * 8), es_u32x2{__float_as_uint(sm[row]), __float_as_uint(sl[row])});
                }
            }
            ES_MARK(3);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== P3: wo partials of (head, 160-column group) ============================================
        {
            ES_IDS();
            const int head = xcd * 4 + (widx & 3);
            const int ng = widx >> 2;                     // 0 .. 7
            // wave wv: W tiles wv and wv + 8 (< 10) of the group's 10; K = this head's 64 attention columns = 2 k-steps
            uint4 wq[2][2];
            const __amdgpu_buffer_rsrc_t wr = es_rsrc(L.wo);
            auto issue = [&]() {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int tile = min(wv + 8 * t, 9);
                    const unsigned wo_ = (unsigned)(((160 * ng + 16 * tile + li) * ES_QD + head * 64 + kb * 8) * 2);
                    wq[t][0] = es_ldw(wr, wo_);
                    wq[t][1] = es_ldw(wr, wo_ + 64);
                }
            };
            es_poll(a, a.epoch + g, &s_ok, EsNeedHead{xcd, widx & 3});      // (wave 0 only)
            issue();
            if (!es_join(&s_ok)) return;
            ES_MARK(4);
            // merge the head's 8 key slices (k_attn_combine's arithmetic): thread = (row, 4 dims)
            unsigned char *afr = es_lds;                  // A fragments [2 ks][3 p][2 u][64 lanes][16 B] = 12 KiB
            {
                const __amdgpu_buffer_rsrc_t por = es_rsrc(a.part_o), pmr = es_rsrc(a.part_ml);
                const int row = tid >> 4, d4 = (tid & 15) * 4;
                const int rl = min(row, n - 1);
                es_u32x4 o[ES_NSL]; es_u32x2 ml[ES_NSL];
#pragma unroll
                for (int sI = 0; sI < ES_NSL; sI++) {
                    const size_t idx = ((size_t)head * ES_NSL + sI) * 32 + rl;
                    o[sI] = es_ld16(por, (unsigned)((idx * 64 + d4) * 4));
                    ml[sI] = es_ld8(pmr, (unsigned)(idx * 8));
                }
                float mm = -1e30f, ll = 0.f;
                float ov[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sI = 0; sI < ES_NSL; sI++) mm = fmaxf(mm, __uint_as_float(ml[sI].x));
#pragma unroll
                for (int sI = 0; sI < ES_NSL; sI++) {
                    const float f = expf(__uint_as_float(ml[sI].x) - mm);
                    ll += __uint_as_float(ml[sI].y) * f;
                    ov[0] += __uint_as_float(o[sI].x) * f; ov[1] += __uint_as_float(o[sI].y) * f;
                    ov[2] += __uint_as_float(o[sI].z) * f; ov[3] += __uint_as_float(o[sI].w) * f;
                }
                const float il = (row < n && ll > 0.f) ? 1.0f / ll : 0.f;
                if (row < 16 * MU) {
                    es_u32x2 ph, pm, pl;
                    es_split4(ov[0] * il, ov[1] * il, ov[2] * il, ov[3] * il, ph, pm, pl);
                    *reinterpret_cast<es_u32x2 *>(afr + es_elem_off(row, d4, 0)) = ph;
                    *reinterpret_cast<es_u32x2 *>(afr + es_elem_off(row, d4, 1)) = pm;
                    *reinterpret_cast<es_u32x2 *>(afr + es_elem_off(row, d4, 2)) = pl;
                }
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t wpr = es_rsrc(a.wo_part);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (wv + 8 * t < 10) {                    // wave-uniform
                    f32x4 acc[MU];
#pragma unroll
                    for (int u = 0; u < MU; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        union { uint4 u4; bf16x8_t v; } fb; fb.u4 = wq[t][j];
#pragma unroll
                        for (int p = 2; p >= 0; p--)
#pragma unroll
                            for (int u = 0; u < MU; u++) {
                                const bf16x8_t fa = *reinterpret_cast<const bf16x8_t *>(afr + es_frag_off(j, p, u, lane));
                                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb.v, acc[u], 0, 0, 0);
                            }
                    }
                    ES_MFMA_SETTLE();
                    const int col = 160 * ng + 16 * (wv + 8 * t) + li;
#pragma unroll
                    for (int u = 0; u < MU; u++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int m = 16 * u + 4 * kb + r;
                            if (m < n) es_st4(wpr, (unsigned)((((size_t)head * 32 + m) * ES_D + col) * 4), acc[u][r]);
                        }
                }
            }
            ES_MARK(5);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== F3: x' = x + bo + wo partials; planes of x' g2 ==========================================
        {
            ES_IDS();
            // (a finish task (row, column block cb) reads the partials of column group ng == cb from all 32 heads: workgroups f with f >> 5 == cb)
            if (w < n * ES_CB) { if (!es_wait(a, a.epoch + g, &s_ok, EsNeedShift{5, w & 7})) return; }
            else if (!es_wait(a, a.epoch + g, &s_ok, EsNeedNone())) return;
            ES_MARK(6);
            if (w < n * ES_CB)
                es_finish(a, w, a.xa, a.xb, a.wo_part, ES_HEADS, (size_t)32 * ES_D, L.bo, L.n2, a.ssq + 32 * ES_CB, true, es_lds);
            ES_MARK(7);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== P4: h = silu(gate) * up, 20 hidden units per workgroup =================================
        {
            ES_IDS();
            const int u0 = 20 * w;
            uint4 wg_[2][5], wu_[2][5];                   // gate (w1) / up (w3) fragments: tile 0 = units 0 .. 15, tile 1 = units 16 .. 19
            const __amdgpu_buffer_rsrc_t w1r = es_rsrc(L.w1), w3r = es_rsrc(L.w3);
            auto issue = [&]() {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int unit = u0 + min(16 * t + li, 19);
                    const unsigned off = (unsigned)((unit * ES_D + (5 * wv) * 32 + kb * 8) * 2);
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        wg_[t][j] = es_ldw(w1r, off + j * 64);
                        wu_[t][j] = es_ldw(w3r, off + j * 64);
                    }
                }
            };
            es_poll(a, a.epoch + g, &s_ok, EsNeedAll());            // (wave 0 only; every finish task: global)
            issue();
            if (!es_join(&s_ok)) return;
            ES_MARK(8);
            es_load_inv(a.ssq + 32 * ES_CB, n, a.eps, s_inv);
            f32x4 ag[2][MU], au[2][MU];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < MU; u++) { ag[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; au[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            es_u32x4 af[3][MU][3];                        // a ring of three k-steps (two ahead of the MFMAs)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int p = 0; p < 3; p++) af[j][u][p] = es_ld16(apl, es_frag_off(5 * wv + j, p, u, lane));
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (j + 2 < 5) {
#pragma unroll
                    for (int u = 0; u < MU; u++)
#pragma unroll
                        for (int p = 0; p < 3; p++) af[(j + 2) % 3][u][p] = es_ld16(apl, es_frag_off(5 * wv + j + 2, p, u, lane));
                }
                __builtin_amdgcn_sched_barrier(0);        // (keeps the unrolled loop from requesting all five k-steps at once: 120 registers next to 80 of weights)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    union { uint4 u4; bf16x8_t v; } fg, fu; fg.u4 = wg_[t][j]; fu.u4 = wu_[t][j];
#pragma unroll
                    for (int p = 2; p >= 0; p--)
#pragma unroll
                        for (int u = 0; u < MU; u++) {
                            union { es_u32x4 u4; bf16x8_t v; } fa; fa.u4 = af[j % 3][u][p];
                            ag[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fg.v, ag[t][u], 0, 0, 0);
                            au[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fu.v, au[t][u], 0, 0, 0);
                        }
                }
            }
            // red[wave][gate / up][row 32][unit 32 (20 used)], row stride 33
            ES_MFMA_SETTLE();
            float *red = reinterpret_cast<float *>(es_lds);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        red[((wv * 2 + 0) * 32 + 16 * u + 4 * kb + r) * 33 + 16 * t + li] = ag[t][u][r];
                        red[((wv * 2 + 1) * 32 + 16 * u + 4 * kb + r) * 33 + 16 * t + li] = au[t][u][r];
                    }
            __syncthreads();
            // thread = (row m, 4 units): add the splitters in wave order, * inv, silu(gate) * up, planes of h
            if (tid < 32 * 5) {
                const int m = tid / 5, c = (tid - m * 5) * 4;
                if (m < n) {
                    const float inv = s_inv[m];
                    float hv[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float gs = red[(0 * 32 + m) * 33 + c + i], us = red[(1 * 32 + m) * 33 + c + i];
#pragma unroll
                        for (int k = 1; k < 8; k++) { gs += red[((k * 2 + 0) * 32 + m) * 33 + c + i]; us += red[((k * 2 + 1) * 32 + m) * 33 + c + i]; }
                        hv[i] = silu(gs * inv) * (us * inv);
                    }
                    es_u32x2 ph, pm, pl;
                    es_split4(hv[0], hv[1], hv[2], hv[3], ph, pm, pl);
                    const __amdgpu_buffer_rsrc_t hr = es_rsrc(a.hplanes);
                    const int k = u0 + c;
                    es_st8(hr, es_elem_off(m, k, 0), ph); es_st8(hr, es_elem_off(m, k, 1), pm); es_st8(hr, es_elem_off(m, k, 2), pl);
                }
            }
            ES_MARK(9);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== P5: w2 partials of (K group kg, row group ng) ==========================================
        {
            ES_IDS();
            const int kg = xcd * 2 + (widx & 1), ng = widx >> 1;      // the 16 workgroups of a K group share an XCD (and its L2 copy of that h range)
            // wave wv: k-steps wv and wv + 8 (< 10) of the group's 10, all 5 row tiles
            uint4 wq[2][5];
            const __amdgpu_buffer_rsrc_t wr = es_rsrc(L.w2);
            auto issue = [&]() {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int ksl = min(wv + 8 * j, 9);
                    if (wv + 8 * j < 10) {                // (wave-uniform: waves 2 .. 7 own one k-step)
#pragma unroll
                        for (int t = 0; t < 5; t++)
                            wq[j][t] = es_ldw(wr, (unsigned)(((80 * ng + 16 * t + li) * ES_H + (10 * kg + ksl) * 32 + kb * 8) * 2));
                    } else {
#pragma unroll
                        for (int t = 0; t < 5; t++) wq[j][t] = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            };
            es_poll(a, a.epoch + g, &s_ok, EsNeedShift{4, kg});     // (wave 0 only; h units 320 kg .. + 319 = P4's workgroups 16 kg .. + 15)
            issue();
            if (!es_join(&s_ok)) return;
            ES_MARK(10);
            const __amdgpu_buffer_rsrc_t hr = es_rsrc(a.hplanes);
            es_u32x4 af[2][MU][3];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int p = 0; p < 3; p++)
                        af[j][u][p] = (wv + 8 * j < 10) ? es_ld16(hr, es_frag_off(10 * kg + min(wv + 8 * j, 9), p, u, lane)) : es_u32x4{0u, 0u, 0u, 0u};
            f32x4 acc[5][MU];
#pragma unroll
            for (int t = 0; t < 5; t++)
#pragma unroll
                for (int u = 0; u < MU; u++) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (wv + 8 * j < 10) {                    // wave-uniform (the accumulators are read behind ES_MFMA_SETTLE)
#pragma unroll
                    for (int t = 0; t < 5; t++) {
                        union { uint4 u4; bf16x8_t v; } fb; fb.u4 = wq[j][t];
#pragma unroll
                        for (int p = 2; p >= 0; p--)
#pragma unroll
                            for (int u = 0; u < MU; u++) {
                                union { es_u32x4 u4; bf16x8_t v; } fa; fa.u4 = af[j][u][p];
                                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[t][u], 0, 0, 0);
                            }
                    }
                }
            }
            // red[wave][row 32][col 80], row stride 81
            ES_MFMA_SETTLE();
            float *red = reinterpret_cast<float *>(es_lds);
#pragma unroll
            for (int t = 0; t < 5; t++)
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[(wv * 32 + 16 * u + 4 * kb + r) * 81 + 16 * t + li] = acc[t][u][r];
            __syncthreads();
            const __amdgpu_buffer_rsrc_t w2r = es_rsrc(a.w2_part);
            for (int e = tid; e < 32 * 20; e += ES_THREADS) {
                const int m = e / 20, c = (e - m * 20) * 4;
                if (m < n) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        v[i] = red[m * 81 + c + i];
#pragma unroll
                        for (int k = 1; k < 8; k++) v[i] += red[(k * 32 + m) * 81 + c + i];
                    }
                    es_st16(w2r, (unsigned)((((size_t)kg * 32 + m) * ES_D + 80 * ng + c) * 4), es_f4(v[0], v[1], v[2], v[3]));
                }
            }
            ES_MARK(11);
            es_arrive(a, a.epoch + (++g));
        }

        // =========================== F5: x = x' + b2 + w2 partials; planes of x g1[l + 1] ===================================
        {
            ES_IDS();
            // (columns 160 cb .. + 159 = P5's row groups 2 cb and 2 cb + 1, all 16 K groups: workgroups f with f >> 5 == cb)
            if (w < n * ES_CB) { if (!es_wait(a, a.epoch + g, &s_ok, EsNeedShift{5, w & 7})) return; }
            else if (!es_wait(a, a.epoch + g, &s_ok, EsNeedNone())) return;
            ES_MARK(12);
            const bool last = l + 1 == a.n_layers;
            if (w < n * ES_CB)
                es_finish(a, w, a.xb, a.xa, a.w2_part, ES_KG5, (size_t)32 * ES_D, L.b2, last ? nullptr : a.layers[l + 1].n1, a.ssq, !last, es_lds);
            ES_MARK(13);
            es_arrive(a, a.epoch + (++g));
        }
        if (TL && tlon) {
            ES_IDS();
            __syncthreads();
            if (tid == 0) {
                unsigned long long *r = a.tl + (size_t)ES_TL_STRIDE * w;
                r[0] = tl0; r[1] = wall_clock64();
                r[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
                for (int k = 0; k < 2 * ES_PHASES; k++) r[3 + k] = es_stamp[k];
            }
        }
    }
}

}  // namespace vox
