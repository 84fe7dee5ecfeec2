// vox_decfuse.h — the attention half of a decoder step as ONE launch (4B geometry).
//
// Replaces, per layer and per token, three launches of the launch-per-GEMV chain
//     k_gemv3<RMS,QKV> (37.7 MB)  ->  k_attn_dec (KV window)  ->  k_gemv3<ATTN,RESID> Wo (25.2 MB)
// (reference voxtral_decoder.c:653-676: attention_norm, wq/wk/wv, RoPE, KV append, attention, wo).
// Measured in round 1 (profiles/r01_run7_kernel_stats.csv): 9.6 + 6.3 + 9.0 us for 63 MB = 2.5 TB/s —
// these three short kernels carry most of the decode step's distance to the HBM roofline, because each
// pays a kernel boundary, a first-byte latency and a drain for 4-6 us of actual streaming.
//
// What makes one launch possible without a grid-wide barrier: the chain is all-to-all only at its two
// ends.  In between it splits into 8 independent strands, one per KV head g:
//     q heads 4g..4g+3 (512 rows of Wq), k / v head g (128 rows each)  ->  attention of those 4 heads
//     ->  their 512 columns of Wo (a K-split of the output projection).
// A strand is run by a group of 32 workgroups (grid = 8 x 32 = 256 = one workgroup per CU, group =
// blockIdx % 8 so that a group sits on one XCD when placement follows the usual b % 8 rule — for speed
// only, nothing depends on it).  Inside a group two hand-offs are needed: the 768 q/k/v values every
// member needs for its key slice, and the split-K attention partials every member needs to merge.  Both
// use data-tagged 8-byte {epoch, value} granules written with one write-through (sc1) store and swept
// with L1-bypassing loads until every tag matches (MI355X_MICROARCH.md, visibility section, form R2):
// no flag, no fence, the data is the flag.  The epoch is the engine's launch counter, so buffers are
// reused launch after launch without clearing.
//
// Memory queue of a workgroup (a CU returns loads in issue order, so order = schedule; DESIGN.md 8.1 has the timeline):
//   t0  inv_freq, x, norm weights                    by LDS-DMA   (needed first)
//       3 weight rows per wave (18 x 16 B per lane)  registers, non-temporal, ordinary loads (compiler-counted waits)
//   t1  RMSNorm, then the dot products piece by piece as the weights land; RoPE; publish q/k/v
//   attention members (j < nsplit):  K/V tile (LDS-DMA) behind the last weight piece -> sweep q/k/v -> attention over their
//       key slice -> publish the partial.  Up to 8 slices they carry no Wo rows and are done here.
//   the others:  13-16 Wo rows per wave queued after their publish -> ONE thread polls for the partials -> batch sweep, merge
//       -> Wo partial product from registers.
// Output: wo_part[g][3072], the 8 K-split partial sums of the projection; the next launch
// (k_gemv_w13x, below) adds them to the residual stream in a fixed order in its prologue.
//
// Every spin is bounded (wall clock); a timeout sets *err and the host re-runs the batch on the
// launch-per-GEMV chain and keeps it (vox_hip_active_paths loses VOX_PATH_DEC_FUSED).
//
// What is in this file, in the order of a decoder step (round 4):
//   k_dec_attn_fused   attention block of a layer, 8 waves: layer 0 (with the embedding gather), every layer beyond 1024 keys
//   k_gemv_w13x / k_gemv_w2x   FFN block as two launches (fp8 mode; A/B of k_ffn_fused)
//   k_ffn_fused = ffn_body     FFN block as one launch (h handed over inside, behind the W2 bytes); last layer, every layer beyond 1024 keys
//   df_attn12_body / k_attn12  the attention block re-cut for the FFN kernels' shape (12 waves, <= 168 registers), <= 8 key slices of one or two tiles, bf16 or fp8
//   k_ffn_attn12       FFN block of layer l + attention block of layer l + 1 as ONE launch (x'' in granules): 25 per token up to 1024 keys (two-tile attention members beyond 512)
//   k_w2x_attn12       fp8 mode: W2 launch of layer l + fp8 attention block of layer l + 1 as one launch
#pragma once
#include "vox_common.h"

namespace vox {

constexpr int DF_GROUPS = 8, DF_BPG = 32, DF_BLOCKS = DF_GROUPS * DF_BPG;
constexpr int DF_D = 3072, DF_DQ = 4096, DF_DKV = 1024, DF_HD = 128;
constexpr int DF_NQ = 512;                    // q values per group (4 heads x 128)
constexpr int DF_GQ = DF_NQ + 2 * DF_HD;      // granules of the q/k/v hand-off per group
constexpr int DF_GP = 4 * DF_HD + 8;          // granules of one attention partial: o[4][128], (m, l)[4]
constexpr int DF_WO_ROWS = DF_D / DF_BPG;     // 96 rows of Wo per workgroup

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------------------
// L2 prefetch across a kernel boundary (round 4).  From ~11 us on k_dec_attn_fused keeps the memory system idle: the projection
// weights and the Wo rows have landed and what remains is a chain of L2 round trips (partials -> sweep -> Wo product), then the
// kernel boundary and the next launch's ramp: ~6 us per layer in which no weight byte moves (profiles/r03_fuse_timeline_kv232.txt).
// The next launch (k_gemv_w13x) is HBM bound from its first microsecond.  Workgroup b of every 256-block launch of the step runs on
// the same XCD (observed: XCC_ID = (b + 6) % 8 for all three launches, tools/fuse_timeline.py), so this kernel can pull the first
// 1 KiB pieces of the rows k_gemv_w13x's block b will ask for into THAT XCD's L2: LDS-DMA into a scratch slot (no registers, the
// data is discarded), default cache policy.  Unit u of target block bt: matrix m = (u % 72) / 36 (w1, w3), row 36 bt + u % 36,
// piece u / 72 - i.e. piece 0 of all 72 rows first, in the order k_gemv_w13x issues them.  The units of the 32 target blocks of
// an XCD class (blockIdx % 8) are dealt round-robin over that class's prefetching waves, so that every target block is covered
// to the same depth whoever does the work: the attention members (done once their partial is out; they take `member_units` of the
// class's 32 * units) and the others (either right behind their Wo rows, when = 1, or after the partial sweep, when = 2).
// Nothing depends on it: wrong placement or an evicted line only costs the gain.
// ---------------------------------------------------------------------------------------------------------
struct DfPrefetch {
    const unsigned char *w;    // w1 rows, then w3 rows (row r of w3 = row rows_m + r)
    int row_bytes;             // 6144 (bf16) / 3072 (fp8)
    int rows_m;                // 9216
    int units;                 // 1 KiB units per target block (0 = off)
    int member_units;          // of the 32 * units of an XCD class, how many the attention members fetch
    int when;                  // non-members: 1 = behind their Wo rows, 2 = after the partial sweep
};
__device__ __forceinline__ void df_prefetch_units(const DfPrefetch &pf, int cls, int v0, int v1, int slot, int nslots, int lane, unsigned lds_dst) {
    for (int v = v0 + slot; v < v1; v += nslots) {
        const int bt = 8 * (v & 31) + cls, u = v >> 5;
        const int piece = u / 72, r72 = u - 72 * piece;
        const int m = r72 >= 36 ? 1 : 0, r = r72 - 36 * m;
        const unsigned char *src = pf.w + ((size_t)m * pf.rows_m + 36 * bt + r) * (size_t)pf.row_bytes + piece * 1024 + lane * 16;
        glds16(src, lds_dst);
    }
}

struct DecFuseArgs {
    const uint16_t *wqkv;      // [6144][3072] bf16: q rows, then k rows, then v rows  (W8: fp8 e4m3 bytes, one f32 scale per row in sqkv)
    const uint16_t *wo;        // [3072][4096] bf16                                     (W8: fp8 e4m3 bytes, one f32 scale per row in so)
    const float *sqkv, *so;
    const float *x;            // [3072] residual stream (layers > 0)
    const float *norm_w;       // [3072] attention_norm
    float eps;
    const float *inv_freq;     // [64], allocation padded to 1 KiB
    float *kring, *vring;      // [kv_cap][1024]
    int kv_cap, pos, window;
    float scale;
    // layer 0: x = adapter[st->adapter_row] + tok_emb[st->token]  (voxtral.c:1057-1061), also written to x_out
    const float *adapter; const uint16_t *tok_emb; const DecState *st; float *x_out;
    u64 *gq;                   // [8][DF_GQ]
    u64 *gp;                   // [8][32][DF_GP]
    float *wo_part;            // [8][3072]
    unsigned epoch;
    int split_keys, nsplit;    // keys per slice (multiple of 64), active slices (<= 32)
    unsigned *err;
    unsigned long long spin_limit;   // wall_clock64 ticks
    unsigned long long *tl;          // optional (tuning): per-workgroup timeline, see tl_begin / tl_end
    // Round 4: L2 prefetch of the NEXT launch's (k_gemv_w13x) first weight bytes in this kernel's tail, when the memory system is
    // idle (DfPrefetch below).  pf.units = 0 switches it off.
    DfPrefetch pf;
};
// stamps stay in registers until the end (no stores in the middle of the memory schedule)
#define DF_MARK(k) do { if (a.tl) df_stamp[k] = wall_clock64(); } while (0)

__device__ __forceinline__ void df_store_granule(u64 *g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 df_load_granule(const u64 *g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The thread index as a value the compiler cannot see through.  k_dec_stack runs the FFN and attention bodies inside a loop over the
// layers: with a plain threadIdx.x every address the bodies derive from it is loop-invariant, gets hoisted out of the loop and kept
// live across it - hundreds of spilled registers.  Laundered once per body entry, the arithmetic stays where it is written.
__device__ __forceinline__ int df_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int df_bid() { int b = blockIdx.x; asm volatile("" : "+s"(b)); return b; }
// (the bounded-spin helper DfSpin / df_spin_expired lives in vox_common.h: the encoder stack kernel uses it too)
// Re-read one granule until its tag is this launch's epoch (bounded).  Returns the payload.
// Once ANY wait of ANY launch has timed out (*err != 0: e.g. the 256 workgroups were not co-resident because another
// process or stream held CUs), every later wait gives up at once: the launches already queued behind the failure then
// drain in microseconds instead of one time-out each, and the host re-runs the batch on the launch-per-GEMV chain.
//
__device__ __forceinline__ float df_wait_granule_e(const u64 *g, unsigned epoch, unsigned *err, unsigned long long spin_limit, unsigned code) {
    u64 v = df_load_granule(g);
    if ((unsigned)(v >> 32) != epoch) {
        DfSpin sp; df_spin_begin(sp);
        for (unsigned it = 0;; it++) {
            __builtin_amdgcn_s_sleep(2);
            v = df_load_granule(g);
            if ((unsigned)(v >> 32) == epoch) break;
            if ((it & 15u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (df_spin_expired(sp, err, spin_limit, code, epoch)) break;
        }
    }
    return __uint_as_float((unsigned)v);
}
__device__ __forceinline__ float df_wait_granule(const u64 *g, unsigned epoch, const DecFuseArgs &a, unsigned code) {
    return df_wait_granule_e(g, epoch, a.err, a.spin_limit, code);
}

// Wave-wide sum: four DPP row steps, then the four row sums through readlane (no LDS traffic, unlike ds_bpermute
// butterflies: measured 2.8 us for 24 of those per wave at the end of this kernel).  Result is wave-uniform.
template <bool USE_DPP>
__device__ __forceinline__ float df_wave_sum(float v) {
    if constexpr (USE_DPP) {
        v = row16_sum<true>(v);
        const int iv = __float_as_int(v);
        return __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
               __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    } else {
        return wave_sum(v);
    }
}

template <bool USE_DPP>
__device__ __forceinline__ float df_wave_max(float v) {
    if constexpr (USE_DPP) {
        int x;
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));
        const int iv = __float_as_int(v);
        return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                     fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
    } else {
        return wave_max(v);
    }
}

constexpr int DF_THREADS = 512, DF_WAVES = 8;
constexpr int DF_TILE = 64;                       // keys per K/V tile in LDS
constexpr int DF_TILE_BYTES = DF_TILE * 512;      // one of K or V: 64 keys x 128 f32
// LDS: [xs | nw] 24 KB (scratch after the projections) | K/V tiles, double buffered 2 x (32 + 32) KB | inv_freq | red
constexpr int DF_LDS_BYTES = 2 * DF_D * 4 + 4 * DF_TILE_BYTES + 1024 + 512;

// One K/V tile (keys t0 .. t0+63 of KV head g, clamped to last) from the ring into LDS by LDS-DMA: 8 instructions per
// wave (4 for K, 4 for V), each moving two 512-byte head rows.  K is stored with its 16-byte chunks XOR-swizzled by the
// key index (the DMA destination is linear, but which global chunk a lane fetches is free) so that the score phase, where
// a lane owns a key and walks its row, reads conflict-free; V stays linear (the PV phase walks keys with lanes on dims).
// Op k of a wave's 8 (k >> 1 = key pair, k & 1 = V).  `real` false (workgroups without a key slice) turns the op into a
// 16-byte broadcast read of one valid address, so that every workgroup issues the same number of memory operations and
// the hand-counted s_waitcnt values below hold for all of them.
__device__ __forceinline__ void df_tile_op(const DecFuseArgs &a, int g, int t0, int last, unsigned kt_lds, unsigned vt_lds, int wave, int lane,
                                           int k, bool real, int slot0) {
    const int pair = 4 * wave + (k >> 1);
    const int key = 2 * pair + (lane >> 5);
    // ring slot of key t0 + key (clamped to last): slot0 = t0 % kv_cap is computed once per tile by the caller (a 32-bit
    // modulo per lane and operation cost the attention members ~0.3 us right in front of their publish)
    int sl = slot0 + min(key, last - t0); if (sl >= a.kv_cap) sl -= a.kv_cap;
    const size_t row = (size_t)sl * DF_DKV + g * DF_HD;
    const int cs = lane & 31;
    const float *src = (k & 1) ? a.vring + row + (cs << 2) : a.kring + row + ((cs ^ (key & 31)) << 2);
    if (!real) src = a.kring;
    glds16(src, ((k & 1) ? vt_lds : kt_lds) + (unsigned)pair * 1024u);
}
__device__ __forceinline__ void df_tile_dma(const DecFuseArgs &a, int g, int t0, int last, unsigned kt_lds, unsigned vt_lds, int wave, int lane) {
    const int slot0 = __builtin_amdgcn_readfirstlane(t0 % a.kv_cap);
#pragma unroll
    for (int k = 0; k < 8; k++) df_tile_op(a, g, t0, last, kt_lds, vt_lds, wave, lane, k, true, slot0);
}

// W8 (round 4, BASELINE config 5): the projection and Wo matrices are the row-scaled fp8 copies (half the bytes: 9 instead of
// 18 weight loads per lane, 8 instead of 16 Wo loads per wave - a Wo load then covers TWO rows, 32 lanes x 16 weights each).
template <bool EMBED, bool USE_DPP, bool W8 = false>
__global__ __launch_bounds__(DF_THREADS, 2) void k_dec_attn_fused(const DecFuseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *xs = reinterpret_cast<float *>(smem_raw);                           // [3072] x, then [3072] norm weights (contiguous)
    float *nw = xs + DF_D;
    unsigned char *tiles = smem_raw + 2 * DF_D * 4;                            // [2 buffers][K tile | V tile]
    float *frq = reinterpret_cast<float *>(tiles + 4 * DF_TILE_BYTES);         // [256] inv_freq (64 valid)
    float *red = frq + 256;                                                    // [128]: wave sums, the 24 row results
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Up to 8 key slices a group sits on one XCD (group = blockIdx % 8: the hand-offs stay inside one L2); beyond, its members are
    // spread over all XCDs (group = blockIdx / 32): the hand-offs then cross XCDs (+0.3 us each), but a group's K/V tiles and Wo rows
    // come through all eight L2s - measured, ms per step at 232 / 600 / 1900 / 3800 / 8000 keys: 1.382 / 1.485 / 1.593 / 1.724 / 1.896
    // on one XCD per group against 1.392 / 1.478 / 1.562 / 1.679 / 1.860 spread.
    const bool spread = a.nsplit > 8;
    const int g = spread ? blockIdx.x / DF_BPG : blockIdx.x % DF_GROUPS;
    const int j = spread ? blockIdx.x % DF_BPG : blockIdx.x / DF_GROUPS;
    const unsigned epoch = a.epoch;
    const int pos = a.pos;
    unsigned long long df_stamp[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl0 = tl_begin(a.tl);
    DF_MARK(0);

    // ---- this workgroup's 24 weight rows: 16 of Wq, 4 of Wk, 4 of Wv; wave w streams rows 3w .. 3w+2 -------------
    constexpr int NPW = W8 ? 3 : 6;               // 1 KiB pieces per projection row
    constexpr int NWO = W8 ? 8 : 16;              // Wo loads per wave (16 rows)
    const unsigned char *rp[3];
    int prow[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int lr = 3 * wave + i;
        const int row = lr < 16 ? DF_NQ * g + 16 * j + lr
                      : lr < 20 ? DF_DQ + DF_HD * g + 4 * j + (lr - 16)
                                : DF_DQ + DF_DKV + DF_HD * g + 4 * j + (lr - 20);
        prow[i] = row;
        rp[i] = reinterpret_cast<const unsigned char *>(a.wqkv) + (size_t)row * (W8 ? DF_D : DF_D * 2) + lane * 16;
    }
    // ---- this workgroup's key slice ---------------------------------------------------------------------------------
    const int ns = a.nsplit;
    int lo = pos - a.window + 1; if (lo < 0) lo = 0;
    const int s_lo = lo + j * a.split_keys;
    int s_hi = s_lo + a.split_keys - 1; if (s_hi > pos) s_hi = pos;
    const bool att_block = j < ns;
    const int n_tiles = att_block ? (s_hi - s_lo + DF_TILE) / DF_TILE : 0;

    // ---- t0: everything this workgroup will read, in the order it is needed.  A CU's memory path is one FIFO for all of
    // its waves, so the barrier below keeps every wave's projection weights ahead of anybody's K/V tile and Wo slice. ------
    glds16(reinterpret_cast<const unsigned char *>(a.inv_freq) + lane * 16, lds_addr(frq));     // same 1 KiB from every wave
    if constexpr (!EMBED) {
        // [x | norm_w] = 24 KB as three 8 KB rounds into the contiguous [xs | nw]
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int e = p * 2048 + tid * 4;
            const float *src = e < DF_D ? a.x + e : a.norm_w + (e - DF_D);
            glds16(src, lds_addr(xs) + (unsigned)(p * 8192 + wave * 1024));
        }
    }
    // The CU issues its waves oldest first: without this barrier wave 0 would put all of its weight loads into the CU's
    // memory FIFO before wave 7 has issued its share of the activation vector (measured: activations landing at 4.7 us
    // instead of ~2, and at 15 us in the 12-wave k_gemv_w13x below).
    DF_MARK(11);
    __builtin_amdgcn_s_barrier();
    DF_MARK(12);
    // The weight loads (and the Wo rows later) are ordinary non-temporal loads the compiler counts.  They used to be inline
    // asm with hand-counted s_waitcnt: a register an asm load is still writing looks "defined" to the compiler, which is then
    // free to copy it or to reuse it before the data has landed - it did both as soon as the surrounding code changed (seen in
    // the ISA: v_mov of the last piece's registers in front of the wait; the sweep's temporaries on top of in-flight Wo rows).
    // What made asm necessary is gone: no LDS-DMA (asm, uncounted) is issued between the weight loads and their use any
    // more, so the compiler's in-order wait counts are exact here, and where asm DMAs are queued in front they only over-wait.
    uint4 w[3][NPW];
#pragma unroll
    for (int c = 0; c < NPW; c++)
#pragma unroll
        for (int i = 0; i < 3; i++) w[i][c] = ld_stream(reinterpret_cast<const uint4 *>(rp[i] + c * 1024));
    __builtin_amdgcn_sched_barrier(0);
    // The first K/V tile (it does not depend on this step's q) and this workgroup's slice of Wo (registers: row i =
    // Wo[96 j + i][512 g .. 512 g + 511] = 1 KiB, 12 rows per wave) follow the projection weights in the queue.  They are
    // issued four at a time behind each weight piece as it lands: a wave that tried to issue them all up front would sit in
    // the issue stage behind the CU's full memory queue instead of doing the RMSNorm and the dot products (measured: +4 us).
    // (Issuing Wo only after the first hand-off sweep was measured too: the sweep is gated by the slowest of the group's 32
    // producers, ~3 us behind the median, and Wo streaming during that wait is worth more than a shorter sweep.)
    // Wo rows (row r = Wo[r][512 g .. 512 g + 511] = 1 KiB).  Up to 8 key slices (KV <= 512: the 30 s clip) the attention members
    // carry none - they are the critical path, everybody waits for their partials - and the other 32 - ns members share the
    // group's 3072 rows (13 .. 16 per wave); beyond that every member takes 96 (12 per wave).  A wave always issues 16 loads
    // (rows past its share re-read its last one) so that the hand-counted waits below hold for every workgroup.
    uint4 wv[NWO];       // compiler-visible non-temporal loads (see DF_LATE_WO)
    float wo_sc[W8 ? NWO : 1];      // W8: the dequantisation scale of the row a lane's half of load i belongs to, fetched with the rows
    const bool wo_light = ns <= 8;
    int wo_rpw = 12, wo_row0 = DF_WO_ROWS * j + 12 * wave;
    if (wo_light) {
        const int nb = DF_BPG - ns, rpb = ((DF_D + nb - 1) / nb + 7) & ~7;
        wo_rpw = rpb >> 3;
        wo_row0 = att_block ? DF_D : (j - ns) * rpb + wave * wo_rpw;
    }
    const int wo_n = max(0, min(wo_rpw, DF_D - wo_row0));
    const int wo_rmax = wo_n > 0 ? wo_row0 + wo_n - 1 : DF_D - 1;
    // bf16: load i = row wo_row0 + i, 64 lanes x 8 weights; fp8: load i = rows wo_row0 + 2 i (lanes 0-31) and + 2 i + 1 (lanes 32-63), 16 weights per lane
    const unsigned char *wo_base = reinterpret_cast<const unsigned char *>(a.wo) + (size_t)(DF_NQ * g) * (W8 ? 1 : 2) + (W8 ? (lane & 31) : lane) * 16;
#define DF_WO_PTR(i) (wo_base + (size_t)min(wo_row0 + (W8 ? 2 * (i) + (lane >> 5) : (i)), wo_rmax) * (W8 ? DF_DQ : DF_DQ * 2))
#define DF_WO_SCALE(i) a.so[min(wo_row0 + 2 * (i) + (lane >> 5), wo_rmax)]
    const int tile_last = att_block ? s_hi : 0;
    const int tile_slot0 = __builtin_amdgcn_readfirstlane(att_block ? s_lo % a.kv_cap : 0);
    DF_MARK(1);

    // ---- the activation vector -----------------------------------------------------------------------------------
    if constexpr (EMBED) {
        // layer 0: compiler-counted loads issued behind the weight stream (they wait for all of it; it is needed before
        // the dot products anyway); the Wo slice goes after them
        const float *arow = a.adapter + (size_t)a.st->adapter_row * DF_D;
        const uint16_t *erow = a.tok_emb + (size_t)a.st->token * DF_D;
        for (int e = tid * 4; e < DF_D; e += DF_THREADS * 4) {
            float4 v = *reinterpret_cast<const float4 *>(arow + e);
            const uint2 eb = *reinterpret_cast<const uint2 *>(erow + e);
            v.x += bf16_lo(eb.x); v.y += bf16_hi(eb.x); v.z += bf16_lo(eb.y); v.w += bf16_hi(eb.y);
            *reinterpret_cast<float4 *>(xs + e) = v;
            *reinterpret_cast<float4 *>(nw + e) = *reinterpret_cast<const float4 *>(a.norm_w + e);
            if (blockIdx.x == 0) *reinterpret_cast<float4 *>(a.x_out + e) = v;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) df_tile_op(a, g, s_lo, tile_last, lds_addr(tiles), lds_addr(tiles + DF_TILE_BYTES), wave, lane, k, att_block, tile_slot0);
#pragma unroll
        for (int i = 0; i < NWO; i++) { wv[i] = ld_stream(reinterpret_cast<const uint4 *>(DF_WO_PTR(i))); if constexpr (W8) wo_sc[i] = DF_WO_SCALE(i); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + (W8 ? 2 * NWO : NWO)) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory");   // only the 18 (fp8: 9) weight loads are younger than the activation DMAs
    }
    __syncthreads();
    DF_MARK(2);
    {   // RMSNorm (voxtral_kernels.c:346-363), under the weight stream
        float ss = 0.f;
        for (int e = tid * 4; e < DF_D; e += DF_THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + e);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = df_wave_sum<USE_DPP>(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < DF_WAVES; i++) tot += red[i];
        const float inv = 1.0f / sqrtf(tot / (float)DF_D + a.eps);
        for (int e = tid * 4; e < DF_D; e += DF_THREADS * 4) {
            const float4 gw = *reinterpret_cast<const float4 *>(nw + e);
            float4 o = *reinterpret_cast<const float4 *>(xs + e);
            o.x = o.x * inv * gw.x; o.y = o.y * inv * gw.y; o.z = o.z * inv * gw.z; o.w = o.w * inv * gw.w;
            *reinterpret_cast<float4 *>(xs + e) = o;
        }
        __syncthreads();
    }
    DF_MARK(3);

    // RoPE factors of the 10 rotated row pairs (8 of q, 2 of k), computed by the 12 threads that will publish them while
    // the weights are still landing: angle = pos * inv_freq, accurate cosf / sinf (voxtral_kernels.c:488-500)
    float rope_c = 1.f, rope_s = 0.f;
    if (tid < 10) {
        const int in_head = tid < 8 ? (16 * j + 2 * tid) % DF_HD : 4 * j + 2 * (tid - 8);
        const float ang = (float)pos * frq[in_head >> 1];
        rope_c = cosf(ang); rope_s = sinf(ang);
    }

    // ---- dot products, piece by piece as the weights land ------------------------------------------------------------
    float acc[3] = {0.f, 0.f, 0.f};
#define DF_PIECE(C)                                                                                      \
    if constexpr ((C) < NPW) {                                                                           \
        constexpr int CC = (C) < NPW ? (C) : 0;                                                          \
        if constexpr (W8) {                                                                              \
            const float *xp = xs + (CC * 64 + lane) * 16;                                                \
            const float4 x0 = *reinterpret_cast<const float4 *>(xp), x1 = *reinterpret_cast<const float4 *>(xp + 4);          \
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8), x3 = *reinterpret_cast<const float4 *>(xp + 12);     \
            _Pragma("unroll") for (int i = 0; i < 3; i++) acc[i] = dot16_fp8(w[i][CC], x0, x1, x2, x3, acc[i]);               \
        } else {                                                                                         \
            const float4 x0 = *reinterpret_cast<const float4 *>(xs + (CC * 64 + lane) * 8);              \
            const float4 x1 = *reinterpret_cast<const float4 *>(xs + (CC * 64 + lane) * 8 + 4);          \
            _Pragma("unroll") for (int i = 0; i < 3; i++) acc[i] = dot8_bf16(w[i][CC], x0, x1, acc[i]);  \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
    // the Wo loads come in four batches (I0 = 0, 4, 8, 12 in units of bf16 loads: a quarter of the wave's loads each)
#define DF_LATE_WO(I0)                                                                                   \
    if constexpr (!EMBED) { _Pragma("unroll") for (int i = (I0) * NWO / 16; i < ((I0) + 4) * NWO / 16; i++) {                \
        wv[i] = ld_stream(reinterpret_cast<const uint4 *>(DF_WO_PTR(i))); if constexpr (W8) wo_sc[i] = DF_WO_SCALE(i); } }
    // The two kinds of workgroup order their memory queue differently (per-workgroup timeline, tools/fuse_timeline.py).
    // The members that run attention (j < nsplit: 4 of a group's 32 at the 30 s clip's KV length) gate everybody: nobody's
    // sweep completes before the LAST member has published, and all wait for their partials.  With the K/V tile (64 KB) and
    // Wo rows (96 KB) interleaved into the weight stream they published 3 us after the others (12.2 vs 9.1 us), and Wo rows
    // of the early finishers competed with the projection weights of the late ones.  So: everybody takes nothing but the
    // projection weights first; an attention member queues its K/V tile behind the last piece (it lands under the publish
    // and the sweep); the others, who have no use for q/k/v and nothing to do until the partials arrive, queue their Wo rows
    // after their publish.
    DF_PIECE(0) DF_PIECE(1) DF_PIECE(2) DF_PIECE(3) DF_PIECE(4) DF_PIECE(5)
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) :: "memory");     // the tile DMAs below stay below the dot products
    if (att_block) {
        if constexpr (!EMBED) {
#pragma unroll
            for (int k = 0; k < 8; k++) df_tile_op(a, g, s_lo, tile_last, lds_addr(tiles), lds_addr(tiles + DF_TILE_BYTES), wave, lane, k, true, tile_slot0);
        }
    }
#undef DF_PIECE
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float sres = df_wave_sum<USE_DPP>(acc[i]);
        if constexpr (W8) sres *= a.sqkv[prow[i]];                  // per-row dequantisation scale
        if (lane == 0) red[16 + 3 * wave + i] = sres;
    }
    DF_MARK(4);
    __syncthreads();

    // ---- RoPE, KV append, publish q/k/v: 12 threads, one (even, odd) row pair each ---------------------------------------
    u64 *gq = a.gq + (size_t)g * DF_GQ;
    if (tid < 12) {
        const int p = tid;
        float e0 = red[16 + 2 * p], e1 = red[16 + 2 * p + 1];
        if (p < 10) {          // q pairs 0..7, k pairs 8..9: interleaved-pair RoPE at this position (voxtral_kernels.c:502-526)
            const float r0 = e0 * rope_c - e1 * rope_s, r1 = e0 * rope_s + e1 * rope_c;
            e0 = r0; e1 = r1;
        }
        if (p < 8) {
            df_store_granule(gq + 16 * j + 2 * p, epoch, e0); df_store_granule(gq + 16 * j + 2 * p + 1, epoch, e1);
        } else {
            const bool isk = p < 10;
            const int kl = 4 * j + 2 * (isk ? p - 8 : p - 10);
            const size_t slot = (size_t)(pos % a.kv_cap) * DF_DKV + DF_HD * g + kl;
            float *ring = isk ? a.kring : a.vring;
            ring[slot] = e0; ring[slot + 1] = e1;                                   // for the following steps
            const int gi = DF_NQ + (isk ? 0 : DF_HD) + kl;
            df_store_granule(gq + gi, epoch, e0); df_store_granule(gq + gi + 1, epoch, e1);
        }
    }
    __syncthreads();                       // xs / nw are dead from here on: 24 KB of scratch for the attention stage
    DF_MARK(5);
    // L2 prefetch for the next launch (see DfPrefetch): only in the short-context regime (members without Wo rows, one XCD per group)
    const bool pf_on = a.pf.units > 0 && wo_light;
    const int pf_V = 32 * a.pf.units, pf_Vm = min(a.pf.member_units, pf_V);
    const unsigned pf_lds = lds_addr(tiles) + 65536u + (unsigned)wave * 1024u;      // beyond the Wo reduction scratch; the tiles are dead where this is used
    if (pf_on && !att_block && a.pf.when == 3) {      // A/B: in front of the Wo rows
        df_prefetch_units(a.pf, g, pf_Vm, pf_V, (j - ns) * DF_WAVES + wave, (DF_BPG - ns) * DF_WAVES, lane, pf_lds);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!att_block) {
        DF_LATE_WO(0) DF_LATE_WO(4) DF_LATE_WO(8) DF_LATE_WO(12)
    }
    if (pf_on && !att_block && a.pf.when == 1) {
        __builtin_amdgcn_sched_barrier(0);
        df_prefetch_units(a.pf, g, pf_Vm, pf_V, (j - ns) * DF_WAVES + wave, (DF_BPG - ns) * DF_WAVES, lane, pf_lds);
    }

    float *qs = xs;                        // [512] the group's q
    float *kvn = xs + 512;                 // [256] this step's k | v of head g
    float *att = xs + 768;                 // [512] merged attention output of the group's 4 heads
    float *pt = xs + 1792;                 // [4 heads][64 keys] softmax numerators of the current tile
    float *cr = xs + 2048;                 // [4] rescale of the running output, [4] running max, [4] running sum
    float *sc8 = xs + 2560;                // [4 heads][8 dim slices][64 keys] partial scores, then [4 key quarters][4 heads][128] partial outputs

    // ---- hand-off 1 (attention members only): sweep the group's 768 granules ------------------------------------------------
    // The granule loads go out here; their tags are looked at INSIDE the tile loop's first iteration: everything the compiler hoists in
    // front of that loop (a few hundred address computations - 0.6 us between "sweep done" and "first tile in" in the round-4 timelines)
    // then runs while the loads are in flight instead of after them, on the path every workgroup of the group waits for.
    u64 gv1[2] = {0, 0};
    if (att_block) {
        gv1[0] = df_load_granule(gq + tid);
        if (tid < 256) gv1[1] = df_load_granule(gq + 512 + tid);
    } else {
        __syncthreads();
        DF_MARK(6);
    }

    // ---- attention over this workgroup's key slice, 64-key tiles through LDS -------------------------------------------
    //   scores: wave -> (head, half of the 128 dims), lane -> key: a 64-term dot per thread, halves added through LDS;
    //   softmax: online over tiles, one wave per head reduces max / sum over its 64 keys;
    //   PV: thread -> (head, dim), walks the tile's keys.  Arithmetic of voxtral_kernels.c:412-482 up to summation order.
    u64 *gp = a.gp + ((size_t)g * DF_BPG) * DF_GP;
    if (att_block) {
        const int ho = tid >> 7, dd = tid & 127;         // PV phase
        float o_acc = 0.f;
        if (tid < 4) { cr[4 + tid] = -1e30f; cr[8 + tid] = 0.f; }
        for (int ti = 0; ti < n_tiles; ti++) {
            unsigned char *kt = tiles + (ti & 1) * 2 * DF_TILE_BYTES, *vt = kt + DF_TILE_BYTES;
            const int t0 = s_lo + ti * DF_TILE;
            if (ti == 0) {      // hand-off 1 completes here (see above)
                qs[tid] = (unsigned)(gv1[0] >> 32) == epoch ? __uint_as_float((unsigned)gv1[0]) : df_wait_granule(gq + tid, epoch, a, 1u);
                if (tid < 256)
                    kvn[tid] = (unsigned)(gv1[1] >> 32) == epoch ? __uint_as_float((unsigned)gv1[1]) : df_wait_granule(gq + 512 + tid, epoch, a, 1u);
                DF_MARK(6);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this tile (and everything older) has landed
            __syncthreads();
            if (ti + 1 < n_tiles)                                    // next tile into the other buffer, under this tile's math
                df_tile_dma(a, g, t0 + DF_TILE, s_hi, lds_addr(tiles + ((ti + 1) & 1) * 2 * DF_TILE_BYTES),
                            lds_addr(tiles + ((ti + 1) & 1) * 2 * DF_TILE_BYTES + DF_TILE_BYTES), wave, lane);
            if (t0 + DF_TILE > pos && t0 <= pos) {
                // this step's own K/V row is not visible in the ring to other CUs yet: patch it in from the hand-off
                const int key = pos - t0;
                if (tid < 32) *reinterpret_cast<float4 *>(kt + key * 512 + ((tid ^ (key & 31)) << 4)) = *reinterpret_cast<const float4 *>(kvn + tid * 4);
                else if (tid < 64) *reinterpret_cast<float4 *>(vt + key * 512 + ((tid - 32) << 4)) = *reinterpret_cast<const float4 *>(kvn + DF_HD + (tid - 32) * 4);
                __syncthreads();
            }
            if (ti == 0) DF_MARK(11);
            // Round 3: K and V rows are read from LDS ONCE for the 4 query heads that share them (they were read once per head:
            // 2 x 128 KB of LDS traffic per tile, ~0.9 us of the 2.4 us a member spent between its sweep and its partial).
            //   scores: wave -> 16 of the 128 dims for all 4 heads, lane -> key; the 8 partial sums per (head, key) meet in LDS;
            //   PV: thread -> (quarter of the tile's keys, dim) for all 4 heads; the 4 quarter sums meet in LDS.
            {
                const unsigned char *krow = kt + lane * 512;
                float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int ch = 4 * wave + c;
                    const float4 kv4 = *reinterpret_cast<const float4 *>(krow + ((ch ^ (lane & 31)) << 4));
#pragma unroll
                    for (int hh = 0; hh < 4; hh++) {
                        const float4 q4 = *reinterpret_cast<const float4 *>(qs + hh * DF_HD + ch * 4);
                        s4[hh] = fmaf(q4.x, kv4.x, s4[hh]); s4[hh] = fmaf(q4.y, kv4.y, s4[hh]);
                        s4[hh] = fmaf(q4.z, kv4.z, s4[hh]); s4[hh] = fmaf(q4.w, kv4.w, s4[hh]);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 4; hh++) sc8[(hh * 8 + wave) * 64 + lane] = s4[hh];
            }
            __syncthreads();
            if (ti == 0) DF_MARK(12);
            if (wave < 4) {   // one wave per head: online softmax over this tile's keys
                float s = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; w8++) s += sc8[(wave * 8 + w8) * 64 + lane];
                s *= a.scale;
                if (t0 + lane > s_hi) s = -INFINITY;
                const float m_old = cr[4 + wave], l_old = cr[8 + wave];
                const float m_new = fmaxf(m_old, df_wave_max<USE_DPP>(s));
                const float p = expf(s - m_new);
                const float corr = expf(m_old - m_new);
                const float l_new = l_old * corr + df_wave_sum<USE_DPP>(p);
                pt[wave * 64 + lane] = p;
                if (lane == 0) { cr[wave] = corr; cr[4 + wave] = m_new; cr[8 + wave] = l_new; }
            }
            __syncthreads();
            {   // quarter sums: keys 16 kq .. 16 kq + 15 of the tile, dim dd, all 4 heads
                // only the tile's valid keys: the rows past s_hi hold whatever the ring slot had (0 x NaN would poison the sum)
                const int kq = tid >> 7;
                const float *vcol = reinterpret_cast<const float *>(vt) + dd;
                const int nv = min(DF_TILE, s_hi - t0 + 1);
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k0 = 16 * kq + 4 * i;
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = k0 + u < nv ? vcol[(k0 + u) * 128] : 0.f;
#pragma unroll
                    for (int hh = 0; hh < 4; hh++) {
                        const float4 p4 = *reinterpret_cast<const float4 *>(pt + hh * 64 + k0);
                        a4[hh] = fmaf(p4.x, v[0], a4[hh]); a4[hh] = fmaf(p4.y, v[1], a4[hh]);
                        a4[hh] = fmaf(p4.z, v[2], a4[hh]); a4[hh] = fmaf(p4.w, v[3], a4[hh]);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 4; hh++) sc8[(kq * 4 + hh) * DF_HD + dd] = a4[hh];     // the score area is free again
            }
            __syncthreads();
            {
                const float accv = (sc8[(0 * 4 + ho) * DF_HD + dd] + sc8[(1 * 4 + ho) * DF_HD + dd]) +
                                   (sc8[(2 * 4 + ho) * DF_HD + dd] + sc8[(3 * 4 + ho) * DF_HD + dd]);
                o_acc = o_acc * cr[ho] + accv;
            }
        }
        __syncthreads();
        {   // this slice's partial, published as granules: thread -> (head, dim)
            u64 *mine = gp + (size_t)j * DF_GP;
            df_store_granule(mine + ho * DF_HD + dd, epoch, o_acc);
            if (dd == 0) { df_store_granule(mine + 4 * DF_HD + 2 * ho, epoch, cr[4 + ho]); df_store_granule(mine + 4 * DF_HD + 2 * ho + 1, epoch, cr[8 + ho]); }
        }
        // Long contexts (members carry Wo rows too): the rows are requested only now, BEHIND the partial.  A CU's stores leave through
        // the same queue as its loads: published behind 96 KB of queued rows (rounds 2 - 3: they were requested under the first tile's
        // math) the partial became visible to the group only when they had landed, and their issue stalled the score phase by 1.3 us.
        // Same box, alternating, ms per step at 600 / 1000 / 1900 / 3800 / 8000 keys: 1.4331 / 1.4490 / 1.5265 / 1.6296 / 1.8045 ->
        // 1.4289 / 1.4404 / 1.5141 / 1.5882 / 1.7540 (requesting them right behind the q/k/v sweep instead: +2 % up to 1900 keys - a
        // sweep that finds a stale tag repeats its load behind the rows - and -1 % beyond).
        if (!wo_light) { __builtin_amdgcn_sched_barrier(0); DF_LATE_WO(0) DF_LATE_WO(4) DF_LATE_WO(8) }
        if (pf_on && pf_Vm > 0) {      // a member is done: its share of the next launch's first bytes (every tile read is behind the barrier above)
            __builtin_amdgcn_sched_barrier(0);
            df_prefetch_units(a.pf, g, 0, pf_Vm, j * DF_WAVES + wave, ns * DF_WAVES, lane, pf_lds);
        }
    }
#undef DF_LATE_WO
#undef DF_WO_PTR
#undef DF_WO_SCALE
    DF_MARK(7);

    // ---- hand-off 2: sweep the group's partials and merge them in slice order (thread -> head, dim) ---------------------
    // (an attention member without Wo rows is done: it only has to let its loads drain)
    if (!(wo_light && att_block)) {
        const int h = tid >> 7, d = tid & 127;
        float M = -1e30f, L = 0.f, O = 0.f;
        // (Round 4, built, measured and removed: the sweep queued ONCE behind the Wo rows / prefetch units, so that it is back when
        //  they have landed - about when the partials exist - and the poll, its barrier and the second trip to L2 can be skipped:
        //  +22 .. +50 us per step at every prefetch depth, gpurun_out/r4l.  224 workgroups x 512 threads x 12 granule loads in
        //  flight while the members run their attention delay the members: attention end 11.2 -> 12.1 us, everybody's sweep 13.4 -> 15.4.)
        // 28 of a group's 32 workgroups arrive here long before the partials exist.  ONE thread per workgroup polls (one granule
        // of the last slice); the other 511 sleep at the barrier: with every thread re-reading its granules the pollers took so
        // much of the fabric that the attention members - whom they are waiting for - finished 2 us later (measured).
        if (tid == 0) (void)df_wait_granule(gp + (size_t)(ns - 1) * DF_GP + 4 * DF_HD, epoch, a, 2u);
        __syncthreads();
        if (ns > 8) {
            // Many slices (long contexts): three round trips instead of one per 4 slices (the 30 slices of a 1900-key window took
            // 8 of them, 6.5 - 8.7 us per layer, timeline).  First the (max, sum) pairs - 8 per slice, one thread each - into
            // LDS and from them every slice's weight exp(m_s - M) / L per head; then every thread fetches its (head, dim) of
            // all slices, 16 in flight, and adds them up in slice order.
            float *mlv = xs + 2080;                    // [32 slices][4 heads][2]
            float *scl = xs + 2080 + 256;              // [4 heads][32 slices]
            // Round 3: the thread's own (head, dim) granules of all slices are requested BEFORE the (max, sum) pairs are waited
            // for - loads return in order, so the two fetches share one trip to L2 instead of following each other (the merge
            // took 4.6 us at 1900 keys, profiles/r02_fuse_timeline_kv1900.txt).  VOX_HIP_FUSE_MERGE3 = the old three trips.
            // (one shared trip pays while a member's slice is one 64-key tile - 1.576 vs 1.589 ms per step at 1900 keys; with two tiles
            //  per member the members finish further apart, the early loads miss and are repeated: 1.731 vs 1.710 at 3800)
            const bool three_trips = a.split_keys > 64;
            u64 gv[32];
            if (!three_trips) {
#pragma unroll
                for (int u = 0; u < 32; u++) gv[u] = df_load_granule(gp + (size_t)min(u, ns - 1) * DF_GP + h * DF_HD + d);
            }
            if (tid < ns * 8) {
                const u64 *src = gp + (size_t)(tid >> 3) * DF_GP + 4 * DF_HD + (tid & 7);
                mlv[tid] = df_wait_granule(src, epoch, a, 2u);
            }
            __syncthreads();
            if (tid < 128) {       // thread -> (head = tid >> 5, slice = tid & 31): max and sum over the 32-lane segment
                const int hh = tid >> 5, s1 = tid & 31;
                const float ms = s1 < ns ? mlv[(s1 * 4 + hh) * 2] : -1e30f, lsum = s1 < ns ? mlv[(s1 * 4 + hh) * 2 + 1] : 0.f;
                float Mx = ms;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) Mx = fmaxf(Mx, __shfl_xor(Mx, o, 32));
                const float e = expf(ms - Mx);
                float Ls = lsum * e;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) Ls += __shfl_xor(Ls, o, 32);
                scl[hh * 32 + s1] = Ls > 0.f ? e / Ls : 0.f;
            }
            __syncthreads();
            {
                DfSpin sp; df_spin_begin(sp);
                for (unsigned it = 0;; it++) {
                    bool ok = true;
                    if (it > 0 || three_trips) {
#pragma unroll
                        for (int u = 0; u < 32; u++) gv[u] = df_load_granule(gp + (size_t)min(u, ns - 1) * DF_GP + h * DF_HD + d);
                    }
#pragma unroll
                    for (int u = 0; u < 32; u++) ok = ok && (unsigned)(gv[u] >> 32) == epoch;
                    if (ok) break;
                    if ((it & 15u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    if (df_spin_expired(sp, a.err, a.spin_limit, 2u, epoch)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
#pragma unroll
                for (int u = 0; u < 32; u++)
                    if (u < ns) O = fmaf(scl[h * 32 + u], __uint_as_float((unsigned)gv[u]), O);
            }
            L = 1.0f;
        } else
        for (int s0 = 0; s0 < ns; s0 += 4) {           // 4 slices = 12 loads in flight per thread, re-issued together until every tag matches
            u64 gv[4][3];
            DfSpin sp; df_spin_begin(sp);
            for (unsigned it = 0;; it++) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const u64 *part = gp + (size_t)min(s0 + u, ns - 1) * DF_GP;
                    gv[u][0] = df_load_granule(part + h * DF_HD + d);
                    gv[u][1] = df_load_granule(part + 4 * DF_HD + 2 * h);
                    gv[u][2] = df_load_granule(part + 4 * DF_HD + 2 * h + 1);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int k = 0; k < 3; k++) ok = ok && (unsigned)(gv[u][k] >> 32) == epoch;
                if (ok) break;
                // bounded like df_wait_granule: a timeout (or an earlier one of any launch) gives up and flags the batch
                if ((it & 15u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (df_spin_expired(sp, a.err, a.spin_limit, 2u, epoch)) break;
                __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (s0 + u >= ns) break;
                const float v0 = __uint_as_float((unsigned)gv[u][0]), vm = __uint_as_float((unsigned)gv[u][1]), vl = __uint_as_float((unsigned)gv[u][2]);
                const float mn = fmaxf(M, vm);
                const float c0 = expf(M - mn), c1 = expf(vm - mn);
                L = L * c0 + vl * c1;
                O = O * c0 + v0 * c1;
                M = mn;
            }
        }
        att[h * DF_HD + d] = L > 0.f ? O / L : 0.f;
    }
    DF_MARK(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the Wo slice has landed in its registers
    __syncthreads();
    DF_MARK(9);
    if (pf_on && !att_block && a.pf.when == 2) {      // in flight under the Wo product; the wave ends when they have landed
        df_prefetch_units(a.pf, g, pf_Vm, pf_V, (j - ns) * DF_WAVES + wave, (DF_BPG - ns) * DF_WAVES, lane, pf_lds);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- this wave's rows of  Wo[:, 512 g .. 512 g + 511] . att ------------------------------------------------------------------
    if constexpr (W8) {
        if (wo_n > 0) {
            // lane l holds 16 weights of row wo_row0 + 2 i + (l >> 5), columns 16 (l & 31) .. + 15 of the group's 512
            const float *xp = att + (lane & 31) * 16;
            const float4 x0 = *reinterpret_cast<const float4 *>(xp), x1 = *reinterpret_cast<const float4 *>(xp + 4);
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8), x3 = *reinterpret_cast<const float4 *>(xp + 12);
            float sums[NWO];
#pragma unroll
            for (int i = 0; i < NWO; i++) sums[i] = row16_sum<USE_DPP>(dot16_fp8(wv[i], x0, x1, x2, x3, 0.f));
#pragma unroll
            for (int i = 0; i < NWO; i++) {
                // the other 16-lane row of this 32-lane half: a DPP row move within the same SIMD row pair is not available, so one
                // cross-row exchange per load (all 8 issued back to back, one wait)
                const float o = __shfl_xor(sums[i], 16, 64);
                const int row = wo_row0 + 2 * i + (lane >> 5);
                if ((lane & 31) == 0 && row < wo_row0 + wo_n) a.wo_part[(size_t)g * DF_D + row] = (sums[i] + o) * wo_sc[i];
            }
        }
    } else
    if (wo_n > 0) {
        const float4 x0 = *reinterpret_cast<const float4 *>(att + lane * 8);
        const float4 x1 = *reinterpret_cast<const float4 *>(att + lane * 8 + 4);
        // The 16 row sums of a wave as ONE transposed reduction through LDS (the K/V tile area is dead by now: every DMA has
        // landed and every reader is past the barrier above): lane l parks its 16 partial dots in column l of a [16][68]
        // scratch (row stride 68 floats: the reads below then hit 16 distinct 16-byte slots per service group), lane
        // (row = l >> 2, quarter = l & 3) adds 16 of them in a fixed order, two quad steps finish the row.  LDS operations
        // of one wave execute in order, so no barrier is needed between the stores and the loads.
        float *wr = reinterpret_cast<float *>(tiles) + wave * (16 * 68);
#pragma unroll
        for (int i = 0; i < 16; i++) wr[i * 68 + lane] = dot8_bf16(wv[i], x0, x1, 0.f);      // rows past wo_n repeat the last one: never stored
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int row = lane >> 2, q = lane & 3;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 v = *reinterpret_cast<const float4 *>(wr + row * 68 + q * 16 + 4 * c);
            s += (v.x + v.y) + (v.z + v.w);
        }
        s += __shfl_xor(s, 1, 4);
        s += __shfl_xor(s, 2, 4);
        if (q == 0 && row < wo_n) a.wo_part[(size_t)g * DF_D + wo_row0 + row] = s;
    }
    if (pf_on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the scratch slot is this workgroup's LDS: no DMA may outlive it
    DF_MARK(10);
    tl_end(a.tl, tl0, df_stamp, 13);
}

// ---------------------------------------------------------------------------------------------------------
// k_gemv_w13x — the FFN up-projection of a decoder step behind k_dec_attn_fused:
//     x' = x + sum_g wo_part[g]  (fixed order)  ->  ffn_norm, ada scaling  ->  silu(W1 x) * (W3 x)
// (voxtral_decoder.c:676-687).  Same memory-queue discipline as k_gemv3, but ONE 768-thread workgroup per
// CU (12 waves, 3 row pairs of W1/W3 each: 36 x 16 B per lane in flight) instead of three 256-thread ones,
// so that the nine 12 KB vectors of the prologue are staged once per CU (LDS-DMA, 132 KB).  Block 0 writes
// x' back (the W2 launch reads it as its residual).
// ---------------------------------------------------------------------------------------------------------
struct W13xArgs {
    const uint16_t *w1, *w3;   // [9216][3072] each (W8: fp8 e4m3 bytes, one f32 scale per row in s1 / s3)
    const float *s1, *s3;
    const float *x;            // [3072]
    const float *wo_part;      // [8][3072]
    const float *norm_w, *ada; // [3072]
    float eps;
    float *x_out;              // [3072] x' (may alias x)
    float *h;                  // [9216]
    unsigned long long *tl;    // optional (tuning): per-workgroup timeline
};
constexpr int W13X_THREADS = 768;
constexpr int W13X_LDS_BYTES = (9 + 2 + 1) * DF_D * 4 + 256;

// (Round 4, measured and removed: issuing round 1 of the weight stream earlier - right behind the prologue barrier or together with
// round 0 - and an L2 prefetch of the W2 launch's first bytes by waves that finish early: no gain, profiles/NOTES.md.)
template <bool W8>
__global__ __launch_bounds__(W13X_THREADS, 1) void k_gemv_w13x(const W13xArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *stage = smem;                 // [9][3072]: x, wo_part[0..7]
    float *nws = smem + 9 * DF_D;        // [3072] norm weights
    float *ads = nws + DF_D;             // [3072] ada
    float *xs = ads + DF_D;              // [3072]
    float *red = xs + DF_D;              // [16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wofs = (unsigned)wave * 1024u;        // 12 waves x 1 KiB = one 12 KB vector per DMA round
    unsigned long long df_stamp[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tl0 = tl_begin(a.tl);
    DF_MARK(0);

    glds16(a.x + tid * 4, lds_addr(stage) + wofs);
#pragma unroll
    for (int gi = 0; gi < 8; gi++) glds16(a.wo_part + (size_t)gi * DF_D + tid * 4, lds_addr(stage + (gi + 1) * DF_D) + wofs);
    glds16(a.norm_w + tid * 4, lds_addr(nws) + wofs);
    glds16(a.ada + tid * 4, lds_addr(ads) + wofs);
    __builtin_amdgcn_sched_barrier(0);
    // Weight stream in three rounds of two 1 KiB pieces per row.  A CU's memory queue holds far less than this kernel's
    // 574 KB per CU, and a wave that issues more than fits waits IN the issue stage: with everything issued up front the
    // slowest of the 12 waves reached the prologue barrier only after ~15 us (measured), so the sum / RMSNorm and all dot
    // products ran after the stream instead of under it.  Round 0 (144 KB per CU) is issued before the prologue and covers
    // it; rounds 1 and 2 are issued as the previous round's dot products retire.
    // bf16: 6 pieces of 1 KiB per row (8 weights per lane and piece); fp8: 3 pieces (16 weights per lane and piece)
    constexpr int NP = W8 ? 3 : 6, EPP = W8 ? 16 : 8;
    uint4 w[2][3][NP];
    const int pair0 = blockIdx.x * 36 + wave * 3;
    const uint4 *p1[3], *p3[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        constexpr size_t ROWB = W8 ? DF_D : 2 * DF_D;
        p1[r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.w1) + (size_t)(pair0 + r) * ROWB) + lane;
        p3[r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.w3) + (size_t)(pair0 + r) * ROWB) + lane;
    }
#define W13_ISSUE(C)                                                                    \
    if constexpr ((C) < NP) { _Pragma("unroll") for (int r = 0; r < 3; r++) w[0][r][(C) < NP ? (C) : 0] = ld_stream(p1[r] + (C) * 64); \
      _Pragma("unroll") for (int r = 0; r < 3; r++) w[1][r][(C) < NP ? (C) : 0] = ld_stream(p3[r] + (C) * 64); }
    W13_ISSUE(0) W13_ISSUE(1)
    __builtin_amdgcn_sched_barrier(0);
    DF_MARK(1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // the 11 DMAs are in; round 0 of the weights still streams
    __syncthreads();
    DF_MARK(2);
    {
        float4 v = *reinterpret_cast<const float4 *>(stage + tid * 4);
#pragma unroll
        for (int gi = 1; gi <= 8; gi++) {
            const float4 p = *reinterpret_cast<const float4 *>(stage + gi * DF_D + tid * 4);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (blockIdx.x == 0) *reinterpret_cast<float4 *>(a.x_out + tid * 4) = v;
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        ss = df_wave_sum<true>(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 12; i++) tot += red[i];
        const float inv = 1.0f / sqrtf(tot / (float)DF_D + a.eps);
        const float4 gw = *reinterpret_cast<const float4 *>(nws + tid * 4);
        const float4 sc = *reinterpret_cast<const float4 *>(ads + tid * 4);
        v.x = v.x * inv * gw.x * (1.0f + sc.x); v.y = v.y * inv * gw.y * (1.0f + sc.y);
        v.z = v.z * inv * gw.z * (1.0f + sc.z); v.w = v.w * inv * gw.w * (1.0f + sc.w);
        *reinterpret_cast<float4 *>(xs + tid * 4) = v;
        __syncthreads();
    }
    DF_MARK(3);
    float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#define W13_DOT(C)                                                                      \
    if constexpr ((C) < NP) {                                                           \
        constexpr int CC = (C) < NP ? (C) : 0;                                          \
        const float *xp = xs + (CC * 64 + lane) * EPP;                                  \
        const float4 x0 = *reinterpret_cast<const float4 *>(xp);                        \
        const float4 x1 = *reinterpret_cast<const float4 *>(xp + 4);                    \
        if constexpr (W8) {                                                             \
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8);                \
            const float4 x3 = *reinterpret_cast<const float4 *>(xp + 12);               \
            _Pragma("unroll") for (int m = 0; m < 2; m++)                               \
                _Pragma("unroll") for (int r = 0; r < 3; r++) acc[m][r] = dot16_fp8(w[m][r][CC], x0, x1, x2, x3, acc[m][r]); \
        } else {                                                                        \
            _Pragma("unroll") for (int m = 0; m < 2; m++)                               \
                _Pragma("unroll") for (int r = 0; r < 3; r++) acc[m][r] = dot8_bf16(w[m][r][CC], x0, x1, acc[m][r]); \
        }                                                                               \
    }
    W13_ISSUE(2) W13_ISSUE(3)
    __builtin_amdgcn_sched_barrier(0);
    W13_DOT(0) W13_DOT(1)
    __builtin_amdgcn_sched_barrier(0);
    W13_ISSUE(4) W13_ISSUE(5)
    __builtin_amdgcn_sched_barrier(0);
    W13_DOT(2) W13_DOT(3)
    __builtin_amdgcn_sched_barrier(0);
    W13_DOT(4) W13_DOT(5)
#undef W13_ISSUE
#undef W13_DOT
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 3; r++) acc[m][r] = df_wave_sum<true>(acc[m][r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float g = acc[0][r], u = acc[1][r];
            if constexpr (W8) { g *= a.s1[pair0 + r]; u *= a.s3[pair0 + r]; }          // per-row dequantisation scale
            a.h[pair0 + r] = silu(g) * u;                                               // voxtral_decoder.c:684-687
        }
    }
    DF_MARK(4);
    tl_end(a.tl, tl0, df_stamp, 5);
}


// ---------------------------------------------------------------------------------------------------------
// k_gemv_w2x - the FFN down-projection of a decoder step, same shape as k_gemv_w13x:
//     x'' = x' + W2 h        (voxtral_decoder.c:688-690)
// ONE 768-thread workgroup per CU, wave w streams row 12 b + w of W2 (18 KB = 18 x 16 B per lane) in three staged rounds,
// h (36 KB) is staged once per CU by LDS-DMA.  (k_gemv3 ran this as 512 four-wave workgroups, two per CU, each staging
// its own copy of h and splitting every row between two waves: 5.5 TB/s against this layout's 6.3 in k_gemv_w13x.)
// ---------------------------------------------------------------------------------------------------------
struct W2xArgs {
    const uint16_t *w2;        // [3072][9216] (W8: fp8 e4m3 bytes, one f32 scale per row in s2)
    const float *s2;
    const float *h;            // [9216]
    float *x;                  // [3072] residual stream, updated in place (every row is read and written by its one wave)
    unsigned long long *tl;    // optional (tuning): per-workgroup timeline
};
constexpr int W2X_THREADS = 768, W2X_K = 9216;
constexpr int W2X_LDS_BYTES = W2X_K * 4 + 64;

// gx (optional): x'' also leaves as {epoch, value} granules (k_w2x_attn12: the attention block of the next layer in the same launch)
template <bool W8>
__device__ __forceinline__ void w2x_body(const W2xArgs &a, float *smem, u64 *gx, unsigned gx_epoch) {
    float *hs = smem;                    // [9216]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wofs = (unsigned)wave * 1024u;
    const unsigned long long tl0 = tl_begin(a.tl);
    const int row = blockIdx.x * 12 + wave;
    const float resid = a.x[row];
#pragma unroll
    for (int p = 0; p < 3; p++) glds16(a.h + p * 3072 + tid * 4, lds_addr(hs + p * 3072) + wofs);
    __builtin_amdgcn_sched_barrier(0);
    // bf16: 18 pieces of 1 KiB per row in rounds of 6; fp8: 9 pieces in rounds of 3
    constexpr int NP = W8 ? 9 : 18, RND = NP / 3, EPP = W8 ? 16 : 8;
    uint4 w[NP];
    const uint4 *wp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.w2) + (size_t)row * (W8 ? W2X_K : 2 * W2X_K)) + lane;
#define W2_ISSUE(R) { _Pragma("unroll") for (int c = RND * (R); c < RND * (R) + RND; c++) w[c] = ld_stream(wp + c * 64); }
#define W2_DOT(R)                                                                                  \
    { _Pragma("unroll") for (int c = RND * (R); c < RND * (R) + RND; c++) {                         \
        const float *xp = hs + (c * 64 + lane) * EPP;                                               \
        const float4 x0 = *reinterpret_cast<const float4 *>(xp);                                    \
        const float4 x1 = *reinterpret_cast<const float4 *>(xp + 4);                                \
        if constexpr (W8) {                                                                         \
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8);                            \
            const float4 x3 = *reinterpret_cast<const float4 *>(xp + 12);                           \
            acc = dot16_fp8(w[c], x0, x1, x2, x3, acc);                                             \
        } else {                                                                                    \
            acc = dot8_bf16(w[c], x0, x1, acc);                                                     \
        } } }
    W2_ISSUE(0)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RND) : "memory");      // the residual load and the 3 DMAs are in; round 0 still streams
    __syncthreads();
    float acc = 0.f;
    W2_ISSUE(1)
    __builtin_amdgcn_sched_barrier(0);
    W2_DOT(0)
    __builtin_amdgcn_sched_barrier(0);
    W2_ISSUE(2)
    __builtin_amdgcn_sched_barrier(0);
    W2_DOT(1)
    __builtin_amdgcn_sched_barrier(0);
    W2_DOT(2)
#undef W2_ISSUE
#undef W2_DOT
    acc = df_wave_sum<true>(acc);
    if constexpr (W8) acc *= a.s2[row];
    if (lane == 0) {
        a.x[row] = resid + acc;
        if (gx) df_store_granule(gx + row, gx_epoch, resid + acc);
    }
    tl_end(a.tl, tl0);
}
template <bool W8>
__global__ __launch_bounds__(W2X_THREADS, 1) void k_gemv_w2x(const W2xArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    w2x_body<W8>(a, smem, nullptr, 0u);
}

// ---------------------------------------------------------------------------------------------------------
// k_ffn_fused (round 4) - the whole FFN block of a decoder step as ONE launch (bf16 weights):
//     x' = x + sum_g wo_part[g] -> ffn_norm, ada -> h = silu(W1 x) * (W3 x)   |hand-off|   x'' = x' + W2 h
// (voxtral_decoder.c:676-690; replaces k_gemv_w13x + k_gemv_w2x: 20.7 + 10.7 us per layer, of which one kernel boundary, one
// start-up ramp and one drain - ~3 us - move no weight byte).
//
// Why this hand-off can be free where the others were not (a hand-off under a full-rate stream costs what a kernel
// boundary costs, because a load issued into a saturated memory system returns behind everything queued before it): here that
// property is used, not fought.  A wave that has finished its three row pairs of W1 / W3 publishes its three h values as
// {epoch, value} granules (one write-through store each).  Once all 12 waves of the workgroup are there, every wave queues its
// 18 KB of W2 (registers: the W1 / W3 registers are dead) and, BEHIND them, its 12 granule loads per lane of the h sweep.  They
// come back when the W2 bytes have landed, ~7 us later, and every producer on the chip published its h long before that (the
// workgroups of a launch finish their first phase within ~3 us of each other): the sweep normally needs no retry, and its round
// trip is hidden behind bytes that had to be streamed anyway.
//
// The W2 stage is K-SLICED over the waves: wave w does not take ONE row of the workgroup's 12 but columns 768 w .. 768 w + 767 of
// ALL of them.  Its 12 x 768 weights are 18 loads of 64 lanes x 8: load 3m = row 2m, columns 0..511 of the slice; load 3m+1 =
// row 2m, columns 512..767 (lanes 0-31) and row 2m+1, columns 0..255 (lanes 32-63); load 3m+2 = row 2m+1, columns 256..767.  The
// 768 h values a wave needs are exactly the 12 granules per lane of its share of the sweep: they go through a 3 KB LDS slice only
// this wave touches (no workgroup barrier), and a lane reads 24 of them once for all rows.  (With one row per wave every wave
// read the whole h vector from LDS - 12 x 36 KB per workgroup, 1.6 of the 1.8 us that used to follow the sweep.)  What remains
// exposed after the stream: tags checked, 18 x 8 FMAs per lane, 12 DPP row sums, one barrier, 12 stores per workgroup.
//
// Hand-off schedules that were built and measured (ms per step at 232 / 1900 keys, one box, alternating runs; two launches:
// 1.3872 / 1.5689; this one, "B": 1.3662 / 1.5539 before the K-slicing):
//   A  early waves queue their W2 bytes at once, workgroup barrier, sweep: 1.4436 / 1.6373 - the early W2 loads delay the late
//      waves' last W1 / W3 pieces, and the sweep behind the barrier has nothing left to hide behind: 5.6 us, which is what a
//      72 KB all-to-all granule sweep costs on this chip;
//   B' B with the sweep behind the first two thirds of the W2 bytes: no gain;
//   C  no barrier, wave w sweeps the granules of the waves w of all workgroups right behind its own W2 bytes (1.82 / 2.01: early
//      sweeps find stale tags, every retry queues behind the CU's whole stream and 3072 polling waves take bandwidth from it);
//   E  the sweep IN FRONT of the W2 bytes so that the dot products run piece by piece under the stream (1.3988 / 1.5908: the sweep
//      still takes its ~6 us, now in front of everything, and slows the loads it shares the queue with); D+E: 1.4165 / 1.6006;
//   fp8 weights: with half the W2 bytes the sweep is no longer hidden (1.1055 against 1.0271 ms on two launches): fp8 keeps
//      k_gemv_w13x<true> + k_gemv_w2x<true>;
//   "projection ahead": the FOLLOWING layer's attention_norm -> wq / wk / wv -> RoPE at the end of this launch (+0.18 ms per step:
//      a CU's stores leave through the same queue as its loads, so the x'' hand-off published behind the projection rows was
//      visible only when they had landed; profiles/r04_qkv_ahead_timeline.txt, the code is profiles/r04_qkv_ahead.patch).
//
// Needs its 256 workgroups co-resident (one per CU) like k_dec_attn_fused: every wait is bounded, a time-out sets *err and the
// host re-runs the batch on the launch-per-GEMV chain.  The epoch is the launch counter of the layer's attention launch.
// ---------------------------------------------------------------------------------------------------------
struct FfnArgs {
    const uint16_t *w1, *w3;   // [9216][3072] bf16 each
    const uint16_t *w2;        // [3072][9216] bf16
    const float *x;            // [3072] residual stream before the attention block's output is added
    const float *wo_part;      // [8][3072]
    const float *norm_w, *ada; // [3072]
    float eps;
    float *x_out;              // [3072] x'' (the other residual buffer: nobody reads it during this launch)
    float *xprime_out;         // optional [3072]: x' (debug taps), written by block 0
    u64 *gh;                   // [9216] hand-off granules
    unsigned epoch;
    unsigned *err;
    unsigned long long spin_limit;
    unsigned long long *tl;    // optional (tuning): per-workgroup timeline
};
constexpr int FFN_THREADS = 768, FFN_H = 9216;
constexpr int FFN_LDS_BYTES = W13X_LDS_BYTES;       // phase 2 reuses the prologue's staging area: x' (12 KB) + h (36 KB)

// XP (round 5, k_dec_stack): x' arrives by granules instead of through memory after a kernel boundary, in two hops.  The Wo partial
// sums of this layer's attention block leave as granules (gw[8][3072], tagged w_epoch); the workgroup that owns rows 12 b .. 12 b + 11
// of the residual stream (the rows whose x'' it produced in the previous FFN block) fetches their 8 partials and its own x'' values,
// adds them in the order the memory prologue uses (x, then groups 0 .. 7) and publishes 12 x' granules; everybody sweeps the 3072 x'
// granules.  Both trips are queued BEHIND weight bytes that have to be streamed anyway (the first behind round 0 of W1 / W3, the
// second behind round 1): a workgroup starts streaming the moment its own Wo rows are out, nobody waits for the slowest workgroup
// of a launch, and no launch boundary follows.
struct FfnXp {
    const u64 *gw;             // [8][3072] Wo partial sums of this layer's attention block
    const u64 *gxin;           // [3072] x'' of the previous FFN block
    u64 *gxp;                  // [3072] x' of this block (tagged a.epoch)
    unsigned w_epoch, x_epoch;
    const float *x_adapter;    // layer 0 (gxin == nullptr): x = x_adapter[row] + bf16(x_emb[row]), the step's embedding
    const uint16_t *x_emb;
};
// gx (optional): x'' also leaves as {epoch, value} granules (k_ffn_attn12: the attention block of the next layer in the same launch)
// w: the caller's registers for the W1 / W3 pieces (an array indexed with constants only, so that it stays in registers across the
// two bodies of k_dec_stack; a pointer to it would send it to scratch memory).
template <bool XP = false>
__device__ __forceinline__ void ffn_body(const FfnArgs &a, float *smem, u64 *gx, uint4 (&w)[2][3][6], const FfnXp &xp = FfnXp{}) {
    float *stage = smem;                 // [9][3072]: x, wo_part[0..7]; after the prologue: row 0 = x', rows 1..3 = h
    float *nws = smem + 9 * DF_D;        // [3072] norm weights
    float *ads = nws + DF_D;             // [3072] ada
    float *xs = ads + DF_D;              // [3072] normalised x'; phase 2: [12 waves][12 rows] partial sums
    float *red = xs + DF_D;              // [16]
    float *hs = stage + DF_D;            // [9216] h (phase 2), wave w owns [768 w, 768 w + 768)
    const int tid = df_tid(), lane = tid & 63;
    const int bid = df_bid();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wofs = (unsigned)wave * 1024u;
    unsigned long long df_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl0 = tl_begin(a.tl);
#define FFN_MARK(k) do { if (a.tl) df_stamp[k] = wall_clock64(); } while (0)
    FFN_MARK(0);

    // ---- phase 1: exactly k_gemv_w13x<false> (same memory-queue order, same arithmetic) --------------------------------------
    if constexpr (!XP) {
        glds16(a.x + tid * 4, lds_addr(stage) + wofs);
#pragma unroll
        for (int gi = 0; gi < 8; gi++) glds16(a.wo_part + (size_t)gi * DF_D + tid * 4, lds_addr(stage + (gi + 1) * DF_D) + wofs);
    }
    glds16(a.norm_w + tid * 4, lds_addr(nws) + wofs);
    glds16(a.ada + tid * 4, lds_addr(ads) + wofs);
    __builtin_amdgcn_sched_barrier(0);
    const int pair0 = bid * 36 + wave * 3;
    const uint4 *p1[3], *p3[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        p1[r] = reinterpret_cast<const uint4 *>(a.w1 + (size_t)(pair0 + r) * DF_D) + lane;
        p3[r] = reinterpret_cast<const uint4 *>(a.w3 + (size_t)(pair0 + r) * DF_D) + lane;
    }
#define FFN_ISSUE_A(C) { _Pragma("unroll") for (int r = 0; r < 3; r++) w[0][r][C] = ld_stream(p1[r] + (C) * 64); }
#define FFN_ISSUE_B(C) { _Pragma("unroll") for (int r = 0; r < 3; r++) w[1][r][C] = ld_stream(p3[r] + (C) * 64); }
#define FFN_ISSUE(C) { FFN_ISSUE_A(C) FFN_ISSUE_B(C) }
    FFN_ISSUE(0) FFN_ISSUE(1)
    __builtin_amdgcn_sched_barrier(0);
    FFN_MARK(1);
    if constexpr (XP) {
        // Schedule (per CU, 25 KB / us): pieces 0, 1 | hop-1 loads | pieces 2, 3 | sweep loads | piece 4.  A load returns behind
        // everything queued before it and is looked up in L2 about 2 us earlier: hop 1 is looked up ~4 us after this workgroup left
        // its attention block (the other workgroups leave theirs within ~2.5 us of each other), hop 2 ~10 us after, ~4 us after the
        // owners published.  Piece 4 is queued before anybody waits, so the stream never runs dry while x' is in transit.
        // hop 1 (wave 0): lane -> (row r = lane % 12, source s = lane / 12): sources s and s + 5 of the 9 (8 partials, then x'')
        float *hop = stage + DF_D;                                 // [9][16] (rows 1.. of the staging area are free in this mode)
        // (the addresses are derived again, from a laundered lane index, where a stale tag has to be waited for: kept live across the
        //  weight issue below they cost the registers of half a weight piece)
#define FFN_HOP_ADDR(LN)                                                                                         \
        const int r = (LN) % 12, s0 = (LN) / 12;                                                                   \
        const int orow = bid * 12 + r;                                                                             \
        const bool act0 = (LN) < 60, act1 = (LN) < 48;                                                             \
        const u64 *pa = xp.gw + (size_t)(act0 ? s0 : 0) * DF_D + orow;                                             \
        const u64 *pb = (s0 + 5 < 8 || !act1) ? xp.gw + (size_t)(act1 ? s0 + 5 : 0) * DF_D + orow : xp.gxin + orow; \
        const unsigned eb = (s0 + 5 < 8 || !act1) ? xp.w_epoch : xp.x_epoch;
        u64 va = 0, vb = 0;
        float xemb = 0.f;
        if (wave == 0) {
            FFN_HOP_ADDR(lane) (void)eb;
            va = df_load_granule(pa);
            if (s0 == 3 && !xp.gxin) xemb = xp.x_adapter[orow] + bf16_to_f32(xp.x_emb[orow]);      // layer 0: source 8 is the embedding itself
            else vb = df_load_granule(pb);
        }
        __builtin_amdgcn_sched_barrier(0);
        FFN_ISSUE(2) FFN_ISSUE(3)
        __builtin_amdgcn_sched_barrier(0);
        // hop 2 (waves 1 .. 11): the x' sweep, 5 granules per thread
        const int i0 = (wave - 1) * 64 + lane;
        u64 gv[5] = {0, 0, 0, 0, 0};
        if (wave > 0) {
#pragma unroll
            for (int u = 0; u < 5; u++) gv[u] = df_load_granule(xp.gxp + min(i0 + 704 * u, DF_D - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        FFN_ISSUE_A(4)                  // (half a piece - 36 KB per CU, 1.4 us - is what the register budget allows and what the RMSNorm needs as cover)
        __builtin_amdgcn_sched_barrier(0);
        if (wave == 0) {
            const int lane2 = df_tid() & 63;
            FFN_HOP_ADDR(lane2)
            if (act0 && (unsigned)(va >> 32) != xp.w_epoch) va = (u64)__float_as_uint(df_wait_granule_e(pa, xp.w_epoch, a.err, a.spin_limit, 5u));
            if (s0 == 3 && !xp.gxin) vb = (u64)__float_as_uint(xemb);
            else if (act1 && (unsigned)(vb >> 32) != eb) vb = (u64)__float_as_uint(df_wait_granule_e(pb, eb, a.err, a.spin_limit, 5u));
            if (act0) hop[s0 * 16 + r] = __uint_as_float((unsigned)va);
            if (act1) hop[(s0 + 5) * 16 + r] = __uint_as_float((unsigned)vb);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 12) {
                float v = hop[8 * 16 + lane];
#pragma unroll
                for (int gi = 0; gi < 8; gi++) v += hop[gi * 16 + lane];        // x, then groups 0 .. 7: the memory prologue's order
                df_store_granule(xp.gxp + bid * 12 + lane, a.epoch, v);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const int idx = i0 + 704 * u;
                if (idx < DF_D) {
                    if ((unsigned)(gv[u] >> 32) != a.epoch) gv[u] = (u64)__float_as_uint(df_wait_granule_e(xp.gxp + idx, a.epoch, a.err, a.spin_limit, 6u));
                    stage[idx] = __uint_as_float((unsigned)gv[u]);
                }
            }
        }
    } else {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // the 11 DMAs are in; round 0 of the weights still streams
    }
    __syncthreads();
    FFN_MARK(2);
    {
        float4 v = *reinterpret_cast<const float4 *>(stage + tid * 4);
        if constexpr (!XP) {
#pragma unroll
            for (int gi = 1; gi <= 8; gi++) {
                const float4 p = *reinterpret_cast<const float4 *>(stage + gi * DF_D + tid * 4);
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            *reinterpret_cast<float4 *>(stage + tid * 4) = v;          // x' stays in LDS for phase 2 (own elements: no hazard)
        }
        if (bid == 0 && a.xprime_out) *reinterpret_cast<float4 *>(a.xprime_out + tid * 4) = v;
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        ss = df_wave_sum<true>(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 12; i++) tot += red[i];
        const float inv = 1.0f / sqrtf(tot / (float)DF_D + a.eps);
        const float4 gw = *reinterpret_cast<const float4 *>(nws + tid * 4);
        const float4 sc = *reinterpret_cast<const float4 *>(ads + tid * 4);
        v.x = v.x * inv * gw.x * (1.0f + sc.x); v.y = v.y * inv * gw.y * (1.0f + sc.y);
        v.z = v.z * inv * gw.z * (1.0f + sc.z); v.w = v.w * inv * gw.w * (1.0f + sc.w);
        *reinterpret_cast<float4 *>(xs + tid * 4) = v;
        __syncthreads();
    }
    FFN_MARK(3);
    float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#define FFN_DOT(C)                                                                      \
    {   const float4 x0 = *reinterpret_cast<const float4 *>(xs + ((C) * 64 + lane) * 8);          \
        const float4 x1 = *reinterpret_cast<const float4 *>(xs + ((C) * 64 + lane) * 8 + 4);      \
        _Pragma("unroll") for (int m = 0; m < 2; m++)                                   \
            _Pragma("unroll") for (int r = 0; r < 3; r++) acc[m][r] = dot8_bf16(w[m][r][C], x0, x1, acc[m][r]); }
    if constexpr (!XP) { FFN_ISSUE(2) FFN_ISSUE(3) }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XP) {
        FFN_ISSUE_B(4)
        __builtin_amdgcn_sched_barrier(0);
        FFN_DOT(0)
        __builtin_amdgcn_sched_barrier(0);
        FFN_ISSUE(5)
        __builtin_amdgcn_sched_barrier(0);
        FFN_DOT(1)
    } else {
        FFN_DOT(0) FFN_DOT(1)
        __builtin_amdgcn_sched_barrier(0);
        FFN_ISSUE(4) FFN_ISSUE(5)
    }
    __builtin_amdgcn_sched_barrier(0);
    FFN_DOT(2) FFN_DOT(3)
    __builtin_amdgcn_sched_barrier(0);
    FFN_DOT(4) FFN_DOT(5)
#undef FFN_HOP_ADDR
#undef FFN_ISSUE
#undef FFN_ISSUE_A
#undef FFN_ISSUE_B
#undef FFN_DOT
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 3; r++) acc[m][r] = df_wave_sum<true>(acc[m][r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 3; r++) df_store_granule(a.gh + pair0 + r, a.epoch, silu(acc[0][r]) * acc[1][r]);      // voxtral_decoder.c:684-687
    }
    FFN_MARK(4);
    // ---- the phase edge: workgroup barrier (stage rows 1..8 are dead from here), this wave's W2 bytes, its share of the sweep behind them
    __syncthreads();
    uint4 w2r[18];
    {
        const unsigned char *w2s = reinterpret_cast<const unsigned char *>(a.w2) + ((size_t)bid * 12 * FFN_H + 768 * wave) * 2;
#pragma unroll
        for (int c = 0; c < 18; c++) {
            const int m = c / 3, k = c - 3 * m;
            const int r = k == 0 ? 2 * m : k == 2 ? 2 * m + 1 : 2 * m + (lane >> 5);
            const int col = k == 0 ? 8 * lane : k == 2 ? 256 + 8 * lane : (lane < 32 ? 512 + 8 * lane : 8 * (lane - 32));
            w2r[c] = ld_stream(reinterpret_cast<const uint4 *>(w2s + ((size_t)r * FFN_H + col) * 2));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    FFN_MARK(5);
    {
        u64 gv[12];
        const u64 *gsl = a.gh + 768 * wave + lane;
        // First attempt in straight-line code (inside a loop the compiler merges the wait counts of "W2 bytes queued" and "not queued" and
        // waits for all of W2 before it looks at the first tag); the retry loop is entered only if a tag was stale.
#pragma unroll
        for (int u = 0; u < 12; u++) gv[u] = df_load_granule(gsl + 64 * u);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < 12; u++) ok = ok && (unsigned)(gv[u] >> 32) == a.epoch;
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
            DfSpin sp; df_spin_begin(sp);
            for (unsigned it = 0;; it++) {
                if ((it & 15u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (df_spin_expired(sp, a.err, a.spin_limit, 3u, a.epoch)) break;
                __builtin_amdgcn_s_sleep(8);
                ok = true;
#pragma unroll
                for (int u = 0; u < 12; u++) gv[u] = df_load_granule(gsl + 64 * u);
#pragma unroll
                for (int u = 0; u < 12; u++) ok = ok && (unsigned)(gv[u] >> 32) == a.epoch;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;           // the wave moves as one: its loads go out together
            }
        }
#pragma unroll
        for (int u = 0; u < 12; u++) hs[768 * wave + 64 * u + lane] = __uint_as_float((unsigned)gv[u]);
    }
    // the slice is written and read by this wave only: LDS operations of one wave execute in order, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    FFN_MARK(6);

    // ---- phase 2: partial sums of all 12 rows over this wave's 768 columns, then x''[row] = x'[row] + the 12 partials in wave order ----
    {
        const float *hw = hs + 768 * wave;
        const int cb = lane < 32 ? 512 + 8 * lane : 8 * (lane - 32);
        const float4 hA0 = *reinterpret_cast<const float4 *>(hw + 8 * lane), hA1 = *reinterpret_cast<const float4 *>(hw + 8 * lane + 4);
        const float4 hB0 = *reinterpret_cast<const float4 *>(hw + cb), hB1 = *reinterpret_cast<const float4 *>(hw + cb + 4);
        const float4 hC0 = *reinterpret_cast<const float4 *>(hw + 256 + 8 * lane), hC1 = *reinterpret_cast<const float4 *>(hw + 256 + 8 * lane + 4);
        float rs[12];
#pragma unroll
        for (int m = 0; m < 6; m++) {
            const float q0 = dot8_bf16(w2r[3 * m], hA0, hA1, 0.f);
            const float q1 = dot8_bf16(w2r[3 * m + 1], hB0, hB1, 0.f);
            const float q2 = dot8_bf16(w2r[3 * m + 2], hC0, hC1, 0.f);
            rs[2 * m] = df_wave_sum<true>(q0 + (lane < 32 ? q1 : 0.f));
            rs[2 * m + 1] = df_wave_sum<true>(q2 + (lane < 32 ? 0.f : q1));
        }
        float *part = xs;
        if (lane == 0) {
#pragma unroll
            for (int r4 = 0; r4 < 3; r4++)
                *reinterpret_cast<float4 *>(part + wave * 12 + 4 * r4) = make_float4(rs[4 * r4], rs[4 * r4 + 1], rs[4 * r4 + 2], rs[4 * r4 + 3]);
        }
        __syncthreads();
        if (tid < 12) {
            float sum = 0.f;
#pragma unroll
            for (int w12 = 0; w12 < 12; w12++) sum += part[w12 * 12 + tid];         // fixed order
            const int orow = bid * 12 + tid;
            const float xv = stage[orow] + sum;
            a.x_out[orow] = xv;
            if (gx) df_store_granule(gx + orow, a.epoch, xv);
        }
    }
    FFN_MARK(7);
#undef FFN_MARK
    tl_end(a.tl, tl0, df_stamp, 8);
}
__global__ __launch_bounds__(FFN_THREADS, 1) void k_ffn_fused(const FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 w[2][3][6];
    ffn_body(a, smem, nullptr, w);
}

// ---------------------------------------------------------------------------------------------------------
// df_attn12_body (round 4, late) - k_dec_attn_fused's block in the FFN kernel's shape: 768 threads = 12 waves, at most 168 registers,
// specialised to the short-context regime the 30 s clip lives in (<= 8 key slices of 64 keys: one tile per attention member, the
// members carry no Wo rows, one XCD per KV head), bf16 weights, DPP reductions, layers > 0.  It exists so that the FFN block of
// layer l and the attention block of layer l + 1 can be ONE launch (k_ffn_attn12 below).  Same arithmetic as k_dec_attn_fused
// (same row mapping, same hand-off granules, same partials), differences in the shape only:
//   projection: 2 rows per wave (rows 2w, 2w+1 of the workgroup's 24), 12 loads per lane;
//   Wo rows:    the group's 3072 rows over (32 - ns) x 12 waves, <= 12 per wave;
//   attention:  waves 0..7 (the 64-key tile is 8 waves' work), waves 8..11 only keep the barriers.
// XG: x'' of the previous FFN block arrives as granules (gx, tagged x_epoch) instead of through memory after a kernel boundary:
// the sweep (4 loads per thread) is queued behind the first two pieces of the projection rows.
// ---------------------------------------------------------------------------------------------------------
constexpr int DA12_THREADS = 768, DA12_WAVES = 12, DA12_NWO = 12;
constexpr int DA12_LDS_BYTES = 2 * DF_D * 4 + 4 * DF_TILE_BYTES + 1024 + 512;      // [xs | nw], K/V tiles double buffered, inv_freq, red
//
// LONG (round 5): the same body for 9 .. 32 key slices (contexts beyond 1024 keys), so that the merged launch covers every context
// length.  What changes is what k_dec_attn_fused does differently there too: the members of a KV-head group are spread over all
// XCDs (group = blockIdx / 32: a group's K/V tiles and Wo rows then come through all eight L2s), EVERY workgroup carries Wo rows
// (96 per workgroup = 8 per wave) and merges the partials, an attention member requests its Wo rows BEHIND its partial (a CU's
// stores leave through the same queue as its loads), and the merge is the many-slices form: the (max, sum) pairs meet in LDS, every
// slice's weight exp(m_s - M) / L is computed once per head, and a thread's 32 (head, dim) granules share one trip to L2 with them
// while a slice is one tile (beyond that the members finish further apart and the early fetch would only be repeated).
// gw (k_dec_stack): the Wo partial sums leave as {epoch, value} granules (gw[8][3072]) instead of a.wo_part, see ffn_body<XP>.
// EMB (k_dec_stack, layer 0): x = adapter[st->adapter_row] + tok_emb[st->token] (voxtral.c:1057-1061), built here as k_dec_attn_fused<EMBED>
// builds it: the adapter row comes by LDS-DMA where x would, the embedding row is added on top.
template <bool XG, bool W8 = false, bool LONG = false, bool EMB = false>
__device__ __forceinline__ void df_attn12_body(const DecFuseArgs &a, unsigned char *smem_raw, const u64 *gx, unsigned x_epoch, u64 *gw = nullptr) {
    static_assert(!(EMB && XG), "layer 0 has no x'' in front of it");
    static_assert(!(LONG && W8), "the long-context form is bf16 only");
    float *xs = reinterpret_cast<float *>(smem_raw);                           // [3072] x, then [3072] norm weights (contiguous)
    float *nw = xs + DF_D;
    unsigned char *tiles = smem_raw + 2 * DF_D * 4;                            // [2 buffers][K tile | V tile]; later the Wo reduction scratch
    float *frq = reinterpret_cast<float *>(tiles + 4 * DF_TILE_BYTES);         // [256] inv_freq (64 valid)
    float *red = frq + 256;                                                    // [128]: wave sums, the 24 row results
    const int tid = df_tid(), lane = tid & 63;
    const int bid = df_bid();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = LONG ? bid / DF_BPG : bid % DF_GROUPS, j = LONG ? bid % DF_BPG : bid / DF_GROUPS;
    const unsigned epoch = a.epoch;
    const int pos = a.pos;
    unsigned long long df_stamp[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl0 = tl_begin(a.tl);
    DF_MARK(0);
    // ---- this workgroup's 24 projection rows: wave w streams rows 2w, 2w+1 ----------------------------------------------------------
    constexpr int NPW = W8 ? 3 : 6;               // 1 KiB pieces per projection row (fp8: 16 weights per lane and piece)
    constexpr int XSP = W8 ? 1 : 2;               // XG: pieces requested in front of the x'' sweep
    constexpr int NWL = W8 ? 6 : LONG ? 8 : DA12_NWO;        // Wo loads per wave (fp8: two rows per load, as in k_dec_attn_fused<W8>; LONG: 96 rows per workgroup)
    const unsigned char *rp[2];
    int prow[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int lr = 2 * wave + i;
        const int row = lr < 16 ? DF_NQ * g + 16 * j + lr
                      : lr < 20 ? DF_DQ + DF_HD * g + 4 * j + (lr - 16)
                                : DF_DQ + DF_DKV + DF_HD * g + 4 * j + (lr - 20);
        prow[i] = row;
        rp[i] = reinterpret_cast<const unsigned char *>(a.wqkv) + (size_t)row * (W8 ? DF_D : DF_D * 2) + lane * 16;
    }
    const int ns = a.nsplit;                                     // <= 8 (LONG: <= 32)
    int lo = pos - a.window + 1; if (lo < 0) lo = 0;
    const int s_lo = lo + j * a.split_keys;
    int s_hi = s_lo + a.split_keys - 1; if (s_hi > pos) s_hi = pos;
    const bool att_block = j < ns;
    // ---- t0: the vectors by LDS-DMA, then the weights -------------------------------------------------------------------------------
    if (wave < 1) glds16(reinterpret_cast<const unsigned char *>(a.inv_freq) + lane * 16, lds_addr(frq));
    if constexpr (XG) {
        glds16(a.norm_w + tid * 4, lds_addr(nw) + (unsigned)wave * 1024u);
    } else {
        const float *xsrc = a.x;
        if constexpr (EMB) xsrc = a.adapter + (size_t)a.st->adapter_row * DF_D;
#pragma unroll
        for (int p = 0; p < 2; p++) glds16((p ? a.norm_w : xsrc) + tid * 4, lds_addr(xs) + (unsigned)(p * 12288 + wave * 1024));
    }
    uint2 emb4 = make_uint2(0u, 0u);          // EMB: this thread's 4 embedding values (bf16), requested in front of the weights
    if constexpr (EMB) emb4 = *reinterpret_cast<const uint2 *>(a.tok_emb + (size_t)a.st->token * DF_D + tid * 4);
    DF_MARK(11);
    __builtin_amdgcn_s_barrier();          // every wave's share of the vectors is in the CU's queue before anybody's weights (see k_dec_attn_fused)
    DF_MARK(12);
    uint4 w[2][NPW];
    u64 gxv[4] = {0, 0, 0, 0};
    if constexpr (XG) {
#pragma unroll
        for (int c = 0; c < XSP; c++)
#pragma unroll
            for (int i = 0; i < 2; i++) w[i][c] = ld_stream(reinterpret_cast<const uint4 *>(rp[i] + c * 1024));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) gxv[u] = df_load_granule(gx + u * DA12_THREADS + tid);      // back when the first pieces are
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = XSP; c < NPW; c++)
#pragma unroll
            for (int i = 0; i < 2; i++) w[i][c] = ld_stream(reinterpret_cast<const uint4 *>(rp[i] + c * 1024));
    } else {
#pragma unroll
        for (int c = 0; c < NPW; c++)
#pragma unroll
            for (int i = 0; i < 2; i++) w[i][c] = ld_stream(reinterpret_cast<const uint4 *>(rp[i] + c * 1024));
    }
    __builtin_amdgcn_sched_barrier(0);
    // Wo rows (row r = Wo[r][512 g .. 512 g + 511] = 1 KiB): the 32 - ns non-members share the group's 3072 rows, <= 12 per wave
    uint4 wv[NWL];
    float wo_sc[W8 ? NWL : 1];
    const int nb = LONG ? DF_BPG : DF_BPG - ns, rpb = LONG ? DF_WO_ROWS : ((DF_D + nb - 1) / nb + 11) / 12 * 12;
    const int wo_rpw = rpb / 12;
    const int wo_row0 = LONG ? j * rpb + wave * wo_rpw : att_block ? DF_D : (j - ns) * rpb + wave * wo_rpw;
    const int wo_n = max(0, min(wo_rpw, DF_D - wo_row0));
    const int wo_rmax = wo_n > 0 ? wo_row0 + wo_n - 1 : DF_D - 1;
    const unsigned char *wo_base = reinterpret_cast<const unsigned char *>(a.wo) + (size_t)(DF_NQ * g) * (W8 ? 1 : 2) + (W8 ? (lane & 31) : lane) * 16;
    const int tile_slot0 = __builtin_amdgcn_readfirstlane(att_block ? s_lo % a.kv_cap : 0);
    DF_MARK(1);
    // ---- the activation vector -----------------------------------------------------------------------------------------------------
    if constexpr (XG) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < 4; u++) ok = ok && (unsigned)(gxv[u] >> 32) == x_epoch;
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
            DfSpin sp; df_spin_begin(sp);
            for (unsigned it = 0;; it++) {
                if ((it & 15u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (df_spin_expired(sp, a.err, a.spin_limit, 4u, x_epoch)) break;
                __builtin_amdgcn_s_sleep(4);
                ok = true;
#pragma unroll
                for (int u = 0; u < 4; u++) gxv[u] = df_load_granule(gx + u * DA12_THREADS + tid);
#pragma unroll
                for (int u = 0; u < 4; u++) ok = ok && (unsigned)(gxv[u] >> 32) == x_epoch;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) xs[u * DA12_THREADS + tid] = __uint_as_float((unsigned)gxv[u]);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NPW - XSP)) : "memory");    // the vector DMAs are older than everything else; the later pieces may still stream
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");             // only the weight loads are younger than the activation DMAs
        if constexpr (EMB) {     // a thread's 16 bytes of the adapter row were brought in by its own DMA lane
            float4 v = *reinterpret_cast<const float4 *>(xs + tid * 4);
            v.x += bf16_lo(emb4.x); v.y += bf16_hi(emb4.x); v.z += bf16_lo(emb4.y); v.w += bf16_hi(emb4.y);
            *reinterpret_cast<float4 *>(xs + tid * 4) = v;
        }
    }
    __syncthreads();
    DF_MARK(2);
    {   // RMSNorm (voxtral_kernels.c:346-363), under the weight stream: 4 elements per thread
        float4 v = *reinterpret_cast<const float4 *>(xs + tid * 4);
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        ss = df_wave_sum<true>(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < DA12_WAVES; i++) tot += red[i];
        const float inv = 1.0f / sqrtf(tot / (float)DF_D + a.eps);
        const float4 gw = *reinterpret_cast<const float4 *>(nw + tid * 4);
        v.x = v.x * inv * gw.x; v.y = v.y * inv * gw.y; v.z = v.z * inv * gw.z; v.w = v.w * inv * gw.w;
        *reinterpret_cast<float4 *>(xs + tid * 4) = v;
        __syncthreads();
    }
    DF_MARK(3);
    float rope_c = 1.f, rope_s = 0.f;
    if (tid < 10) {
        const int in_head = tid < 8 ? (16 * j + 2 * tid) % DF_HD : 4 * j + 2 * (tid - 8);
        const float ang = (float)pos * frq[in_head >> 1];
        rope_c = cosf(ang); rope_s = sinf(ang);
    }
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NPW; c++) {
        if constexpr (W8) {
            const float *xp = xs + (c * 64 + lane) * 16;
            const float4 x0 = *reinterpret_cast<const float4 *>(xp), x1 = *reinterpret_cast<const float4 *>(xp + 4);
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8), x3 = *reinterpret_cast<const float4 *>(xp + 12);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = dot16_fp8(w[i][c], x0, x1, x2, x3, acc[i]);
        } else {
            const float4 x0 = *reinterpret_cast<const float4 *>(xs + (c * 64 + lane) * 8);
            const float4 x1 = *reinterpret_cast<const float4 *>(xs + (c * 64 + lane) * 8 + 4);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = dot8_bf16(w[i][c], x0, x1, acc[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) :: "memory");     // the tile DMAs below stay below the dot products
    if (att_block && wave < 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) df_tile_op(a, g, s_lo, s_hi, lds_addr(tiles), lds_addr(tiles + DF_TILE_BYTES), wave, lane, k, true, tile_slot0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float sres = df_wave_sum<true>(acc[i]);
        if constexpr (W8) sres *= a.sqkv[prow[i]];                  // per-row dequantisation scale
        if (lane == 0) red[16 + 2 * wave + i] = sres;
    }
    DF_MARK(4);
    __syncthreads();
    // ---- RoPE, KV append, publish q/k/v: 12 threads, one (even, odd) row pair each ------------------------------------------------
    u64 *gq = a.gq + (size_t)g * DF_GQ;
    if (tid < 12) {
        const int p = tid;
        float e0 = red[16 + 2 * p], e1 = red[16 + 2 * p + 1];
        if (p < 10) { const float r0 = e0 * rope_c - e1 * rope_s, r1 = e0 * rope_s + e1 * rope_c; e0 = r0; e1 = r1; }     // voxtral_kernels.c:502-526
        if (p < 8) {
            df_store_granule(gq + 16 * j + 2 * p, epoch, e0); df_store_granule(gq + 16 * j + 2 * p + 1, epoch, e1);
        } else {
            const bool isk = p < 10;
            const int kl = 4 * j + 2 * (isk ? p - 8 : p - 10);
            const size_t slot = (size_t)(pos % a.kv_cap) * DF_DKV + DF_HD * g + kl;
            float *ring = isk ? a.kring : a.vring;
            ring[slot] = e0; ring[slot + 1] = e1;
            const int gi = DF_NQ + (isk ? 0 : DF_HD) + kl;
            df_store_granule(gq + gi, epoch, e0); df_store_granule(gq + gi + 1, epoch, e1);
        }
    }
    __syncthreads();                       // xs / nw are dead from here on: scratch for the attention stage
    DF_MARK(5);
    // A member whose slice has a second tile (513 .. 1024 keys) requests it HERE, behind its own publish and in front of its sweep: the
    // tile does not depend on q, the other buffer is free, and the sweep's first poll - which would wait for the slowest workgroup's
    // publish anyway - is what queues behind the 64 KB.  Before (round 5), tile 1 was requested under tile 0's math, i.e. after the sweep,
    // and its latency stood in the chain: same box, alternating builds, 520 / 600 / 800 / 1000 keys 1.372 / 1.345 / 1.369 / 1.354 ->
    // 1.346 / 1.333 / 1.346 / 1.330 ms per step (profiles/r05_tile1_early_ab.txt).  Not in the long form: there it costs 0.3 - 0.7 % at
    // 2500 - 3800 keys (a member's Wo rows are queued behind its partial, and with 32 members per group the slowest publish is later).
    if (!LONG && att_block && wave < 8 && s_hi - s_lo >= DF_TILE)
        df_tile_dma(a, g, s_lo + DF_TILE, s_hi, lds_addr(tiles + 2 * DF_TILE_BYTES), lds_addr(tiles + 3 * DF_TILE_BYTES), wave, lane);
    const bool pf_on = !LONG && a.pf.units > 0;
    const int pf_V = 32 * a.pf.units;
    const unsigned pf_lds = lds_addr(tiles) + 53248u + (unsigned)wave * 896u;       // 12 x 896 B behind the Wo reduction scratch (52 KB)
#define DA12_ISSUE_WO()                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NWL; i++) {                                                                         \
        if constexpr (W8) {      /* load i = rows wo_row0 + 2 i (lanes 0-31) and + 2 i + 1 (lanes 32-63), 16 weights per lane */ \
            const int r = min(wo_row0 + 2 * i + (lane >> 5), wo_rmax);                                                        \
            wv[i] = ld_stream(reinterpret_cast<const uint4 *>(wo_base + (size_t)r * DF_DQ));                                  \
            wo_sc[i] = a.so[r];                                                                                               \
        } else {                                                                                                              \
            wv[i] = ld_stream(reinterpret_cast<const uint4 *>(wo_base + (size_t)min(wo_row0 + i, wo_rmax) * (DF_DQ * 2)));    \
        }                                                                                                                     \
    }
    if (!att_block) {
        if (pf_on) {
            df_prefetch_units(a.pf, g, 0, pf_V, (j - ns) * DA12_WAVES + wave, (DF_BPG - ns) * DA12_WAVES, lane, pf_lds);
            __builtin_amdgcn_sched_barrier(0);
        }
        DA12_ISSUE_WO()
    }
    float *qs = xs;                        // [512] the group's q
    float *kvn = xs + 512;                 // [256] this step's k | v of head g
    float *att = xs + 768;                 // [512] merged attention output of the group's 4 heads
    float *pt = xs + 1792;                 // [4 heads][64 keys] softmax numerators
    float *cr = xs + 2048;                 // [4] max, [4] sum
    float *sc8 = xs + 2560;                // [4 heads][8 dim slices][64 keys] partial scores, then [4 key quarters][4 heads][128] partial outputs
    u64 *gp = a.gp + ((size_t)g * DF_BPG) * DF_GP;
    if (att_block) {
        // ---- hand-off 1: sweep the group's 768 granules ---------------------------------------------------------------------------------
        if (tid < 512) {
            u64 gv0 = df_load_granule(gq + tid), gv1 = 0;
            if (tid < 256) gv1 = df_load_granule(gq + 512 + tid);
            qs[tid] = (unsigned)(gv0 >> 32) == epoch ? __uint_as_float((unsigned)gv0) : df_wait_granule(gq + tid, epoch, a, 1u);
            if (tid < 256) kvn[tid] = (unsigned)(gv1 >> 32) == epoch ? __uint_as_float((unsigned)gv1) : df_wait_granule(gq + 512 + tid, epoch, a, 1u);
        }
        DF_MARK(6);
        // ---- attention over this workgroup's key slice: 64-key tiles, double buffered, online softmax across tiles (one tile up to 512 keys) ----
        const int n_tiles = (s_hi - s_lo + DF_TILE) / DF_TILE;
        const int ho = (tid >> 7) & 3, dd = tid & 127;
        float o_acc = 0.f;
        if (tid < 4) { cr[4 + tid] = -1e30f; cr[8 + tid] = 0.f; }
        for (int ti = 0; ti < n_tiles; ti++) {
            unsigned char *kt = tiles + (ti & 1) * 2 * DF_TILE_BYTES, *vt = kt + DF_TILE_BYTES;
            const int t0 = s_lo + ti * DF_TILE;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this tile has landed
            __syncthreads();
            if ((LONG || ti > 0) && ti + 1 < n_tiles && wave < 8)     // next tile into the other buffer, under this tile's math (short form: tile 1 was requested above)
                df_tile_dma(a, g, t0 + DF_TILE, s_hi, lds_addr(tiles + ((ti + 1) & 1) * 2 * DF_TILE_BYTES),
                            lds_addr(tiles + ((ti + 1) & 1) * 2 * DF_TILE_BYTES + DF_TILE_BYTES), wave, lane);
            if (t0 + DF_TILE > pos && t0 <= pos) {
                // this step's own K/V row is not visible in the ring to other CUs yet: patch it in from the hand-off
                const int key = pos - t0;
                if (tid < 32) *reinterpret_cast<float4 *>(kt + key * 512 + ((tid ^ (key & 31)) << 4)) = *reinterpret_cast<const float4 *>(kvn + tid * 4);
                else if (tid < 64) *reinterpret_cast<float4 *>(vt + key * 512 + ((tid - 32) << 4)) = *reinterpret_cast<const float4 *>(kvn + DF_HD + (tid - 32) * 4);
                __syncthreads();
            }
            if (ti == 0) DF_MARK(11);
            // scores: wave (0..7) -> 16 of the 128 dims for all 4 heads, lane -> key; the 8 partial sums per (head, key) meet in LDS
            if (wave < 8) {
                // v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 outer products per instruction, block b = lanes 4b .. 4b + 3: lane l supplies
                // A[i = l & 3] and B[j = l & 3] of its block and receives D[i = register][j = l & 3].  Block = 4 keys, i = head, j = key:
                // a = q[head l & 3][d], b = K[key l][d] -> register i of lane l = S[head i][key l], one instruction per dim (f32 products, f32
                // accumulation: the FMA chain it replaces, 16 matrix instructions instead of 64 FMAs + 12 LDS reads on the members' critical path)
                const unsigned char *krow = kt + lane * 512;
                const float *qh = qs + (lane & 3) * DF_HD;
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int ch = 4 * wave + c;
                    const float4 kv4 = *reinterpret_cast<const float4 *>(krow + ((ch ^ (lane & 31)) << 4));
                    const float4 q4 = *reinterpret_cast<const float4 *>(qh + ch * 4);
                    s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(q4.x, kv4.x, s4, 0, 0, 0);
                    s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(q4.y, kv4.y, s4, 0, 0, 0);
                    s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(q4.z, kv4.z, s4, 0, 0, 0);
                    s4 = __builtin_amdgcn_mfma_f32_4x4x1f32(q4.w, kv4.w, s4, 0, 0, 0);
                }
#pragma unroll
                for (int hh = 0; hh < 4; hh++) sc8[(hh * 8 + wave) * 64 + lane] = s4[hh];
            }
            __syncthreads();
            if (ti == 0) DF_MARK(12);
            if (wave < 4) {   // one wave per head: online softmax over this tile's keys
                float sv = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; w8++) sv += sc8[(wave * 8 + w8) * 64 + lane];
                sv *= a.scale;
                if (t0 + lane > s_hi) sv = -INFINITY;
                const float m_old = cr[4 + wave], l_old = cr[8 + wave];
                const float m_new = fmaxf(m_old, df_wave_max<true>(sv));
                const float p = expf(sv - m_new);
                const float corr = expf(m_old - m_new);
                const float l_new = l_old * corr + df_wave_sum<true>(p);
                pt[wave * 64 + lane] = p;
                if (lane == 0) { cr[wave] = corr; cr[4 + wave] = m_new; cr[8 + wave] = l_new; }
            }
            __syncthreads();
            if (tid < 512) {  // quarter sums: keys 16 kq .. 16 kq + 15 of the tile, dim dd, all 4 heads (only the valid keys)
                const int kq = tid >> 7;
                const float *vcol = reinterpret_cast<const float *>(vt) + dd;
                const int nv = min(DF_TILE, s_hi - t0 + 1);
                // the same instruction with block = 4 dims, i = head, j = dim: a = P[head l & 3][key], b = V[key][dim l] -> register i = O[head i][dim]
                f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
                const float *ph = pt + (tid & 3) * 64;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k0 = 16 * kq + 4 * i;
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = k0 + u < nv ? vcol[(k0 + u) * 128] : 0.f;
                    const float4 p4 = *reinterpret_cast<const float4 *>(ph + k0);
                    a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(p4.x, v[0], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(p4.y, v[1], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(p4.z, v[2], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(p4.w, v[3], a4, 0, 0, 0);
                }
#pragma unroll
                for (int hh = 0; hh < 4; hh++) sc8[(kq * 4 + hh) * DF_HD + dd] = a4[hh];
            }
            __syncthreads();
            if (tid < 512) {
                const float accv = (sc8[(0 * 4 + ho) * DF_HD + dd] + sc8[(1 * 4 + ho) * DF_HD + dd]) +
                                   (sc8[(2 * 4 + ho) * DF_HD + dd] + sc8[(3 * 4 + ho) * DF_HD + dd]);
                o_acc = o_acc * cr[ho] + accv;
            }
        }
        __syncthreads();
        if (tid < 512) {
            u64 *mine = gp + (size_t)j * DF_GP;
            df_store_granule(mine + ho * DF_HD + dd, epoch, o_acc);
            if (dd == 0) { df_store_granule(mine + 4 * DF_HD + 2 * ho, epoch, cr[4 + ho]); df_store_granule(mine + 4 * DF_HD + 2 * ho + 1, epoch, cr[8 + ho]); }
        }
        if constexpr (LONG) {      // a member's Wo rows: requested behind its partial (see k_dec_attn_fused)
            __builtin_amdgcn_sched_barrier(0);
            DA12_ISSUE_WO()
        }
        DF_MARK(7);
    } else {
        DF_MARK(6);
        DF_MARK(7);
    }
#undef DA12_ISSUE_WO
    if (LONG || !att_block) {
        // ---- hand-off 2: one thread polls, then everybody sweeps the group's partials and merges them in slice order -------------------
        const int h = (tid >> 7) & 3, d = tid & 127;
        float M = -1e30f, L = 0.f, O = 0.f;
        if (tid == 0) (void)df_wait_granule(gp + (size_t)(ns - 1) * DF_GP + 4 * DF_HD, epoch, a, 2u);
        __syncthreads();
        if (LONG && ns > 8) {
            // many slices: see k_dec_attn_fused (same arithmetic, same order)
            float *mlv = xs + 2080;                    // [32 slices][4 heads][2]
            float *scl = xs + 2080 + 256;              // [4 heads][32 slices]
            const bool early = a.split_keys <= 64;     // one tile per slice: the value granules share the (max, sum) pairs' trip to L2
            u64 gv[32];
            // slice base = wave-uniform (SGPR pair), per-thread part = one 32-bit byte offset: 32 addresses cost one VGPR, not 64
            const unsigned toff = (unsigned)(h * DF_HD + d) * 8u;
#define DA12_GV_PTR(u) reinterpret_cast<const u64 *>(reinterpret_cast<const unsigned char *>(gp) + (size_t)min((u), ns - 1) * (DF_GP * 8) + toff)
            if (early && tid < 512) {
#pragma unroll
                for (int u = 0; u < 32; u++) gv[u] = df_load_granule(DA12_GV_PTR(u));
            }
            if (tid < ns * 8) {
                const u64 *src = gp + (size_t)(tid >> 3) * DF_GP + 4 * DF_HD + (tid & 7);
                mlv[tid] = df_wait_granule(src, epoch, a, 2u);
            }
            __syncthreads();
            if (tid < 128) {       // thread -> (head = tid >> 5, slice = tid & 31): max and sum over the 32-lane segment
                const int hh = tid >> 5, s1 = tid & 31;
                const float ms = s1 < ns ? mlv[(s1 * 4 + hh) * 2] : -1e30f, lsum = s1 < ns ? mlv[(s1 * 4 + hh) * 2 + 1] : 0.f;
                float Mx = ms;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) Mx = fmaxf(Mx, __shfl_xor(Mx, o, 32));
                const float ew = expf(ms - Mx);
                float Ls = lsum * ew;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) Ls += __shfl_xor(Ls, o, 32);
                scl[hh * 32 + s1] = Ls > 0.f ? ew / Ls : 0.f;
            }
            __syncthreads();
            if (tid < 512) {
                if (!early) {
#pragma unroll
                    for (int u = 0; u < 32; u++) gv[u] = df_load_granule(DA12_GV_PTR(u));
                }
                // a stale tag (rare: the pairs of every slice are in) is waited for granule by granule - no retry loop over all 32
                // addresses, which the compiler would keep in 64 registers across the loop
#pragma unroll
                for (int u = 0; u < 32; u++)
                    if (u < ns && (unsigned)(gv[u] >> 32) != epoch) gv[u] = (u64)__float_as_uint(df_wait_granule(DA12_GV_PTR(u), epoch, a, 2u));
#pragma unroll
                for (int u = 0; u < 32; u++)
                    if (u < ns) O = fmaf(scl[h * 32 + u], __uint_as_float((unsigned)gv[u]), O);
                att[h * DF_HD + d] = O;
#undef DA12_GV_PTR
            }
        } else
        {
            for (int s0 = 0; s0 < ns; s0 += 4) {
                u64 gv[4][3];
                DfSpin sp; df_spin_begin(sp);
                for (unsigned it = 0;; it++) {
                    bool ok = true;
                    if (tid < 512) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const u64 *part = gp + (size_t)min(s0 + u, ns - 1) * DF_GP;
                            gv[u][0] = df_load_granule(part + h * DF_HD + d);
                            gv[u][1] = df_load_granule(part + 4 * DF_HD + 2 * h);
                            gv[u][2] = df_load_granule(part + 4 * DF_HD + 2 * h + 1);
                        }
                    }
                    if (tid < 512) {
#pragma unroll
                        for (int u = 0; u < 4; u++)
#pragma unroll
                            for (int k = 0; k < 3; k++) ok = ok && (unsigned)(gv[u][k] >> 32) == epoch;
                    }
                    if (ok) break;
                    if ((it & 15u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    if (df_spin_expired(sp, a.err, a.spin_limit, 2u, epoch)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                if (tid < 512) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (s0 + u >= ns) break;
                        const float v0 = __uint_as_float((unsigned)gv[u][0]), vm = __uint_as_float((unsigned)gv[u][1]), vl = __uint_as_float((unsigned)gv[u][2]);
                        const float mn = fmaxf(M, vm);
                        const float c0 = expf(M - mn), c1 = expf(vm - mn);
                        L = L * c0 + vl * c1;
                        O = O * c0 + v0 * c1;
                        M = mn;
                    }
                }
            }
            if (tid < 512) att[h * DF_HD + d] = L > 0.f ? O / L : 0.f;
        }
        DF_MARK(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the Wo rows (and the prefetch DMAs) have landed
        __syncthreads();
        DF_MARK(9);
        if constexpr (W8) {
            if (wo_n > 0) {
                // lane l holds 16 weights of row wo_row0 + 2 i + (l >> 5), columns 16 (l & 31) .. + 15 of the group's 512
                const float *xp = att + (lane & 31) * 16;
                const float4 x0 = *reinterpret_cast<const float4 *>(xp), x1 = *reinterpret_cast<const float4 *>(xp + 4);
                const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8), x3 = *reinterpret_cast<const float4 *>(xp + 12);
                float sums[NWL];
#pragma unroll
                for (int i = 0; i < NWL; i++) sums[i] = row16_sum<true>(dot16_fp8(wv[i], x0, x1, x2, x3, 0.f));
#pragma unroll
                for (int i = 0; i < NWL; i++) {
                    const float o = __shfl_xor(sums[i], 16, 64);
                    const int row = wo_row0 + 2 * i + (lane >> 5);
                    if ((lane & 31) == 0 && row < wo_row0 + wo_n) {
                        if (gw) df_store_granule(gw + (size_t)g * DF_D + row, epoch, (sums[i] + o) * wo_sc[i]);
                        else a.wo_part[(size_t)g * DF_D + row] = (sums[i] + o) * wo_sc[i];
                    }
                }
            }
        } else
        if (wo_n > 0) {
            // the wave's <= 12 row sums as ONE transposed reduction through LDS (see k_dec_attn_fused)
            const float4 x0 = *reinterpret_cast<const float4 *>(att + lane * 8);
            const float4 x1 = *reinterpret_cast<const float4 *>(att + lane * 8 + 4);
            float *wr = reinterpret_cast<float *>(tiles) + wave * (16 * 68);
#pragma unroll
            for (int i = 0; i < NWL; i++) wr[i * 68 + lane] = dot8_bf16(wv[i], x0, x1, 0.f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int row = lane >> 2, q = lane & 3;
            float sres = 0.f;
            if (row < NWL) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float4 v = *reinterpret_cast<const float4 *>(wr + row * 68 + q * 16 + 4 * c);
                    sres += (v.x + v.y) + (v.z + v.w);
                }
            }
            sres += __shfl_xor(sres, 1, 4);
            sres += __shfl_xor(sres, 2, 4);
            if (q == 0 && row < wo_n) {
                if (gw) df_store_granule(gw + (size_t)g * DF_D + wo_row0 + row, epoch, sres);
                else a.wo_part[(size_t)g * DF_D + wo_row0 + row] = sres;
            }
        }
        DF_MARK(10);
    }
    tl_end(a.tl, tl0, df_stamp, 13);
}
template <bool LONG>
__global__ __launch_bounds__(DA12_THREADS, 1) void k_attn12(const DecFuseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem12[];
    df_attn12_body<false, false, LONG>(a, smem12, nullptr, 0u);
}
// fp8 mode: the W2 launch of layer l, then the attention block of layer l + 1 (fp8 projection / Wo rows), one launch
__global__ __launch_bounds__(W2X_THREADS, 1) void k_w2x_attn12(const W2xArgs f, const DecFuseArgs a, u64 *gx, unsigned gx_epoch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    w2x_body<true>(f, smem, gx, gx_epoch);
    __syncthreads();
    df_attn12_body<true, true>(a, reinterpret_cast<unsigned char *>(smem), gx, gx_epoch);
}
constexpr int FA12_LDS_BYTES = FFN_LDS_BYTES > DA12_LDS_BYTES ? FFN_LDS_BYTES : DA12_LDS_BYTES;
// FFN block of layer l, then the attention block of layer l + 1, one launch (fa = the attention block's arguments)
template <bool LONG>
__global__ __launch_bounds__(FFN_THREADS, 1) void k_ffn_attn12(const FfnArgs f, const DecFuseArgs a, u64 *gx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 w[2][3][6];
    ffn_body(f, smem, gx, w);
    __syncthreads();                        // every reader of the FFN block's LDS is done
    df_attn12_body<true, false, LONG>(a, reinterpret_cast<unsigned char *>(smem), gx, f.epoch);
}

// ---------------------------------------------------------------------------------------------------------
// k_dec_stack (round 5) - layers' FFN and attention blocks of a decoder step as ONE launch: FFN(0), then for l = 1 .. L-1 the attention
// block of layer l and its FFN block.  (Layer 0's attention block stays a launch of its own: it builds x from the adapter row and the
// previous token's embedding, which the argmax kernel of the previous step has only just produced.)  Every hand-off inside is one of
// the {epoch, value} granule sweeps of the kernels above, each queued behind weight bytes: h behind W2, x'' behind the first projection
// pieces, and - new - x' in two hops behind rounds 0 and 1 of W1 / W3 (ffn_body<XP>).  The granule buffers are reused layer after
// layer; a layer's tag is epoch0 + l.  Reuse is safe without any clearing because every buffer's producers of layer l + 1 depend,
// through the all-to-all hand-offs in between, on every consumer of layer l having finished: see DESIGN.md 3.
// bf16 only.  The fp8 form of this kernel (k_dec_stack8: fp8 attention body + an fp8 FFN body with the same two x' hops, built and correct -
// profiles/r05_fp8_stack.patch) is SLOWER than fp8 mode's two launches per layer: 1.043 / 1.122 / 1.135 against 0.963 / 1.057 / 1.078 ms per
// step at 232 / 600 / 1000 keys (profiles/r05_fp8_stack_ab.txt) - with half the bytes per piece every hand-off sweep returns before its
// producers have published, each retry costs a piece time, and the h sweep cannot hide behind 28 MB of W2.
// Up to 8 key slices (1024 keys).  Beyond, the step stays one k_ffn_attn12<LONG> launch per layer: measured, the long form of this
// kernel gains nothing there (-1 .. +1 %, profiles/r05_stack_ab.txt) - its attention block ends in a chain of three trips to L2
// that no weight byte covers, and the boundary it saves is short next to that.
// ---------------------------------------------------------------------------------------------------------
struct DecStackLayer {
    const uint16_t *wqkv, *wo, *w1, *w3, *w2;
    const float *n1, *n2, *ada;
    float *kring, *vring;
};
struct DecStackArgs {
    const DecStackLayer *layers;
    int n_layers;
    float eps;
    const float *inv_freq;
    int kv_cap, pos, window;
    float scale;
    const float *x0;           // [3072] residual stream in front of layer 0's attention output (written by its launch)
    const float *wo_part;      // [8][3072] layer 0's Wo partial sums (memory, from its launch)
    // embed != 0: layer 0's attention block runs in here too, on x = adapter[st->adapter_row] + tok_emb[st->token] (then x0 / wo_part are unused)
    int embed;
    const float *adapter; const uint16_t *tok_emb; const DecState *st;
    float *x_out;              // [3072] the stack's output
    u64 *gq, *gp, *gh, *gx, *gw, *gxp;
    unsigned epoch0;           // epoch of layer 0's attention launch; layer l uses epoch0 + l
    int split_keys, nsplit;
    unsigned *err;
    unsigned long long spin_limit;
    unsigned long long *tl;    // k_dec_stack<true> only (tuning): per-workgroup timeline of layer tl_layer's two blocks
    int tl_layer;
};
// Measured and not kept (profiles/r05_stack_parked_piece.patch, r05_stack_timeline_pre_p0_kv232.txt): the attention block parking piece 0
// of the following FFN block's rows in LDS by LDS-DMA once its partial sweep is in, to stream during the merge and the Wo product.
// A CU's stores leave through the same queue as its loads: the Wo partial sums (gw) then became visible only behind those 72 KB,
// every owner's hop 1 waited for them, and the block took 20 us instead of 16.3 (step +1 .. +3 %).  The only free slot for bulk
// requests is behind a block's LAST hand-off - where every block of this kernel already issues its weights.
template <bool TL>
__global__ __launch_bounds__(FFN_THREADS, 1) void k_dec_stack(const DecStackArgs s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 w[2][3][6];
    for (int l = 0; l < s.n_layers; l++) {
        const DecStackLayer &L = s.layers[l];
        if (l > 0 || s.embed) {
            DecFuseArgs a{};
            a.wqkv = L.wqkv; a.wo = L.wo; a.norm_w = L.n1; a.eps = s.eps; a.inv_freq = s.inv_freq;
            a.kring = L.kring; a.vring = L.vring; a.kv_cap = s.kv_cap; a.pos = s.pos; a.window = s.window; a.scale = s.scale;
            a.gq = s.gq; a.gp = s.gp; a.epoch = s.epoch0 + l; a.split_keys = s.split_keys; a.nsplit = s.nsplit;
            a.err = s.err; a.spin_limit = s.spin_limit;
            if constexpr (TL) a.tl = (l == s.tl_layer) ? s.tl : nullptr;
            if (l == 0) {
                a.adapter = s.adapter; a.tok_emb = s.tok_emb; a.st = s.st;
                df_attn12_body<false, false, false, true>(a, reinterpret_cast<unsigned char *>(smem), nullptr, 0u, s.gw);
            } else {
                df_attn12_body<true, false, false>(a, reinterpret_cast<unsigned char *>(smem), s.gx, s.epoch0 + l - 1, s.gw);
            }
            __syncthreads();
        }
        FfnArgs f{};
        f.w1 = L.w1; f.w3 = L.w3; f.w2 = L.w2; f.x = s.x0; f.wo_part = s.wo_part; f.norm_w = L.n2; f.ada = L.ada; f.eps = s.eps;
        f.x_out = s.x_out; f.gh = s.gh; f.epoch = s.epoch0 + l; f.err = s.err; f.spin_limit = s.spin_limit;
        if constexpr (TL) f.tl = (l == s.tl_layer && s.tl) ? s.tl + TL_STRIDE * 1024 : nullptr;
        if (l == 0 && !s.embed) {
            ffn_body<false>(f, smem, s.gx, w);
        } else {
            FfnXp xp{s.gw, l ? s.gx : nullptr, s.gxp, s.epoch0 + (unsigned)l, s.epoch0 + (unsigned)l - 1u, nullptr, nullptr};
            if (l == 0) { xp.x_adapter = s.adapter + (size_t)s.st->adapter_row * DF_D; xp.x_emb = s.tok_emb + (size_t)s.st->token * DF_D; }
            ffn_body<true>(f, smem, s.gx, w, xp);
        }
        __syncthreads();                        // every reader of the FFN block's LDS is done
    }
}

}  // namespace vox
