// vox_common.h — device helpers shared by all gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vox {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Device-resident decoder cursor.  Every kernel of a decode step reads its position
// from here, so an identical launch sequence (or one captured hipGraph) serves every
// step and several steps can be queued without a host round trip.
struct DecState {
    int pos;          // logical position of the token being processed (RoPE / KV slot)
    int token;        // previous token id -> embedding of the next step
    int n_out;        // tokens written to tokens_out so far (this run)
    int stop;         // set once eos was produced (later queued steps become no-ops)
    long long adapter_row;  // physical adapter row of this step
};

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// Full-wave (64 lane) butterfly reductions; every lane ends with the result.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum across the 16 lanes of a DPP row (lanes 16r..16r+15); all 16 lanes get the sum.
// DPP path: quad_perm xor1 (0xB1), quad_perm xor2 (0x4E), row_half_mirror (0x141),
// row_mirror (0x140) — four VALU ops, no LDS traffic.  USE_DPP=false is the
// __shfl_xor (ds_bpermute) form with identical semantics; the engine self-tests
// the DPP form at start-up and falls back if the two ever disagree.
template <bool USE_DPP>
__device__ __forceinline__ float row16_sum(float v) {
    if (USE_DPP) {
        int x;
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);
        v += __int_as_float(x);
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);
        v += __int_as_float(x);
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);
        v += __int_as_float(x);
        x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true);
        v += __int_as_float(x);
        return v;
    } else {
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        return v;
    }
}

// tanh-approximation GELU exactly as voxtral_kernels.c:376-384.
__device__ __forceinline__ float gelu_tanh(float v) {
    float x3 = v * v * v;
    float inner = 0.7978845608028654f * (v + 0.044715f * x3);
    return 0.5f * v * (1.0f + tanhf(inner));
}
// SiLU as voxtral_kernels.c:369-374.
__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

// 16-byte streaming load of weights that are read exactly once per pass.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4 *p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Async global -> LDS copy of 16 bytes per lane (gfx950 LDS-DMA): the wave's 64 pieces land at
// lds_base (wave-uniform LDS byte address) + lane*16; no VGPR is used for the data.  hipcc does
// not count asm memory operations: the caller waits with an explicit s_waitcnt vmcnt(N), N =
// the number of YOUNGER loads it allows to stay in flight (loads return in order).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {     // generic pointer into LDS -> LDS byte offset
    return (unsigned)(unsigned long long)p;
}

__device__ __forceinline__ float dot8_bf16(const uint4 w, const float4 x0, const float4 x1, float acc) {
    acc = fmaf(bf16_lo(w.x), x0.x, acc);
    acc = fmaf(bf16_hi(w.x), x0.y, acc);
    acc = fmaf(bf16_lo(w.y), x0.z, acc);
    acc = fmaf(bf16_hi(w.y), x0.w, acc);
    acc = fmaf(bf16_lo(w.z), x1.x, acc);
    acc = fmaf(bf16_hi(w.z), x1.y, acc);
    acc = fmaf(bf16_lo(w.w), x1.z, acc);
    acc = fmaf(bf16_hi(w.w), x1.w, acc);
    return acc;
}

// 16 fp8 (e4m3) weights x 16 f32 activations.  v_cvt_pk_f32_fp8 unpacks two bytes per instruction;
// the quantiser (k_quant_fp8_rows) uses the inverse instruction, so the pair is self-consistent.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot16_fp8(const uint4 w, const float4 x0, const float4 x1, const float4 x2, const float4 x3,
                                           float acc) {
    f32x2 p;
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.x, false); acc = fmaf(p.x, x0.x, acc); acc = fmaf(p.y, x0.y, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.x, true);  acc = fmaf(p.x, x0.z, acc); acc = fmaf(p.y, x0.w, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.y, false); acc = fmaf(p.x, x1.x, acc); acc = fmaf(p.y, x1.y, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.y, true);  acc = fmaf(p.x, x1.z, acc); acc = fmaf(p.y, x1.w, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.z, false); acc = fmaf(p.x, x2.x, acc); acc = fmaf(p.y, x2.y, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.z, true);  acc = fmaf(p.x, x2.z, acc); acc = fmaf(p.y, x2.w, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.w, false); acc = fmaf(p.x, x3.x, acc); acc = fmaf(p.y, x3.y, acc);
    p = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.w, true);  acc = fmaf(p.x, x3.z, acc); acc = fmaf(p.y, x3.w, acc);
    return acc;
}


// ---- bounded spins (round 6) --------------------------------------------------------------------------------------------------
// A hand-off takes microseconds; a spin that has WAITED for spin_limit ticks (5 ms) means a producer is not running (the 256
// workgroups are not co-resident) and flags the batch.  What must not count as waiting is time in which the spinning wave itself
// did not run: when the process has more HSA queues than the hardware scheduler maps (long test processes: every engine and every
// torch stream ever created), the queue is time-sliced - the whole dispatch is saved, another queue runs for milliseconds, the
// dispatch is restored - and a wall-clock difference taken across that hole says "5 ms" although every producer was frozen as
// well.  So the spin budget is ACTIVE time: the sum of the gaps between consecutive polls, each capped at DF_GAP_CAP (a poll of
// these loops takes 1 - 3 us; a gap of more than 40 us is a hole, not a wait).  Holes are recorded in err[8..9] (largest gap,
// count) whether or not anything times out, and a time-out records who, where and what it saw in err[1..7] for the host's message.
constexpr unsigned long long DF_GAP_CAP = 4000ull;          // 40 us at the 100 MHz wall clock
constexpr unsigned long long DF_HOLE = 100000ull;           // gaps beyond 1 ms are reported as holes
struct DfSpin { unsigned last, active, maxgap; };            // (32-bit ticks: differences of the low word are all that is used)
__device__ __forceinline__ void df_spin_begin(DfSpin &s) { s.last = (unsigned)wall_clock64(); s.active = 0; s.maxgap = 0; }
// One poll has failed: true = the budget is spent (err is set and the caller gives up), false = poll again.
__device__ __forceinline__ bool df_spin_expired(DfSpin &s, unsigned *err, unsigned long long spin_limit, unsigned code, unsigned epoch) {
    const unsigned now = (unsigned)wall_clock64(), d = now - s.last;
    s.last = now;
    if (d > s.maxgap) s.maxgap = d;
    if (d > (unsigned)DF_HOLE) {
        __hip_atomic_fetch_max(err + 8, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(err + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s.active += d < (unsigned)DF_GAP_CAP ? d : (unsigned)DF_GAP_CAP;
    if (s.active <= (unsigned)spin_limit) return false;
    if (__hip_atomic_exchange(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {      // the first reporter describes itself
        err[1] = blockIdx.x; err[2] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);       // HW_REG_XCC_ID
        err[3] = s.maxgap; err[4] = s.active; err[6] = epoch; err[7] = threadIdx.x;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Tuning aid (VOX_HIP_FUSE_TL): per-workgroup timeline of one launch, 16 words per workgroup: [0] wall clock at entry,
// [1] at exit, [2] (XCC_ID << 32) | HW_ID, [3 ..] the kernel's phase stamps - start skew, tails, XCD placement and where
// every workgroup spends its time.
constexpr int TL_STRIDE = 16;
__device__ __forceinline__ unsigned long long tl_begin(const unsigned long long *tl) { return tl ? wall_clock64() : 0ull; }
__device__ __forceinline__ void tl_end(unsigned long long *tl, unsigned long long t0, const unsigned long long *stamps = nullptr, int n = 0) {
    if (!tl) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID, bits 3:0
        unsigned long long *r = tl + (size_t)TL_STRIDE * blockIdx.x;
        r[0] = t0; r[1] = wall_clock64(); r[2] = ((unsigned long long)xcc << 32) | hw;
        for (int k = 0; k < n && k < TL_STRIDE - 3; k++) r[3 + k] = stamps[k];
    }
}
}  // namespace vox
