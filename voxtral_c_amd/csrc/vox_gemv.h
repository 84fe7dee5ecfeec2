// vox_gemv.h — decode-time (M = 1) bf16-weight matrix-vector kernels for gfx950.
//
// Replaces bf16_matvec_fused (voxtral_kernels.c:154-195) and everything the reference
// wraps around it per decoder step (voxtral_decoder.c:653-704): RMSNorm, ada scaling,
// RoPE, KV append, residual adds, SwiGLU gating, tied-embedding logits and argmax are
// fused into the prologue / epilogue of the GEMV that produces or consumes them.
//
// Roofline: HBM.  2 FLOP per 2-byte weight; algorithmic bytes = N*K*2 per launch.
// Layout: W row-major [N, K] bf16 exactly as stored in the safetensors file.  One
// wave owns RPW consecutive rows; the 64 lanes stride the row in 16-byte pieces
// (8 bf16), so a wave-instruction reads 1 KiB of contiguous weights (fully coalesced,
// non-temporal: every weight byte is touched once per token).  x (K floats) is staged
// once per block in LDS as f32 after the fused normalisation; fp32 FMA accumulation,
// butterfly reduction across the wave => same arithmetic as the oracle up to the
// summation order.  Deterministic: no atomics, fixed reduction trees.
#pragma once
#include "vox_common.h"

namespace vox {

enum { PRO_NONE = 0, PRO_RMS = 1, PRO_EMBED_RMS = 2, PRO_ATTN = 3 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_SWIGLU = 3, EPI_LOGITS = 4 };

struct GemvArgs {
    const uint16_t *W;     // [N, K]
    const uint16_t *W2;    // EPI_SWIGLU: up-projection rows (w3) [N, K]
    const float *x;        // [K]
    const float *norm_w;   // PRO_RMS: [K]
    const float *ada;      // PRO_RMS: optional [K] (x_norm *= 1 + ada), voxtral_decoder.c:679-682
    float eps;
    float *y;              // EPI_STORE/RESID/SWIGLU/LOGITS: [N]; EPI_QKV: q buffer [q_rows]
    const float *bias;     // optional [N] (EPI_STORE / EPI_RESID)
    int N, K;
    // EPI_QKV
    int q_rows, k_rows;    // rows [0,q_rows) -> q, [q_rows, q_rows+k_rows) -> K cache, rest -> V cache
    int head_dim;
    const float *rope;     // [head_dim/2][2] cos,sin for the current position
    float *kcache, *vcache;  // this layer's ring: [kv_cap][kv_dim]
    int kv_cap, kv_dim;
    const DecState *st;
    // EPI_LOGITS
    float *blk_val;
    int *blk_idx;
    // fast-path extras
    const float *inv_freq;     // EPI_QKV: RoPE computed in the epilogue from st->pos
    const float *adapter;      // PRO_EMBED_RMS: x = adapter[st->adapter_row] + tok_emb[st->token]
    const uint16_t *tok_emb;
    float *x_out;              // PRO_EMBED_RMS: residual stream (written by block 0)
    const float *part_o, *part_ml;   // PRO_ATTN: split-K attention partials [heads][nsplit][HD], [..][2]
    int nsplit, attn_hd;
    int pos_host;              // k_gemv3 EPI_QKV: logical position of this step (RoPE angle, KV slot)
    const float *wscale, *wscale2;   // fp8 weights (W8): per-row dequantisation scale of W / W2
    int row_base;              // EPI_LOGITS: added to the row index reported in blk_idx (vocabulary halves)
    unsigned long long *tl;    // optional (tuning, k_gemv3): per-workgroup timeline, see tl_begin / tl_end
};

template <int PRO, int EPI, int RPW, bool W8 = false>
__global__ __launch_bounds__(256) void k_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;                 // [K]
    float *red = smem + a.K;          // [16] scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    // ---- prologue: stage x in LDS, optionally RMS-normalised --------------------
    float ss = 0.f;
    for (int i = tid * 4; i < K; i += 256 * 4) {
        float4 v = *reinterpret_cast<const float4 *>(a.x + i);
        if (PRO == PRO_RMS) ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        *reinterpret_cast<float4 *>(xs + i) = v;
    }
    if (PRO == PRO_RMS) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float inv = 1.0f / sqrtf(tot / (float)K + a.eps);   // voxtral_kernels.c:356-357
        for (int i = tid * 4; i < K; i += 256 * 4) {
            float4 v = *reinterpret_cast<float4 *>(xs + i);
            const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
            v.x = v.x * inv * w.x; v.y = v.y * inv * w.y; v.z = v.z * inv * w.z; v.w = v.w * inv * w.w;
            if (a.ada) {
                const float4 s = *reinterpret_cast<const float4 *>(a.ada + i);
                v.x *= (1.0f + s.x); v.y *= (1.0f + s.y); v.z *= (1.0f + s.z); v.w *= (1.0f + s.w);
            }
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
    }
    __syncthreads();

    constexpr int EPP = W8 ? 16 : 8;  // weights per 16-byte piece
    const int nchunks = K / EPP;
    constexpr int NMAT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int RPB = 4 * RPW;      // rows per block iteration
    float best_v = -3.0e38f;
    int best_i = 0x7fffffff;

    for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < N; r0 += gridDim.x * RPB) {
        float acc[NMAT][RPW];
        const uint4 *wp[NMAT][RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = min(r0 + r, N - 1);   // clamp: duplicates are discarded below
            const size_t rowb = (size_t)K * (W8 ? 1 : 2);
            wp[0][r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.W) + (size_t)row * rowb);
            if constexpr (NMAT == 2) wp[1][r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.W2) + (size_t)row * rowb);
#pragma unroll
            for (int m = 0; m < NMAT; m++) acc[m][r] = 0.f;
        }
#pragma unroll 2
        for (int c = lane; c < nchunks; c += 64) {
            uint4 w[NMAT][RPW];
#pragma unroll
            for (int m = 0; m < NMAT; m++)
#pragma unroll
                for (int r = 0; r < RPW; r++) w[m][r] = ld_stream(wp[m][r] + c);
            const float4 x0 = *reinterpret_cast<const float4 *>(xs + c * EPP);
            const float4 x1 = *reinterpret_cast<const float4 *>(xs + c * EPP + 4);
            if constexpr (W8) {
                const float4 x2 = *reinterpret_cast<const float4 *>(xs + c * EPP + 8);
                const float4 x3 = *reinterpret_cast<const float4 *>(xs + c * EPP + 12);
#pragma unroll
                for (int m = 0; m < NMAT; m++)
#pragma unroll
                    for (int r = 0; r < RPW; r++) acc[m][r] = dot16_fp8(w[m][r], x0, x1, x2, x3, acc[m][r]);
            } else {
#pragma unroll
                for (int m = 0; m < NMAT; m++)
#pragma unroll
                    for (int r = 0; r < RPW; r++) acc[m][r] = dot8_bf16(w[m][r], x0, x1, acc[m][r]);
            }
        }
#pragma unroll
        for (int m = 0; m < NMAT; m++)
#pragma unroll
            for (int r = 0; r < RPW; r++) acc[m][r] = wave_sum(acc[m][r]);
        if constexpr (W8) {
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int row = min(r0 + r, N - 1);
                acc[0][r] *= a.wscale[row];
                if constexpr (NMAT == 2) acc[1][r] *= a.wscale2[row];
            }
        }

        // ---- epilogue (lane 0 of the wave owns the RPW results) -------------------
        if (lane == 0) {
            if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
                for (int r = 0; r < RPW; r++) {
                    const int row = r0 + r;
                    if (row < N) {
                        float v = acc[0][r];
                        if (a.bias) v += a.bias[row];
                        if (EPI == EPI_RESID) v = a.y[row] + v;    // x += proj (voxtral_decoder.c:676,689)
                        a.y[row] = v;
                    }
                }
            } else if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int r = 0; r < RPW; r++) {
                    const int row = r0 + r;
                    if (row < N) a.y[row] = silu(acc[0][r]) * acc[1][r];  // voxtral_decoder.c:684-687
                }
            } else if constexpr (EPI == EPI_QKV) {
                // rows come in (even, odd) pairs = one RoPE pair (voxtral_kernels.c:502-526)
                const int slot = a.st->pos % a.kv_cap;
                const int qk_rows = a.q_rows + a.k_rows;
#pragma unroll
                for (int r = 0; r < RPW; r += 2) {
                    const int row = r0 + r;
                    if (row + 1 < N) {
                        float o0 = acc[0][r], o1 = acc[0][r + 1];
                        if (row < qk_rows) {
                            const int d = (row % a.head_dim) >> 1;
                            const float c = a.rope[2 * d], s = a.rope[2 * d + 1];
                            const float x0 = o0, x1 = o1;
                            o0 = x0 * c - x1 * s;
                            o1 = x0 * s + x1 * c;
                        }
                        float *dst;
                        if (row < a.q_rows) dst = a.y + row;
                        else if (row < qk_rows) dst = a.kcache + (size_t)slot * a.kv_dim + (row - a.q_rows);
                        else dst = a.vcache + (size_t)slot * a.kv_dim + (row - qk_rows);
                        dst[0] = o0;
                        dst[1] = o1;
                    }
                }
            } else if constexpr (EPI == EPI_LOGITS) {
#pragma unroll
                for (int r = 0; r < RPW; r++) {
                    const int row = r0 + r;
                    if (row < N) {
                        const float v = acc[0][r];
                        a.y[row] = v;
                        // strict '>' scan => lowest index wins ties (voxtral_decoder.c:697-704)
                        if (v > best_v || (v == best_v && row + a.row_base < best_i)) { best_v = v; best_i = row + a.row_base; }
                    }
                }
            }
        }
    }

    if (EPI == EPI_LOGITS) {
        __syncthreads();
        float *rv = red;
        int *ri = reinterpret_cast<int *>(red + 4);
        if (lane == 0) { rv[wave] = best_v; ri[wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = rv[0]; int bi = ri[0];
            for (int w = 1; w < 4; w++)
                if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
            a.blk_val[blockIdx.x] = bv; a.blk_idx[blockIdx.x] = bi;
        }
    }
}


// ---------------------------------------------------------------------------------------
// The decode GEMVs of the 4B shapes (k_gemv3 below).  The per-layer matrices are small (25-113 MB): a launch lasts 5-20 us, so what
// matters is how fast the chip gets from "kernel start" to "every CU has its full share of HBM requests in flight" and that no CU
// gets a second helping while others idle.  Therefore:
//   * one pass per block, no grid-stride loop: grid = N / (rows per block) is chosen per matrix as an exact multiple of the 256 CUs;
//   * KS = 2 splits K across wave pairs for the K = 9216 down-projection with a 2-way LDS reduction;
//   * the RoPE rotation is computed in the QKV epilogue from the position (no table kernel), the step embedding is built in layer
//     0's prologue (no embed kernel), the split-K attention partials are merged in the Wo prologue (no combine kernel).
// (k_gemv2, rounds 1 - 4: the same kernel with all weight loads issued first - +6 us per launch, see k_gemv3 - removed in round 5.)
// ---------------------------------------------------------------------------------------

// bf16 [N, K] -> fp8 e4m3 [N, K] with one f32 scale per row (BASELINE config 5: fp8 decode weights).
// scale = max|w| / 224 keeps every quantised value inside the finite range of both e4m3 flavours;
// block per row.
__global__ __launch_bounds__(256) void k_quant_fp8_rows(const uint16_t *W, uint8_t *Q, float *scale, int K) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const uint16_t *w = W + (size_t)row * K;
    float amax = 0.f;
    for (int k = tid; k < K; k += 256) amax = fmaxf(amax, fabsf(bf16_to_f32(w[k])));
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.f ? amax / 224.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (tid == 0) scale[row] = sc;
    uint32_t *q = reinterpret_cast<uint32_t *>(Q + (size_t)row * K);
    for (int k4 = tid; k4 < K / 4; k4 += 256) {
        const float v0 = bf16_to_f32(w[4 * k4]) * inv, v1 = bf16_to_f32(w[4 * k4 + 1]) * inv;
        const float v2 = bf16_to_f32(w[4 * k4 + 2]) * inv, v3 = bf16_to_f32(w[4 * k4 + 3]) * inv;
        int word = 0;
        word = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, word, false);
        word = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, word, true);
        q[k4] = (uint32_t)word;
    }
}

// Final argmax over the per-block partials + decoder cursor advance.
// One block of 256 threads.  It sits alone on the critical path of every step (nothing overlaps it), so: the decoder state is
// fetched once, up front, under the partials' loads (it used to be three dependent trips to memory behind the reduction), and
// the reduction is one butterfly per wave plus one LDS exchange instead of eight workgroup barriers (7.6 -> ~4 us per step).
__global__ __launch_bounds__(256) void k_argmax_finish(const float *blk_val, const int *blk_idx, int nblk,
                                                       DecState *st, int *tokens_out, int eos_token,
                                                       int advance) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int tid = threadIdx.x;
    DecState cur{};
    if (tid == 0) cur = *st;
    float bv = -3.0e38f; int bi = 0x7fffffff;
    for (int i = tid; i < nblk; i += 256) {
        const float v = blk_val[i]; const int ix = blk_idx[i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(bv, o, 64); const int ix = __shfl_xor(bi, o, 64);
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const float v = sv[w]; const int ix = si[w];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
        // all-NaN / all -inf logits (garbage audio) leave the scan without a winner; the reference's
        // scan (voxtral_decoder.c:697-704) then keeps index 0.  Never hand an out-of-range id to the
        // embedding gather of the next step.
        int tok = bi;
        if (tok < 0 || tok == 0x7fffffff) tok = 0;
        if (!cur.stop) {
            tokens_out[cur.n_out] = tok;
            cur.n_out += 1;
            cur.token = tok;
            if (advance) { cur.pos += 1; cur.adapter_row += 1; }
            if (tok == eos_token) cur.stop = 1;
            *st = cur;
        }
    }
}

// Start of a decode step: RoPE row for the current position and (optionally) the
// step embedding  x = adapter[row] + f32(tok_emb[prev])  (voxtral.c:1057-1061).
// inv_freq[d] = 1/powf(theta, 2d/dim) is computed on the host exactly like
// vox_compute_rope_freqs (voxtral_kernels.c:488-500); angle = (float)pos * inv_freq
// is the same single fp32 multiply, cosf/sinf are the accurate (range-reduced) forms.
__global__ __launch_bounds__(256) void k_step_begin(const DecState *st, const float *inv_freq, int half_dim,
                                                    float *rope, float *x, const float *adapter,
                                                    const uint16_t *tok_emb, int dim, int build_embed) {
    const int tid = threadIdx.x;
    const float p = (float)st->pos;
    for (int d = tid; d < half_dim; d += 256) {
        const float ang = p * inv_freq[d];
        rope[2 * d] = cosf(ang);
        rope[2 * d + 1] = sinf(ang);
    }
    if (build_embed) {
        const float *arow = adapter + (size_t)st->adapter_row * dim;
        const uint16_t *erow = tok_emb + (size_t)st->token * dim;
        for (int i = tid; i < dim; i += 256) x[i] = arow[i] + bf16_to_f32(erow[i]);
    }
}


// ---------------------------------------------------------------------------------------
// k_gemv3 — the decode GEMV with the memory queue in the right order.
//
// Measured on MI355X (profiles/r01_run2_*): a launch that issues its weight loads first pays ~6 us on top of
// bytes / 6.3 TB/s.  A CU's vector-memory path returns loads in issue order, so the x / norm /
// partial loads issued AFTER the 24-36 weight loads per lane come back last: the
// whole prologue (RMSNorm, attention merge) and every FMA then run after the weight stream has
// drained instead of under it.  Here the order is
//   1. epilogue operands (residual rows, position, RoPE frequencies)          [registers]
//   2. the activation vector, norm weights, ada scale                          [LDS-DMA, no VGPRs]
//      (attention partials of the Wo prologue: registers, issued in step 1)
//   3. all weight loads, piece-major so that consumption order = arrival order [registers, nt]
//   4. s_waitcnt vmcnt(#weight loads): (1) and (2) are in, the weights still stream
//   5. prologue math in LDS, then the dot products piece by piece as the weights land.
// K is a template constant (CPL * KS * 512).
// ---------------------------------------------------------------------------------------
template <int PRO, int EPI, int RPW, int CPL, int KS, int MINW, bool W8 = false>
__global__ __launch_bounds__(256, MINW) void k_gemv3(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int EPP = W8 ? 16 : 8;  // weights per 16-byte piece (fp8 e4m3 : bf16)
    constexpr int K = CPL * KS * 64 * EPP;
    constexpr int NX = K / 1024;      // 16-byte pieces of a K-float vector per thread
    constexpr int NMAT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int RG = 4 / KS;        // row groups (waves along rows) per block
    constexpr int NW = NMAT * RPW * CPL;
    float *xs = smem;                 // [K]
    float *red = smem + K;            // [64]: wave sums, KS partials
    float *aux = smem + K + 64;       // PRO_RMS: norm_w [K], ada [K];  PRO_ATTN: scl [heads][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave / KS, kp = wave % KS;
    const int N = a.N;
    const int row0 = (blockIdx.x * RG + rg) * RPW;
    const int ns = (PRO == PRO_ATTN) ? a.nsplit : 0;
    const unsigned long long tl0 = tl_begin(a.tl);

    // ---- operand loads, as lambdas so that the two launch modes can order them ----------------
    float yv[RPW];
    float fr[(RPW + 1) / 2];
    int pos = 0;
    auto load_epilogue_operands = [&]() {      // never produced by the immediate predecessor
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < RPW; r++) yv[r] = a.y[min(row0 + r, N - 1)];
        }
        if constexpr (EPI == EPI_QKV) {
            pos = a.pos_host;              // the host knows the position of every step it enqueues
#pragma unroll
            for (int r = 0; r < RPW; r += 2) fr[r / 2] = a.inv_freq[((row0 + r) % a.head_dim) >> 1];
        }
    };
    float4 ev[PRO == PRO_EMBED_RMS ? NX : 1];
    uint2 eb[PRO == PRO_EMBED_RMS ? NX : 1];
    float4 po[PRO == PRO_ATTN ? NX : 1][PRO == PRO_ATTN ? 8 : 1];
    float2 pml[PRO == PRO_ATTN ? 8 : 1];
    auto load_activation_registers = [&]() {   // step embedding / attention partials -> registers
        if constexpr (PRO == PRO_EMBED_RMS) {
            const float *arow = a.adapter + (size_t)a.st->adapter_row * K;
            const uint16_t *erow = a.tok_emb + (size_t)a.st->token * K;
#pragma unroll
            for (int j = 0; j < NX; j++) {
                ev[j] = *reinterpret_cast<const float4 *>(arow + j * 1024 + tid * 4);
                eb[j] = *reinterpret_cast<const uint2 *>(erow + j * 1024 + tid * 4);
            }
        }
        if constexpr (PRO == PRO_ATTN) {
            // [NX pieces][<= 8 slices] of the split-K partials this thread merges and the (max, sum)
            // pairs of head `tid`; predicated on the slice count; registers, not LDS, so that the
            // 8-slice case does not need 128 KB of it.
            const int HD = a.attn_hd;
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                if (sidx < ns) {
                    if (tid < K / HD) pml[sidx] = *reinterpret_cast<const float2 *>(a.part_ml + ((size_t)tid * ns + sidx) * 2);
#pragma unroll
                    for (int j = 0; j < NX; j++) {
                        const int i = j * 1024 + tid * 4;
                        const int h = i / HD, d = i - h * HD;
                        po[j][sidx] = *reinterpret_cast<const float4 *>(a.part_o + ((size_t)h * ns + sidx) * HD + d);
                    }
                }
            }
        }
    };
    auto issue_lds_dma = [&]() {               // activation vector, norm weights, ada -> LDS
        const unsigned wofs = (unsigned)wave * 1024u;            // this wave's 1 KiB of every 4 KiB slab
        if constexpr (PRO == PRO_RMS || PRO == PRO_NONE) {
#pragma unroll
            for (int j = 0; j < NX; j++) glds16(a.x + j * 1024 + tid * 4, lds_addr(xs) + j * 4096u + wofs);
        }
        if constexpr (PRO == PRO_RMS || PRO == PRO_EMBED_RMS) {
#pragma unroll
            for (int j = 0; j < NX; j++) glds16(a.norm_w + j * 1024 + tid * 4, lds_addr(aux) + j * 4096u + wofs);
            if (a.ada) {
#pragma unroll
                for (int j = 0; j < NX; j++) glds16(a.ada + j * 1024 + tid * 4, lds_addr(aux + K) + j * 4096u + wofs);
            }
        }
    };
    uint4 w[NMAT][RPW][CPL];
    auto issue_weight_loads = [&]() {          // every weight byte of this wave, piece-major
        const uint4 *p0[RPW];
        const uint4 *p1[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = min(row0 + r, N - 1);
            constexpr size_t ROWB = (size_t)K * (W8 ? 1 : 2);           // bytes per weight row
            p0[r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.W) + (size_t)row * ROWB) + kp * (CPL * 64) + lane;
            if constexpr (NMAT == 2)
                p1[r] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.W2) + (size_t)row * ROWB) + kp * (CPL * 64) + lane;
        }
#pragma unroll
        for (int c = 0; c < CPL; c++) {
#pragma unroll
            for (int r = 0; r < RPW; r++) w[0][r][c] = ld_stream(p0[r] + c * 64);
            if constexpr (NMAT == 2) {
#pragma unroll
                for (int r = 0; r < RPW; r++) w[1][r][c] = ld_stream(p1[r] + c * 64);
            }
        }
    };

    // Everything this kernel reads is final (in-order launches).  Activation side first (it is
    // needed first and a CU's memory path is in-order), then the weight stream.
    load_epilogue_operands();
    load_activation_registers();
    __builtin_amdgcn_sched_barrier(0);
    issue_lds_dma();
    __builtin_amdgcn_sched_barrier(0);
    issue_weight_loads();
    __builtin_amdgcn_sched_barrier(0);
    // the older loads have landed once at most NW (the weights) are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");

    // ---- prologue math (under the weight stream) ---------------------------------
    if constexpr (PRO == PRO_EMBED_RMS) {
#pragma unroll
        for (int j = 0; j < NX; j++) {
            float4 v = ev[j];
            v.x += bf16_lo(eb[j].x); v.y += bf16_hi(eb[j].x); v.z += bf16_lo(eb[j].y); v.w += bf16_hi(eb[j].y);
            *reinterpret_cast<float4 *>(xs + j * 1024 + tid * 4) = v;
            if (blockIdx.x == 0) *reinterpret_cast<float4 *>(a.x_out + j * 1024 + tid * 4) = v;
        }
    }
    __syncthreads();
    if constexpr (PRO == PRO_RMS || PRO == PRO_EMBED_RMS) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + j * 1024 + tid * 4);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + a.eps);
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const float4 g = *reinterpret_cast<const float4 *>(aux + j * 1024 + tid * 4);
            float4 o = *reinterpret_cast<const float4 *>(xs + j * 1024 + tid * 4);   // own elements: no hazard
            o.x = o.x * inv * g.x; o.y = o.y * inv * g.y; o.z = o.z * inv * g.z; o.w = o.w * inv * g.w;
            if (a.ada) {
                const float4 sc = *reinterpret_cast<const float4 *>(aux + K + j * 1024 + tid * 4);
                o.x *= (1.0f + sc.x); o.y *= (1.0f + sc.y); o.z *= (1.0f + sc.z); o.w *= (1.0f + sc.w);
            }
            *reinterpret_cast<float4 *>(xs + j * 1024 + tid * 4) = o;
        }
        __syncthreads();
    }
    if constexpr (PRO == PRO_ATTN) {
        // x[h*HD + d] = sum_s o[h][s][d] * exp(m_s - m) / sum_s l_s exp(m_s - m)
        float *scl = aux;                               // [heads][8]
        const int HD = a.attn_hd, heads = K / HD;
        if (tid < heads) {
            float mm = -1e30f;
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) if (sidx < ns) mm = fmaxf(mm, pml[sidx].x);
            float ll = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) if (sidx < ns) ll += pml[sidx].y * expf(pml[sidx].x - mm);
            const float inv = ll > 0.f ? 1.0f / ll : 0.f;
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) if (sidx < ns) scl[tid * 8 + sidx] = expf(pml[sidx].x - mm) * inv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const int i = j * 1024 + tid * 4;
            const int h = i / HD;
            float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                if (sidx < ns) {
                    const float f = scl[h * 8 + sidx];
                    const float4 o = po[j][sidx];
                    acc4.x += o.x * f; acc4.y += o.y * f; acc4.z += o.z * f; acc4.w += o.w * f;
                }
            }
            *reinterpret_cast<float4 *>(xs + i) = acc4;
        }
        __syncthreads();
    }
    float cs[(RPW + 1) / 2], sn[(RPW + 1) / 2];
    if constexpr (EPI == EPI_QKV) {
        const float fpos = (float)pos;
#pragma unroll
        for (int r = 0; r < RPW; r += 2) {
            const float ang = fpos * fr[r / 2];
            cs[r / 2] = cosf(ang);
            sn[r / 2] = sinf(ang);
        }
    }

    // ---- dot products, in arrival order ----------------------------------------------------------
    float acc[NMAT][RPW];
#pragma unroll
    for (int m = 0; m < NMAT; m++)
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[m][r] = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const float *xp = xs + ((kp * CPL + c) * 64 + lane) * EPP;
        const float4 x0 = *reinterpret_cast<const float4 *>(xp);
        const float4 x1 = *reinterpret_cast<const float4 *>(xp + 4);
        if constexpr (W8) {
            const float4 x2 = *reinterpret_cast<const float4 *>(xp + 8);
            const float4 x3 = *reinterpret_cast<const float4 *>(xp + 12);
#pragma unroll
            for (int m = 0; m < NMAT; m++)
#pragma unroll
                for (int r = 0; r < RPW; r++) acc[m][r] = dot16_fp8(w[m][r][c], x0, x1, x2, x3, acc[m][r]);
        } else {
#pragma unroll
            for (int m = 0; m < NMAT; m++)
#pragma unroll
                for (int r = 0; r < RPW; r++) acc[m][r] = dot8_bf16(w[m][r][c], x0, x1, acc[m][r]);
        }
    }
#pragma unroll
    for (int m = 0; m < NMAT; m++)
#pragma unroll
        for (int r = 0; r < RPW; r++) acc[m][r] = wave_sum(acc[m][r]);
    if constexpr (W8) {                                // per-row dequantisation scale
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int row = min(row0 + r, N - 1);
            acc[0][r] *= a.wscale[row];
            if constexpr (NMAT == 2) acc[1][r] *= a.wscale2[row];
        }
    }
    if constexpr (KS == 2) {
        float *part = red + 16;                        // [RG][NMAT*RPW]
        if (kp == 1 && lane == 0) {
#pragma unroll
            for (int m = 0; m < NMAT; m++)
#pragma unroll
                for (int r = 0; r < RPW; r++) part[rg * (NMAT * RPW) + m * RPW + r] = acc[m][r];
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int m = 0; m < NMAT; m++)
#pragma unroll
                for (int r = 0; r < RPW; r++) acc[m][r] += part[rg * (NMAT * RPW) + m * RPW + r];
        }
    }

    // ---- epilogue (no loads left) ------------------------------------------------------------------
    auto put = [&](float *p, float v) { *p = v; };
    if (lane == 0 && kp == 0) {
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int row = row0 + r;
                if (row < N) put(a.y + row, yv[r] + acc[0][r]);
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int row = row0 + r;
                if (row < N) put(a.y + row, silu(acc[0][r]) * acc[1][r]);
            }
        } else if constexpr (EPI == EPI_QKV) {
            const int slot = pos % a.kv_cap;
            const int qk_rows = a.q_rows + a.k_rows;
#pragma unroll
            for (int r = 0; r < RPW; r += 2) {
                const int row = row0 + r;
                if (row + 1 < N) {
                    float o0 = acc[0][r], o1 = acc[0][r + 1];
                    if (row < qk_rows) {
                        const float x0 = o0, x1 = o1;
                        o0 = x0 * cs[r / 2] - x1 * sn[r / 2];
                        o1 = x0 * sn[r / 2] + x1 * cs[r / 2];
                    }
                    float *dst;
                    if (row < a.q_rows) dst = a.y + row;
                    else if (row < qk_rows) dst = a.kcache + (size_t)slot * a.kv_dim + (row - a.q_rows);
                    else dst = a.vcache + (size_t)slot * a.kv_dim + (row - qk_rows);
                    put(dst, o0);
                    put(dst + 1, o1);
                }
            }
        }
    }
    tl_end(a.tl, tl0);
}

}  // namespace vox
