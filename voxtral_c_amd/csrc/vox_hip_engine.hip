// vox_hip_engine.hip — MI355X (gfx950) device engine behind the C-ABI in include/vox_hip.h.
//
// One engine = one GPU, one HIP stream, one active transcription stream (the reference is
// single-stream too: voxtral.c:1227).  All weights stay resident in HBM as the bf16 bytes of
// the safetensors file (QKV and W1;W3 of a layer are placed back to back so one launch
// covers them); activations, both KV windows, the mel queue, the conv-stem boundary state
// and the adapter rows live in HBM for the whole session — the host only moves audio
// samples in and token ids out.
//
// Hot path = three device submissions, exactly where the reference crosses CPU->GPU in
// its Metal build: encoder chunk, decoder prefill, decoder step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <vector>
#include <string>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>

#include "../../include/vox_hip.h"
#include "vox_common.h"
#include "vox_gemv.h"
#include "vox_gemm.h"
#include "vox_gemm_planes.h"
#include "vox_misc.h"
#include "vox_attn.h"
#include "vox_kernel_api.h"
#include "vox_decfuse.h"
#include "vox_skinny.h"
#include "vox_encstack.h"
#include "vox_rowsgemm.h"
#include "vox_rowsgemm_f8.h"

using namespace vox;

static thread_local std::string g_err;
static void set_err(const char *what, hipError_t e, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "vox_hip: %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
    fprintf(stderr, "%s\n", buf);
    (void)hipGetLastError();      // the runtime's last-error word is sticky: leave it clean for the per-step launch checks
}
#define HC(call)                                                        \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) { set_err(#call, _e, __FILE__, __LINE__); return -1; } \
    } while (0)
#define HCV(call)                                                       \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) { set_err(#call, _e, __FILE__, __LINE__); } \
    } while (0)

// One check per enqueued unit of work (decode step, encoder chunk, prefill): a refused launch
// (bad grid, LDS request, ...) must surface as an error return, not as a wrong token at the next sync.
static int launch_check(const char *what, const char *file, int line) {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { set_err(what, le, file, line); return -1; }
    return 0;
}
#define LAUNCH_CHECK(what) do { if (launch_check(what, __FILE__, __LINE__)) return -1; } while (0)

namespace {

constexpr int DEC_SPLIT_KEYS = 128;     // keys per attention block in the decoder
constexpr int DEC_RING_EXTRA = 1024;    // ring = window + 1024 rows (reference allocates 8192+seq+1024, voxtral_decoder.c:422)
constexpr int PREFILL_CHUNK = 512;
constexpr int MAX_RUN_STEPS = 4096;

struct EncLayer {
    uint16_t *wqkv = nullptr, *wo = nullptr, *w13 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b2 = nullptr, *n1 = nullptr, *n2 = nullptr;
    float *kring = nullptr, *vring = nullptr;
};
struct DecLayer {
    uint16_t *wqkv = nullptr, *wo = nullptr, *w13 = nullptr, *w2 = nullptr;
    // optional fp8 (e4m3) copies of the four matrices with one f32 scale per output row
    uint8_t *wqkv8 = nullptr, *wo8 = nullptr, *w138 = nullptr, *w28 = nullptr;
    float *sqkv = nullptr, *so = nullptr, *s13 = nullptr, *s2 = nullptr;
    // optional bf16 copies holding dequantised block-scaled fp8 values (vox_hip_simulate_block_fp8: agreement studies)
    uint16_t *wqkv_s = nullptr, *wo_s = nullptr, *w13_s = nullptr, *w2_s = nullptr;
    float *n1 = nullptr, *n2 = nullptr, *ada = nullptr;
    float *kring = nullptr, *vring = nullptr;
};

struct Buf {   // growable device scratch
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Weight ingest: the checkpoint is an mmap of the page cache.  hipMemcpy from pageable memory copies through
// the runtime's own staging on one thread (27 GB/s measured for the 8.86 GB checkpoint); here a small pool of threads copies
// 8 MB pieces into pinned slots (page-cache faults and memcpy in parallel) and every filled slot goes out with hipMemcpyAsync
// on its own stream while the next one is being filled.  The engine's compute stream is ordered behind the copies with an
// event, so nothing has to wait on the host except a slot that is about to be reused.
struct Uploader {
    static constexpr int NSLOT = 4;
    static constexpr size_t SLOT = (size_t)8 << 20;
    void *pin[NSLOT] = {};
    hipEvent_t ev[NSLOT] = {};
    bool used[NSLOT] = {};
    hipStream_t st = nullptr;
    hipEvent_t last = nullptr;
    int next = 0, nthreads = 0;
    bool ok = false;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    struct Job { const char *src; char *dst; size_t n; };
    std::vector<Job> jobs;
    int pending = 0;
    bool quit = false;

    bool start() {
        if (ok) return true;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&last, hipEventDisableTiming) != hipSuccess) return false;
        for (int i = 0; i < NSLOT; i++)
            if (hipHostMalloc(&pin[i], SLOT, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
        nthreads = (int)std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 4));
        for (int t = 0; t < nthreads; t++)
            workers.emplace_back([this] {
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_job.wait(lk, [this] { return quit || !jobs.empty(); });
                        if (jobs.empty()) return;
                        j = jobs.back(); jobs.pop_back();
                    }
                    memcpy(j.dst, j.src, j.n);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
        ok = true;
        return true;
    }
    // dst (device) <- src (pageable host), bytes; returns 0, or -1 (the caller then falls back to a plain hipMemcpy)
    int copy(void *dst, const void *src, size_t bytes) {
        for (size_t off = 0; off < bytes; off += SLOT) {
            const size_t n = std::min(SLOT, bytes - off);
            const int sl = next; next = (next + 1) % NSLOT;
            if (used[sl] && hipEventSynchronize(ev[sl]) != hipSuccess) return -1;
            {
                std::lock_guard<std::mutex> lk(mu);
                const size_t piece = std::max<size_t>((n + nthreads - 1) / nthreads, (size_t)1 << 20);
                for (size_t o = 0; o < n; o += piece) { jobs.push_back(Job{(const char *)src + off + o, (char *)pin[sl] + o, std::min(piece, n - o)}); pending++; }
            }
            cv_job.notify_all();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_done.wait(lk, [this] { return pending == 0; });
            }
            if (hipMemcpyAsync((char *)dst + off, pin[sl], n, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
            if (hipEventRecord(ev[sl], st) != hipSuccess) return -1;
            used[sl] = true;
        }
        return 0;
    }
    // every copy issued so far is ordered before later work on `compute`
    int fence(hipStream_t compute) {
        if (!ok) return 0;
        if (hipEventRecord(last, st) != hipSuccess || hipStreamWaitEvent(compute, last, 0) != hipSuccess) return -1;
        return 0;
    }
    int flush() { return (!ok || hipStreamSynchronize(st) == hipSuccess) ? 0 : -1; }
    void stop() {
        if (st) hipStreamSynchronize(st);
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_job.notify_all();
        for (auto &w : workers) w.join();
        workers.clear();
        for (int i = 0; i < NSLOT; i++) { if (pin[i]) hipHostFree(pin[i]); if (ev[i]) hipEventDestroy(ev[i]); pin[i] = nullptr; ev[i] = nullptr; }
        if (last) hipEventDestroy(last);
        if (st) hipStreamDestroy(st);
        last = nullptr; st = nullptr; ok = false;
    }
};

struct vox_hip_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    vox_hip_dims_t d{};
    int enc_qd = 0, dec_qd = 0, dec_kvd = 0;
    size_t mem_used = 0;
    bool use_skinny = true;
    bool use_rowsgemm = true;     // 33 .. 128-row chunks (decoder prefill, encoder flush pass) on k_rowsgemm (vox_rowsgemm.h)
    bool use_dpp = true, use_mfma = true, use_fast = true, use_splitk = true, use_bf16x3 = true, use_attn_mfma = true;

    // weights
    uint16_t *tok_emb = nullptr, *conv0_w = nullptr, *conv1_w = nullptr, *adapter0 = nullptr, *adapter1 = nullptr;
    float *conv0_b = nullptr, *conv1_b = nullptr, *enc_final_norm = nullptr, *dec_final_norm = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    // mel tables
    float *hann = nullptr, *cosT = nullptr, *sinT = nullptr, *filtT = nullptr;
    // rope
    float *enc_inv_freq = nullptr, *dec_inv_freq = nullptr, *dec_rope = nullptr;

    // encoder stream state
    int enc_ring_cap = 0;
    int enc_pos = 0;            // logical positions already encoded (= enc_kv_pos_offset + enc_kv_cache_len)
    int mel_q = 0;              // frames waiting in conv_in0 rows [2, 2+mel_q)
    int c0_carry = 0;           // 0/1 conv0 frame waiting for its stride-2 partner (voxtral.c:612-656)
    int enc_res = 0;            // 0..3 encoder rows waiting for 4x alignment (voxtral.c:824-890)
    Buf conv_in0, conv_in1, enc_out;   // see conv stem / alignment notes below
    // large-M scratch
    Buf sx, sxn, sqkv, sattn, sgu, sh, srope, sim2col, ssamples, smid, stmp_in, stmp_out, spart_o, spart_ml, ssplitk;

    // adapter rows (linear buffer; physical row r <-> logical row adapter_row0 + r)
    float *adapter = nullptr;
    int64_t adapter_cap = 0, adapter_row0 = 0, adapter_total = 0, adapter_consumed = 0;

    // decoder
    int dec_ring_cap = 0;
    int dec_pos = 0;            // logical positions stored (= kv_pos_offset + kv_cache_len)
    DecState *d_st = nullptr;
    float *dx2 = nullptr;               // second residual-stream buffer of the fused decode step (see enqueue_step)
    float *dx = nullptr, *dq = nullptr, *dattn = nullptr, *dh = nullptr, *dlogits = nullptr;
    float *blk_val = nullptr; int *blk_idx = nullptr; int logits_grid = 0;
    unsigned skip_kinds = 0;            // timing experiments only: PK_* launches left out of a step
    bool use_fp8 = false;               // decode GEMVs stream the fp8 copies (vox_hip_quantize_decoder_fp8)
    uint8_t *tok_emb8 = nullptr; float *stok = nullptr;
    uint16_t *tok_emb_s = nullptr;      // simulated-quantisation copy of the LM head (see DecLayer::wqkv_s)
    bool sim_on = false, sim_lm = false;
    bool fp8_prefill_bf16 = false;       // VOX_HIP_DISABLE=fp8_prefill: fp8 mode with the prefill on the bf16 matrices (rounds 2 - 4)
    unsigned *d_f8_clamped = nullptr, *h_f8_clamped = nullptr;     // k_rowsgemm_f8: activations beyond the e4m3 range after the fixed prescale (device counter, pinned mirror)
    int fp8_prefill_fallbacks = 0;
    bool fp8_attn_bf16 = false, fp8_lmhead_bf16 = false;     // A/B and agreement-study switches of the fp8 mode, read at creation
    // fused attention half of the decode step (vox_decfuse.h)
    bool use_fused = false;
    bool fused_ok = false;        // the fused kernels exist for this geometry / device (use_fused may be suspended after a time-out)
    long fuse_rearm = 0;          // clean decode steps on the chain before the fused kernel is tried again (0 = not suspended)
    u64 *d_gq = nullptr, *d_gp = nullptr, *d_gh = nullptr;
    int merge12 = 2;              // 2 = k_ffn_attn12 (FFN block of layer l + attention block of layer l + 1 in one launch); 0 (VOX_HIP_DISABLE=merge12) = two launches per layer in the 8-wave shape
    u64 *d_gx = nullptr;          // [3072] x'' hand-off of k_ffn_attn12
    int use_stack = 1;            // 0 (VOX_HIP_DISABLE=stack): one k_ffn_attn12 launch per layer (round 4) instead of ONE k_dec_stack launch for all layers' blocks
    u64 *d_gw = nullptr, *d_gxp = nullptr;   // k_dec_stack: [8][3072] Wo partial sums, [3072] x' (hand-off granules)
    DecStackLayer *d_stack_tab = nullptr;    // per-layer pointers of k_dec_stack (device copy of h_stack_tab)
    std::vector<DecStackLayer> h_stack_tab;
    int merge12_long = 1;         // 0 (VOX_HIP_DISABLE=merge12_long): beyond merge12_maxkeys two launches per layer in the 8-wave shape (round 4); 1 = k_ffn_attn12<LONG> (9 .. 32 key slices)
    static constexpr int merge12_maxkeys = 1024;      // 8 key slices up to this context length (two tiles per member beyond 512 keys; measured: 2048 = four tiles per member loses 2 % at 1900 keys)
    float *d_xprime = nullptr;    // x' of the fused FFN launch, written only for the debug taps
    bool use_ffn = false;         // FFN block as one launch (k_ffn_fused) instead of k_gemv_w13x + k_gemv_w2x
    float *d_wo_part = nullptr;
    unsigned *d_fuse_err = nullptr;
    unsigned fuse_epoch = 0;
    unsigned spin_hole_max = 0; unsigned long long spin_holes = 0;   // holes (> 1 ms between two polls) the bounded spins saw: err[8..9], vox_decfuse.h
    bool use_planes = true;       // large-M GEMMs on pre-split bf16 planes (k_gemm_planes)
    bool use_epi = true, use_attn_small = true, use_staged_upload = true;     // A/B switches, read once per engine (self_test)
    // k_gemm_planes with the weight fragments straight from global memory to registers (VOX_HIP_GP_BDIRECT=1).  Measured and NOT
    // kept as the default: 1627-row pass, qkv 114.8 -> 137.7 us, w1;w3 158 -> 190 us (gpurun_out/p13) - the fragment-layout loads
    // (32 bytes from each of 32 rows per instruction, issued twice: both row halves of the tile need them) cost more than the
    // quarter of the LDS-DMA transport they take away.
    Buf splanes;                  // [3][n][max(D, QD, H)] bf16
    Buf es_carry;                 // copy of enc_out's carried rows while a stack-kernel chunk may have to be repeated
    Uploader *up = nullptr;       // staged weight ingest: lives until vox_hip_upload_done (vox_load calls it) or the engine's end
    bool attn_merge = true;                 // VOX_HIP_DISABLE=attn_merge: k_attn_combine as a launch of its own (round 4)
    unsigned *d_attn_arrive = nullptr;      // k_attn_small: per-head arrival counters (zero between launches)
    // Round 6: the few-rows encoder chunk as ONE persistent launch (k_enc_stack, vox_encstack.h).  enc_stack_ok = the part and the
    // geometry fit (256 CUs, 4B encoder shapes) and it is not switched off; enc_stack_rearm > 0 = suspended after a hand-off time-out
    // (chunks left on the 8-launch path before it is tried again).
    bool enc_stack_ok = false, enc_stack_ready = false, enc_stack_pending = false, enc_stack_now = false;
    int enc_stack_failures = 0; long enc_stack_rearm = 0; long enc_stack_launches = 0; bool enc_stack_inject = false;
    unsigned enc_epoch = 0;
    EncStackLayer *d_es_tab = nullptr;
    float *d_es_xa = nullptr, *d_es_xb = nullptr, *d_es_ssq = nullptr, *d_es_q = nullptr, *d_es_po = nullptr, *d_es_pml = nullptr,
          *d_es_wop = nullptr, *d_es_w2p = nullptr;
    uint16_t *d_es_apl = nullptr, *d_es_hpl = nullptr;
    unsigned *d_es_flags = nullptr, *d_es_err = nullptr, *h_es_err = nullptr;
    // Pinned host memory of the per-feed hot path (round 6): what a feed hands to the device (the decode state, its samples) and gets back
    // (state, error words, token ids) goes through here, so that every copy is asynchronous and a decoder run waits ONCE.
    struct HostPin {
        DecState st_in[2]; DecState st_out;
        unsigned fuse_err[16];
        int tokens[MAX_RUN_STEPS];
    } *pin = nullptr;
    int pin_st_next = 0, pin_st_inflight = 0;      // set_state alternates two slots; a third call without a host wait in between waits first
    static constexpr int SMP_SLOTS = 4; static constexpr size_t SMP_SLOT_BYTES = (size_t)256 << 10;
    void *smp_pin[SMP_SLOTS] = {}; hipEvent_t smp_ev[SMP_SLOTS] = {}; bool smp_used[SMP_SLOTS] = {}; int smp_next = 0;
    unsigned long long *d_es_tl = nullptr;
    bool enc_tl_on = false;
    unsigned long long *d_enc_tl = nullptr;       // VOX_HIP_ENC_TL: [4 GEMM launches][1024 workgroups][16] timeline of one few-rows encoder layer
    unsigned long long *d_fuse_tl = nullptr;      // VOX_HIP_FUSE_TL: [3 kernels][1024 workgroups][3] timeline of the layer-13 launches
    int fuse_failures = 0;
    int *d_tokens = nullptr;
    float *dpart_o = nullptr, *dpart_ml = nullptr;   // decode-step split-K partials (max splits)
    int dec_max_split = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    vox_hip_timing_t timing{};
    // multi-GPU shard in flight (vox_hip_shard_*), events for cross-engine stream ordering
    std::vector<hipEvent_t> xev; size_t xev_next = 0;
    // Round 4: work other engines still have in flight FOR this engine (a sharded chunk, vox_multi.c).  Instead of making this
    // engine's one stream wait for all of it at once, the waits are placed where the data is first touched: the decoder waits
    // for a shard's adapter rows when it reaches them (row_fences, ascending), the encoder side waits for the handed-over stream
    // state before it touches encoder state again (enc_fence).  So the decoder starts on the first shard's rows while the
    // later shards are still encoding.
    struct RowFence { int64_t first_row; hipEvent_t ev; };
    std::vector<RowFence> row_fences;
    std::vector<hipEvent_t> enc_fences;
    float *shard_x = nullptr; int shard_n = 0;
    // per-kernel profiling of the decode step (HIP events between launches)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_kind;
    size_t prof_used = 0;
    unsigned long long n_host_syncs = 0;      // esync() calls (tests: no host wait inside a sharded wavefront)
    // debug taps of the decoder's residual stream (vox_hip_debug_tap_config): at the decode steps whose KV position is listed,
    // x at the start of every layer, x after every attention block and x after the last layer are copied to d_taps in stream order
    std::vector<int> tap_pos; float *d_taps = nullptr;
    // Round 4: L2 prefetch of the next launch's first weight bytes (vox_decfuse.h, DfPrefetch).  (the VOX_HIP_PF override of round 4 is gone: the
    // measured default is a constant)
    // Default (measured): 24 KiB per target block (6.3 MB per launch) issued by the non-members in front of their Wo rows.
    static constexpr int pf_units = 24, pf_member_units = 0, pf_when = 3;      // DfPrefetch (vox_decfuse.h): 24 KiB per target block, by the non-members, in front of their Wo rows
};

// Every host-side wait on the engine stream goes through here and is counted (vox_hip_host_syncs): the multi-GPU
// wavefront must not synchronise between shard_begin and shard_end, and the tests check that with this counter.
static hipError_t esync(vox_hip_engine *e) {
    e->n_host_syncs++;
    e->pin_st_inflight = 0;
    return hipStreamSynchronize(e->stream);
}

// VOX_HIP_DISABLE=name[,name..]: the ONE switch of the fallback ladder.  Every name switches one production kernel family off, so
// that the next older HIP path underneath runs instead (never a CPU path) - what a failed start-up self-test does by itself.  For
// A/B measurements and for the tests that keep the older paths honest.  Names: fused (the launch-per-GEMV decode chain), ffn_fused,
// merge12 (two launches per layer), merge12_long (two launches per layer beyond 1024 keys), stack (one launch per layer),
// fast (the generic decode kernels), dpp, mfma, bf16x3, planes, splitk, skinny, rowsgemm, attn_small, attn_merge (k_attn_combine as a launch of its own), attn_mfma, epi (separate RoPE / SiLU launches), staged_upload, rearm (a timed-out fused kernel stays off),
// fp8_attn / fp8_lmhead / fp8_prefill (fp8 mode: these matrices / this pass stay bf16), multi_overlap (multi-GPU: wait for the whole wavefront).
static bool vox_disabled(const char *name) {
    const char *v = getenv("VOX_HIP_DISABLE");
    if (!v) return false;
    const size_t n = strlen(name);
    for (const char *p = v; *p;) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len == n && !strncmp(p, name, n)) return true;
        if (!q) break;
        p = q + 1;
    }
    return false;
}
extern "C" int vox_hip_switch_disabled(const char *name) { return name && vox_disabled(name) ? 1 : 0; }

// Stream-side waits for other engines' pending work (see vox_hip_engine::row_fences).
static int apply_row_fences(vox_hip_engine *e, int64_t upto_row) {         // rows [.., upto_row] are about to be read on e->stream
    while (!e->row_fences.empty() && e->row_fences.front().first_row <= upto_row) {
        HC(hipStreamWaitEvent(e->stream, e->row_fences.front().ev, 0));
        e->row_fences.erase(e->row_fences.begin());
    }
    return 0;
}
static int apply_enc_fences(vox_hip_engine *e) {                            // encoder state is about to be touched on e->stream
    for (hipEvent_t ev : e->enc_fences) HC(hipStreamWaitEvent(e->stream, ev, 0));
    e->enc_fences.clear();
    return 0;
}
static int drain_fences(vox_hip_engine *e) {                                // host wait: everything anybody still owes this engine
    for (auto &f : e->row_fences) HC(hipEventSynchronize(f.ev));
    for (hipEvent_t ev : e->enc_fences) HC(hipEventSynchronize(ev));
    e->row_fences.clear(); e->enc_fences.clear();
    return 0;
}

enum { PK_BEGIN = 0, PK_QKV, PK_ATTN, PK_COMBINE, PK_WO, PK_SWIGLU, PK_W2, PK_LOGITS, PK_ARGMAX, PK_COUNT };

static void prof_mark(vox_hip_engine *e, int kind) {
    if (!e->prof_on) return;
    if (e->prof_used == e->prof_ev.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        e->prof_ev.push_back(ev);
        e->prof_kind.push_back(kind);
    }
    e->prof_kind[e->prof_used] = kind;
    hipEventRecord(e->prof_ev[e->prof_used++], e->stream);
}

// ------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------
static int dmalloc(vox_hip_engine *e, void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HC(hipMalloc(p, bytes));
    e->mem_used += bytes;
    return 0;
}
template <typename T>
static int dalloc(vox_hip_engine *e, T **p, size_t n) { return dmalloc(e, (void **)p, n * sizeof(T)); }

// A buffer that grows is freed: peer copies other engines still owe this one (vox_hip_encoder_state_push: K/V rings, conv
// history rows, enc_out rows, recorded as enc_fences and otherwise only waited for by the next encoder-side launch) must have
// landed first, or they would write into freed memory and the copied-over state rows would be stale.
static int settle_enc_fences(vox_hip_engine *e) {
    for (hipEvent_t ev : e->enc_fences) HC(hipEventSynchronize(ev));
    e->enc_fences.clear();
    return 0;
}
static int ensure(vox_hip_engine *e, Buf &b, size_t bytes) {
    if (b.bytes >= bytes) return 0;
    size_t nb = std::max(bytes, b.bytes * 3 / 2);
    void *np = nullptr;
    if (settle_enc_fences(e)) return -1;
    HC(esync(e));
    HC(hipMalloc(&np, nb));
    if (b.p) { HC(hipFree(b.p)); e->mem_used -= b.bytes; }
    b.p = np; b.bytes = nb; e->mem_used += nb;
    return 0;
}
// ensure that keeps the first `keep` bytes (state-carrying buffers)
static int ensure_keep(vox_hip_engine *e, Buf &b, size_t bytes, size_t keep) {
    if (b.bytes >= bytes) return 0;
    size_t nb = std::max(bytes, b.bytes * 3 / 2);
    void *np = nullptr;
    if (settle_enc_fences(e)) return -1;
    HC(esync(e));
    HC(hipMalloc(&np, nb));
    HC(hipMemset(np, 0, nb));
    if (b.p) {
        if (keep) HC(hipMemcpy(np, b.p, std::min(keep, b.bytes), hipMemcpyDeviceToDevice));
        HC(hipFree(b.p)); e->mem_used -= b.bytes;
    }
    b.p = np; b.bytes = nb; e->mem_used += nb;
    return 0;
}

static inline int grid1d(size_t work, int per_block = 256, int cap = 4096) {
    size_t g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// ------------------------------------------------------------------------------------
// GEMM / GEMV launchers
// ------------------------------------------------------------------------------------
static int ensure(vox_hip_engine *e, Buf &b, size_t bytes);
// mode: 0 = best available, 1 = scalar reference kernel, 2 = f32-input MFMA kernel
static int launch_gemm(vox_hip_engine *e, const float *X, int ldx, const uint16_t *W, float *Y, int ldy,
                       int M, int N, int K, const float *bias, const float *resid, int ldr, int act,
                       int mode = 0) {
    GemmArgs a{X, ldx, W, Y, ldy, M, N, K, bias, resid, ldr, act, 1, 0, nullptr};
    if (M <= 0 || N <= 0) return 0;
    const bool aligned = (K % GB_K == 0) && (ldx % 4 == 0) && ((size_t)X % 16 == 0);
    if (e->use_mfma && aligned && mode != 1) {
        // bf16 matrix pipe (3 exact bf16 terms per f32 activation) when K allows 64-wide slices
        const bool x3 = e->use_bf16x3 && mode == 0 && (K % GX_K == 0);
        const int slice = x3 ? GX_K : GB_K;
        const size_t lds = x3 ? GEMM_X3_LDS_BYTES : GEMM_LDS_BYTES;
        auto kern = x3 ? k_gemm_mfma_bf16x3 : k_gemm_mfma_f32;
        const int tn = (N + GB_N - 1) / GB_N, tm = (M + GB_M - 1) / GB_M, nk = K / slice;
        // Fewer tiles than ~1.5 per CU: split K so that the chip is full (weights are then
        // streamed by >= 384 blocks instead of a few dozen); partials are reduced in a fixed order.
        int ksplit = 1;
        const int min_slices = x3 ? 2 : 4;            // keep >= 128 k per split
        if (e->use_splitk && tm * tn < 384 && nk >= 2 * min_slices) {
            // 2 workgroups per CU are resident (73.7 KB of LDS each): stay within ONE round of 512.  Rounding up (round 1:
            // 130 tiles x 4 splits = 520 workgroups) cost a second round for the last 8 - 85 us instead of ~57 for wo / w2.
            ksplit = std::min(std::min(std::max(1, 512 / (tm * tn)), nk / min_slices), 16);
            if (ksplit < 2) ksplit = 1;
        }
        if (ksplit > 1) {
            if (ensure(e, e->ssplitk, (size_t)ksplit * M * N * 4)) return -1;     // out of HBM: an error, not a silent un-split
            a.ksplit = ksplit; a.kper = (nk + ksplit - 1) / ksplit; a.partial = (float *)e->ssplitk.p;
            a.ksplit = (nk + a.kper - 1) / a.kper;          // drop empty trailing splits
            dim3 grid(tn, tm, a.ksplit);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, e->stream, a);
            hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, a);
        } else {
            a.ksplit = 1;
            dim3 grid(tn, tm);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, e->stream, a);
        }
    } else {
        dim3 grid((N + 63) / 64, (M + 3) / 4);
        hipLaunchKernelGGL(k_gemm_scalar, grid, dim3(256), 0, e->stream, a);
    }
    return 0;
}

// Large chunks (the 300 s / 600 s clips' one-feed encoder passes): 128 x 256 tiles (TN = 4) move 1.6 x fewer bytes per MFMA through
// LDS than 128 x 128.  Only from 600 such tiles on: with fewer, the last round's tail on the 512 resident workgroups eats the gain
// (the 30 s clip's w1;w3 is 520 tiles: 29 ms for the encoder pass with wide tiles everywhere against 20.4).  Measured, same box,
// alternating (profiles/r05_wide_tiles_ab.txt): encoder pass of the 300 s clip 144.9 -> 137.2 ms, ids = the reference's; with the
// weight fragments loaded straight into registers (BD) on top: 150 ms.
static bool gp_wide(int tiles_wide) { return tiles_wide >= 600; }

// y = sum_p Xp[p] . W^T on pre-split activations (vox_gemm_planes.h); same epilogue / split-K contract as launch_gemm.
// nf (optional): the RMSNorm that follows this launch's output (the residual stream) - when the launch is split along K, its reduce pass
// applies it too and writes the normalised rows as planes (k_splitk_reduce_norm_planes); *nf_done says whether that happened.
struct NormFuse { uint16_t *planes; size_t plane; const float *w, *ada; float eps; };
static int launch_gemm_planes(vox_hip_engine *e, const uint16_t *Xp, size_t plane, int ldxp, const uint16_t *W, float *Y, int ldy,
                              int M, int N, int K, const float *bias, const float *resid, int ldr, int act,
                              int epi = GP_EPI_STD, const GemmArgs *extra = nullptr, const NormFuse *nf = nullptr, bool *nf_done = nullptr) {
    if (nf_done) *nf_done = false;
    GemmArgs a{nullptr, 0, W, Y, ldy, M, N, K, bias, resid, ldr, act, 1, 0, nullptr};
    a.Xp = Xp; a.xp_plane = plane; a.ldxp = ldxp;
    if (epi == GP_EPI_SWIGLU) {       // N = hidden columns, W = [w1; w3]; output = bf16 planes of the gated hidden rows
        a.Yp = extra->Yp; a.yp_plane = extra->yp_plane;
        const int tm_ = (M + GB_M - 1) / GB_M;
        if (gp_wide(tm_ * ((N + 127) / 128))) {
            hipLaunchKernelGGL((k_gemm_planes<2, 4, GP_EPI_SWIGLU>), dim3((N + 127) / 128, tm_), dim3(256), (size_t)2 * gp_stage_bytes(4), e->stream, a);
            return 0;
        }
        const dim3 grid((N + 63) / 64, (M + GB_M - 1) / GB_M);
        hipLaunchKernelGGL((k_gemm_planes<2, 2, GP_EPI_SWIGLU>), grid, dim3(256), (size_t)2 * gp_stage_bytes(2), e->stream, a);
        return 0;
    }
    if (epi == GP_EPI_ROPE) {
        a.rope_tab = extra->rope_tab; a.rope_cols = extra->rope_cols; a.head_dim = extra->head_dim;
        a.kring = extra->kring; a.vring = extra->vring; a.ring_cap = extra->ring_cap; a.ring_kvd = extra->ring_kvd;
        a.ring_col0 = extra->ring_col0; a.ring_row0 = extra->ring_row0; a.ring_pos0 = extra->ring_pos0;
        const int tm_ = (M + GB_M - 1) / GB_M;
        if (gp_wide(tm_ * ((N + 255) / 256))) {
            hipLaunchKernelGGL((k_gemm_planes<2, 4, GP_EPI_ROPE>), dim3((N + 255) / 256, tm_), dim3(256), (size_t)2 * gp_stage_bytes(4), e->stream, a);
            return 0;
        }
        const dim3 grid((N + 127) / 128, (M + GB_M - 1) / GB_M);
        hipLaunchKernelGGL((k_gemm_planes<2, 2, GP_EPI_ROPE>), grid, dim3(256), (size_t)2 * gp_stage_bytes(2), e->stream, a);
        return 0;
    }
    if (M <= 0 || N <= 0) return 0;
    if (gp_wide(((M + GB_M - 1) / GB_M) * ((N + 255) / 256))) {
        a.ksplit = 1;
        hipLaunchKernelGGL((k_gemm_planes<2, 4>), dim3((N + 255) / 256, (M + GB_M - 1) / GB_M), dim3(256), (size_t)2 * gp_stage_bytes(4), e->stream, a);
        return 0;
    }
    constexpr int TN = 2, st = 2;          // MFMA tiles per wave along N: 128 x 128 workgroup tile
    const int BN = 64 * TN;
    const int tn = (N + BN - 1) / BN, tm = (M + GB_M - 1) / GB_M, nk = K / GP_K;
    const int resident = 512;                      // 2 workgroups per CU (64 / 80 KB of LDS each)
    int ksplit = 1;
    if (e->use_splitk && tm * tn < (resident * 3) / 4 && nk >= 8) {
        ksplit = std::min(std::min(std::max(1, resident / (tm * tn)), nk / 4), 16);
        if (ksplit < 2) ksplit = 1;
    }
    auto kern = k_gemm_planes<2, 2>;
    const size_t lds = (size_t)st * gp_stage_bytes(TN);
    if (ksplit > 1) {
        if (ensure(e, e->ssplitk, (size_t)ksplit * M * N * 4)) return -1;
        a.ksplit = ksplit; a.kper = (nk + ksplit - 1) / ksplit; a.partial = (float *)e->ssplitk.p;
        a.ksplit = (nk + a.kper - 1) / a.kper;
        {           // 1-D grid in XCD-aware order (vox_gemm_planes.h): groups of tn workgroups sharing an A slab stay on one XCD
            a.xcd_tn = tn; a.xcd_tm = tm;
            const int groups = tm * a.ksplit;
            hipLaunchKernelGGL(kern, dim3(8 * ((groups + 7) / 8) * tn), dim3(256), lds, e->stream, a);
            a.xcd_tn = 0;
        }
        if (nf && nf_done && act == ACT_NONE && N % 4 == 0 && ldy % 4 == 0 && (!resid || ldr % 4 == 0)) {
            hipLaunchKernelGGL(k_splitk_reduce_norm_planes, dim3(M), dim3(256), 0, e->stream, a, nf->planes, nf->plane, nf->w, nf->ada, nf->eps);
            *nf_done = true;
        } else {
            hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, a);
        }
    } else {
        hipLaunchKernelGGL(kern, dim3(tn, tm), dim3(256), lds, e->stream, a);
    }
    return 0;
}

static int gemv_grid(int N, int rpb) {
    const int iters = (N + rpb - 1) / rpb;
    const int maxg = 768;
    const int per = (iters + maxg - 1) / maxg;
    return (iters + per - 1) / per;
}

template <int PRO, int EPI, int RPW>
static void launch_gemv(vox_hip_engine *e, const GemvArgs &a, int grid_override = 0) {
    const int rpb = 4 * RPW;
    const int grid = grid_override ? grid_override : gemv_grid(a.N, rpb);
    const size_t lds = ((size_t)a.K + 16) * sizeof(float);
    hipLaunchKernelGGL((k_gemv<PRO, EPI, RPW>), dim3(grid), dim3(256), lds, e->stream, a);
}

// Generic y = x.W^T (+bias) for M rows on device buffers; M == 1 streams the weights with
// the GEMV kernel, M > 1 uses the MFMA GEMM.
static int linear_dev(vox_hip_engine *e, float *y, int ldy, const float *x, int ldx, const uint16_t *W,
                      const float *bias, int M, int K, int N, int act, const float *resid, int ldr,
                      int impl) {
    const bool gemv_ok = (K % 8 == 0) && act == ACT_NONE;
    if ((impl == 1 || (impl == 0 && M == 1)) && gemv_ok) {
        for (int m = 0; m < M; m++) {
            GemvArgs a{};
            a.W = W; a.x = x + (size_t)m * ldx; a.y = y + (size_t)m * ldy; a.bias = bias; a.N = N; a.K = K;
            if (resid) {
                if (resid != y) HC(hipMemcpyAsync(a.y, resid + (size_t)m * ldr, (size_t)N * 4, hipMemcpyDeviceToDevice, e->stream));
                launch_gemv<PRO_NONE, EPI_RESID, 2>(e, a);
            } else {
                launch_gemv<PRO_NONE, EPI_STORE, 2>(e, a);
            }
        }
        return 0;
    }
    return launch_gemm(e, x, ldx, W, y, ldy, M, N, K, bias, resid, ldr, act, impl == 3);
}

// ------------------------------------------------------------------------------------
// engine lifetime
// ------------------------------------------------------------------------------------
extern "C" int vox_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" const char *vox_hip_last_error(void) { return g_err.c_str(); }
extern "C" size_t vox_hip_memory_used(const vox_hip_engine_t *e) { return e ? e->mem_used : 0; }

static int self_test(vox_hip_engine *e);

static void host_inv_freq(std::vector<float> &f, int head_dim, float theta) {
    // freq = 1 / powf(theta, 2d/dim) (voxtral_kernels.c:494), computed on the host with glibc so
    // that the fp32 angle p*freq is bit-identical to the reference's.  The reference is built
    // with -ffast-math, which rewrites 1/powf(t,e) as powf(t,-e): that form reproduces its RoPE
    // tables bit-for-bit (checked against oracle/_ref), the literal form is 1 ulp off in places
    // and an ulp of freq is up to 1e-3 in cos/sin at positions ~1e4.
    f.resize(head_dim / 2);
    for (int dd = 0; dd < head_dim / 2; dd++) f[dd] = powf(theta, -((float)(2 * dd) / (float)head_dim));
}

extern "C" vox_hip_engine_t *vox_hip_engine_create(int device, const vox_hip_dims_t *dims) {
    if (!dims) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_err = "vox_hip: no HIP device available (this library has no CPU fallback)";
        fprintf(stderr, "%s\n", g_err.c_str());
        return nullptr;
    }
    if (device < 0 || device >= ndev) { g_err = "vox_hip: bad device index"; return nullptr; }
    const vox_hip_dims_t &d = *dims;
    if (d.enc_head_dim != 64 || d.dec_head_dim != 128 || d.dec_heads != 4 * d.dec_kv_heads ||
        d.enc_dim % 32 || d.dec_dim % 32 || d.enc_hidden % 32 || d.dec_hidden % 32 || d.mel_bins > 256) {
        g_err = "vox_hip: unsupported model geometry (need enc head_dim 64, dec head_dim 128, 4 q heads per kv head)";
        fprintf(stderr, "%s\n", g_err.c_str());
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { g_err = "vox_hip: hipSetDevice failed"; return nullptr; }
    {   // gfx950 only: the code object holds nothing else, and the hand-off protocols of the fused kernels (write-through sc1 stores acknowledged by
        // memory, L1-bypassing sc1 loads, no fences: vox_decfuse.h, vox_encstack.h, vox_attn.h) are statements about this part's memory system
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_err = std::string("vox_hip: device is not gfx950 (") + prop.gcnArchName + "): this engine is written for MI355X only";
            fprintf(stderr, "%s\n", g_err.c_str());
            return nullptr;
        }
    }
    vox_hip_engine *e = new vox_hip_engine();
    e->device = device;
    e->d = d;
    e->enc_qd = d.enc_heads * d.enc_head_dim;
    e->dec_qd = d.dec_heads * d.dec_head_dim;
    e->dec_kvd = d.dec_kv_heads * d.dec_head_dim;
    auto fail = [&]() -> vox_hip_engine_t * { vox_hip_engine_destroy(e); return nullptr; };
    if (const char *cm = getenv("VOX_HIP_CUMASK")) {
        // experiment: run everything on a CU-masked stream ("lo"/"hi" = bits 0-127 / 128-255,
        // "even"/"odd" = alternating bits, "q0" = bits 0-63) to see what half the CUs can stream
        uint32_t mask[8];
        for (int i = 0; i < 8; i++) {
            if (!strcmp(cm, "lo")) mask[i] = i < 4 ? 0xffffffffu : 0u;
            else if (!strcmp(cm, "hi")) mask[i] = i < 4 ? 0u : 0xffffffffu;
            else if (!strcmp(cm, "even")) mask[i] = 0x55555555u;
            else if (!strcmp(cm, "odd")) mask[i] = 0xaaaaaaaau;
            else if (!strcmp(cm, "q0")) mask[i] = i < 2 ? 0xffffffffu : 0u;
            else mask[i] = 0xffffffffu;
        }
        if (hipExtStreamCreateWithCUMask(&e->stream, 8, mask) != hipSuccess) return fail();
        fprintf(stderr, "vox_hip: CU-masked stream (%s)\n", cm);
    } else if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return fail();
    if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) return fail();

    const size_t ED = d.enc_dim, EQ = e->enc_qd, EH = d.enc_hidden;
    const size_t DD = d.dec_dim, DQ = e->dec_qd, DKV = e->dec_kvd, DH = d.dec_hidden;
    int rc = 0;
    rc |= dalloc(e, &e->tok_emb, (size_t)d.vocab * DD);
    rc |= dalloc(e, &e->conv0_w, ED * d.mel_bins * 3);
    rc |= dalloc(e, &e->conv1_w, ED * ED * 3);
    rc |= dalloc(e, &e->adapter0, DD * ED * 4);
    rc |= dalloc(e, &e->adapter1, DD * DD);
    rc |= dalloc(e, &e->conv0_b, ED); rc |= dalloc(e, &e->conv1_b, ED);
    rc |= dalloc(e, &e->enc_final_norm, ED); rc |= dalloc(e, &e->dec_final_norm, DD);
    if (rc) return fail();

    // encoder ring: window rounded up to a multiple of 64 plus slack
    e->enc_ring_cap = ((d.enc_window + 63) / 64) * 64 + 64;
    e->enc.resize(d.enc_layers);
    for (auto &L : e->enc) {
        rc |= dalloc(e, &L.wqkv, 3 * EQ * ED); rc |= dalloc(e, &L.wo, ED * EQ);
        rc |= dalloc(e, &L.w13, 2 * EH * ED);  rc |= dalloc(e, &L.w2, ED * EH);
        rc |= dalloc(e, &L.bqkv, 3 * EQ); rc |= dalloc(e, &L.bo, ED); rc |= dalloc(e, &L.b2, ED);
        rc |= dalloc(e, &L.n1, ED); rc |= dalloc(e, &L.n2, ED);
        rc |= dalloc(e, &L.kring, (size_t)e->enc_ring_cap * EQ);
        rc |= dalloc(e, &L.vring, (size_t)e->enc_ring_cap * EQ);
        if (rc) return fail();
        HCV(hipMemsetAsync(L.bqkv, 0, 3 * EQ * 4, e->stream));   // wk has no bias (voxtral.h:63)
    }
    e->dec_ring_cap = d.dec_window + DEC_RING_EXTRA;
    e->dec.resize(d.dec_layers);
    for (auto &L : e->dec) {
        rc |= dalloc(e, &L.wqkv, (DQ + 2 * DKV) * DD); rc |= dalloc(e, &L.wo, DD * DQ);
        rc |= dalloc(e, &L.w13, 2 * DH * DD); rc |= dalloc(e, &L.w2, DD * DH);
        rc |= dalloc(e, &L.n1, DD); rc |= dalloc(e, &L.n2, DD); rc |= dalloc(e, &L.ada, DD);
        rc |= dalloc(e, &L.kring, (size_t)e->dec_ring_cap * DKV);
        rc |= dalloc(e, &L.vring, (size_t)e->dec_ring_cap * DKV);
        if (rc) return fail();
        HCV(hipMemsetAsync(L.ada, 0, DD * 4, e->stream));
    }
    // mel tables + rope
    rc |= dalloc(e, &e->hann, MEL_NFFT);
    rc |= dalloc(e, &e->cosT, (size_t)MEL_NFFT * MEL_NFREQ);
    rc |= dalloc(e, &e->sinT, (size_t)MEL_NFFT * MEL_NFREQ);
    rc |= dalloc(e, &e->filtT, (size_t)MEL_NFREQ * d.mel_bins);
    rc |= dalloc(e, &e->enc_inv_freq, 256);          // padded to 1 KiB: k_dec_attn_fused fetches the table with one LDS-DMA
    rc |= dalloc(e, &e->dec_inv_freq, 256);
    rc |= dalloc(e, &e->dec_rope, d.dec_head_dim);
    if (rc) return fail();
    HCV(hipMemset(e->enc_inv_freq, 0, 1024)); HCV(hipMemset(e->dec_inv_freq, 0, 1024));
    {
        std::vector<float> f;
        host_inv_freq(f, d.enc_head_dim, d.rope_theta);
        HCV(hipMemcpy(e->enc_inv_freq, f.data(), f.size() * 4, hipMemcpyHostToDevice));
        host_inv_freq(f, d.dec_head_dim, d.rope_theta);
        HCV(hipMemcpy(e->dec_inv_freq, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    }
    // decoder step buffers
    rc |= dalloc(e, &e->d_st, 1);
    rc |= dalloc(e, &e->d_attn_arrive, (size_t)std::max(1, d.enc_heads));
    if (!rc && hipMemset(e->d_attn_arrive, 0, (size_t)std::max(1, d.enc_heads) * 4) != hipSuccess) rc = -1;
    rc |= dalloc(e, &e->dx, DD); rc |= dalloc(e, &e->dx2, DD); rc |= dalloc(e, &e->dq, DQ); rc |= dalloc(e, &e->dattn, DQ);
    rc |= dalloc(e, &e->dh, DH); rc |= dalloc(e, &e->dlogits, (size_t)d.vocab);
    e->logits_grid = (d.vocab % (16 * 1024) == 0) ? 1024 : gemv_grid(d.vocab, 16);   // 131072 rows: 8 even trips
    rc |= dalloc(e, &e->blk_val, e->logits_grid); rc |= dalloc(e, &e->blk_idx, e->logits_grid);
    rc |= dalloc(e, &e->d_tokens, MAX_RUN_STEPS);
    e->dec_max_split = (d.dec_window + 63) / 64 + 1;
    rc |= dalloc(e, &e->dpart_o, (size_t)d.dec_heads * e->dec_max_split * d.dec_head_dim);
    rc |= dalloc(e, &e->dpart_ml, (size_t)d.dec_heads * e->dec_max_split * 2);
    if (rc) return fail();
    HCV(hipMemsetAsync(e->d_st, 0, sizeof(DecState), e->stream));

    // adapter buffer: 4096 rows to start with (grows / compacts on demand)
    e->adapter_cap = 4096;
    if (dalloc(e, &e->adapter, (size_t)e->adapter_cap * DD)) return fail();

    // switches that do not depend on the fused decode kernels being available (round 6: they used to be latched inside the block below,
    // i.e. silently ignored on small presets, under VOX_HIP_CUMASK or with the fused kernels off - where a fallback is most likely needed)
    e->fp8_attn_bf16 = vox_disabled("fp8_attn");
    e->fp8_lmhead_bf16 = vox_disabled("fp8_lmhead");
    e->fp8_prefill_bf16 = vox_disabled("fp8_prefill");
    e->enc_tl_on = getenv("VOX_HIP_ENC_TL") != nullptr;
    if (dalloc(e, &e->d_f8_clamped, 4) || hipMemset(e->d_f8_clamped, 0, 16) != hipSuccess ||
        hipHostMalloc((void **)&e->h_f8_clamped, 16, hipHostMallocDefault) != hipSuccess) return fail();
    e->h_f8_clamped[0] = 0;
    if (hipHostMalloc((void **)&e->pin, sizeof(vox_hip_engine::HostPin), hipHostMallocDefault) != hipSuccess) return fail();
    memset(e->pin, 0, sizeof(vox_hip_engine::HostPin));
    for (int i = 0; i < vox_hip_engine::SMP_SLOTS; i++)
        if (hipHostMalloc(&e->smp_pin[i], vox_hip_engine::SMP_SLOT_BYTES, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&e->smp_ev[i], hipEventDisableTiming) != hipSuccess) return fail();

    // fused attention half of the decode step: exact 4B decoder shapes on a 256-CU part (one workgroup per CU)
    {
        hipDeviceProp_t prop;
        const bool geom = d.dec_dim == DF_D && e->dec_qd == DF_DQ && e->dec_kvd == DF_DKV && d.dec_hidden == 9216 &&
                          d.dec_head_dim == DF_HD && d.dec_heads == 32 && d.dec_kv_heads == DF_GROUPS;
        if (geom && !vox_disabled("fused") && !vox_disabled("fast") && !getenv("VOX_HIP_CUMASK") &&
            hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount == DF_BLOCKS) {
            bool ok = dalloc(e, &e->d_gq, (size_t)DF_GROUPS * DF_GQ) == 0 && dalloc(e, &e->d_gp, (size_t)DF_GROUPS * DF_BPG * DF_GP) == 0 &&
                      dalloc(e, &e->d_wo_part, (size_t)DF_GROUPS * DF_D) == 0 && dalloc(e, &e->d_fuse_err, 64) == 0 &&
                      dalloc(e, &e->d_gh, (size_t)FFN_H) == 0 && dalloc(e, &e->d_xprime, (size_t)DF_D) == 0 &&
                      hipMemset(e->d_gh, 0, (size_t)FFN_H * 8) == hipSuccess;
            ok = ok && hipMemset(e->d_gq, 0, (size_t)DF_GROUPS * DF_GQ * 8) == hipSuccess &&
                 hipMemset(e->d_gp, 0, (size_t)DF_GROUPS * DF_BPG * DF_GP * 8) == hipSuccess &&
                 hipMemset(e->d_fuse_err, 0, 64 * 4) == hipSuccess;
            ok = ok && hipFuncSetAttribute((const void *)k_dec_attn_fused<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_attn_fused<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_attn_fused<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_attn_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_attn_fused<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_attn_fused<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_gemv_w13x<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W13X_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_gemv_w13x<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W13X_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_gemv_w2x<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W2X_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_gemv_w2x<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W2X_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_ffn_fused, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_attn12<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_attn12<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_w2x_attn12, hipFuncAttributeMaxDynamicSharedMemorySize, DA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_ffn_attn12<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_ffn_attn12<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_stack<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FA12_LDS_BYTES) == hipSuccess &&
                 hipFuncSetAttribute((const void *)k_dec_stack<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FA12_LDS_BYTES) == hipSuccess &&
                 dalloc(e, &e->d_gx, (size_t)DF_D) == 0 && hipMemset(e->d_gx, 0, (size_t)DF_D * 8) == hipSuccess &&
                 dalloc(e, &e->d_gw, (size_t)8 * DF_D) == 0 && hipMemset(e->d_gw, 0, (size_t)8 * DF_D * 8) == hipSuccess &&
                 dalloc(e, &e->d_gxp, (size_t)DF_D) == 0 && hipMemset(e->d_gxp, 0, (size_t)DF_D * 8) == hipSuccess &&
                 dalloc(e, &e->d_stack_tab, (size_t)e->d.dec_layers) == 0;
            if (!ok) { (void)hipGetLastError(); fprintf(stderr, "vox_hip: fused decode kernels unavailable; launch-per-GEMV chain\n"); }
            e->use_fused = ok; e->fused_ok = ok;
            e->use_ffn = ok && !vox_disabled("ffn_fused");             // k_gemv_w13x + k_gemv_w2x instead of k_ffn_fused
            if (vox_disabled("merge12")) e->merge12 = 0;
            if (vox_disabled("merge12_long")) e->merge12_long = 0;
            if (vox_disabled("stack")) e->use_stack = 0;
            if (ok && getenv("VOX_HIP_FUSE_TL") && hipMalloc((void **)&e->d_fuse_tl, 3 * 1024 * TL_STRIDE * 8) == hipSuccess)
                hipMemset(e->d_fuse_tl, 0, 3 * 1024 * TL_STRIDE * 8);
        }
    }

    // state-carrying stream buffers (zero = "start of sequence" left padding)
    if (ensure_keep(e, e->conv_in0, (size_t)(2 + 1024) * d.mel_bins * 4, 0)) return fail();
    if (ensure_keep(e, e->conv_in1, (size_t)(2 + 1024) * ED * 4, 0)) return fail();
    if (ensure_keep(e, e->enc_out, (size_t)(3 + 512) * ED * 4, 0)) return fail();
    if (esync(e) != hipSuccess) return fail();
    if (self_test(e) != 0) return fail();
    return e;
}

extern "C" void vox_hip_engine_destroy(vox_hip_engine_t *e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->up) { e->up->stop(); delete e->up; e->up = nullptr; }
    (void)drain_fences(e);
    if (e->stream) esync(e);
    auto F = [](void *p) { if (p) hipFree(p); };
    F(e->splanes.p);
    F(e->tok_emb); F(e->conv0_w); F(e->conv1_w); F(e->adapter0); F(e->adapter1);
    F(e->conv0_b); F(e->conv1_b); F(e->enc_final_norm); F(e->dec_final_norm);
    for (auto &L : e->enc) { F(L.wqkv); F(L.wo); F(L.w13); F(L.w2); F(L.bqkv); F(L.bo); F(L.b2); F(L.n1); F(L.n2); F(L.kring); F(L.vring); }
    for (auto &L : e->dec) {
        F(L.wqkv); F(L.wo); F(L.w13); F(L.w2); F(L.n1); F(L.n2); F(L.ada); F(L.kring); F(L.vring);
        F(L.wqkv8); F(L.wo8); F(L.w138); F(L.w28); F(L.sqkv); F(L.so); F(L.s13); F(L.s2);
    }
    F(e->tok_emb8); F(e->stok); F(e->d_taps); F(e->tok_emb_s);
    for (auto &L : e->dec) { F(L.wqkv_s); F(L.wo_s); F(L.w13_s); F(L.w2_s); }
    F(e->hann); F(e->cosT); F(e->sinT); F(e->filtT); F(e->enc_inv_freq); F(e->dec_inv_freq); F(e->dec_rope);
    F(e->d_st); F(e->dx); F(e->dx2); F(e->dq); F(e->dattn); F(e->dh); F(e->dlogits); F(e->blk_val); F(e->blk_idx);
    F(e->d_tokens); F(e->dpart_o); F(e->dpart_ml); F(e->adapter); F(e->d_gq); F(e->d_gp); F(e->d_wo_part); F(e->d_fuse_err);
    F(e->d_gh); F(e->d_gx); F(e->d_xprime); F(e->d_gw); F(e->d_gxp); F(e->d_stack_tab); F(e->d_attn_arrive);
    F(e->d_es_tab); F(e->d_es_xa); F(e->d_es_xb); F(e->d_es_ssq); F(e->d_es_q); F(e->d_es_po); F(e->d_es_pml); F(e->d_es_wop); F(e->d_es_w2p);
    F(e->d_es_apl); F(e->d_es_hpl); F(e->d_es_flags); F(e->d_es_err); F(e->d_es_tl); F(e->es_carry.p);
    if (e->h_es_err) hipHostFree(e->h_es_err);
    if (e->h_f8_clamped) hipHostFree(e->h_f8_clamped);
    if (e->pin) hipHostFree(e->pin);
    for (int i = 0; i < vox_hip_engine::SMP_SLOTS; i++) { if (e->smp_pin[i]) hipHostFree(e->smp_pin[i]); if (e->smp_ev[i]) hipEventDestroy(e->smp_ev[i]); }
    F(e->d_f8_clamped);
    Buf *bufs[] = {&e->conv_in0, &e->conv_in1, &e->enc_out, &e->sx, &e->sxn, &e->sqkv, &e->sattn, &e->sgu, &e->sh,
                   &e->srope, &e->sim2col, &e->ssamples, &e->smid, &e->stmp_in, &e->stmp_out, &e->spart_o, &e->spart_ml, &e->ssplitk};
    for (Buf *b : bufs) F(b->p);
    for (auto ev : e->xev) if (ev) hipEventDestroy(ev);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

// ------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------
extern "C" int vox_hip_upload_bf16(vox_hip_engine_t *e, int tensor, int layer, const uint16_t *src, size_t n) {
    if (!e || !src) return -1;
    HC(hipSetDevice(e->device));
    const vox_hip_dims_t &d = e->d;
    const size_t ED = d.enc_dim, EQ = e->enc_qd, EH = d.enc_hidden;
    const size_t DD = d.dec_dim, DQ = e->dec_qd, DKV = e->dec_kvd, DH = d.dec_hidden;
    uint16_t *dst = nullptr; size_t expect = 0;
    const bool enc_l = layer >= 0 && layer < d.enc_layers, dec_l = layer >= 0 && layer < d.dec_layers;
    switch (tensor) {
        case VOXT_TOK_EMB: dst = e->tok_emb; expect = (size_t)d.vocab * DD; break;
        case VOXT_CONV0_W: dst = e->conv0_w; expect = ED * d.mel_bins * 3; break;
        case VOXT_CONV1_W: dst = e->conv1_w; expect = ED * ED * 3; break;
        case VOXT_ADAPTER0: dst = e->adapter0; expect = DD * ED * 4; break;
        case VOXT_ADAPTER1: dst = e->adapter1; expect = DD * DD; break;
        case VOXT_ENC_WQ: if (enc_l) { dst = e->enc[layer].wqkv; expect = EQ * ED; } break;
        case VOXT_ENC_WK: if (enc_l) { dst = e->enc[layer].wqkv + EQ * ED; expect = EQ * ED; } break;
        case VOXT_ENC_WV: if (enc_l) { dst = e->enc[layer].wqkv + 2 * EQ * ED; expect = EQ * ED; } break;
        case VOXT_ENC_WO: if (enc_l) { dst = e->enc[layer].wo; expect = ED * EQ; } break;
        case VOXT_ENC_W1: if (enc_l) { dst = e->enc[layer].w13; expect = EH * ED; } break;
        case VOXT_ENC_W3: if (enc_l) { dst = e->enc[layer].w13 + EH * ED; expect = EH * ED; } break;
        case VOXT_ENC_W2: if (enc_l) { dst = e->enc[layer].w2; expect = ED * EH; } break;
        case VOXT_DEC_WQ: if (dec_l) { dst = e->dec[layer].wqkv; expect = DQ * DD; } break;
        case VOXT_DEC_WK: if (dec_l) { dst = e->dec[layer].wqkv + DQ * DD; expect = DKV * DD; } break;
        case VOXT_DEC_WV: if (dec_l) { dst = e->dec[layer].wqkv + (DQ + DKV) * DD; expect = DKV * DD; } break;
        case VOXT_DEC_WO: if (dec_l) { dst = e->dec[layer].wo; expect = DD * DQ; } break;
        case VOXT_DEC_W1: if (dec_l) { dst = e->dec[layer].w13; expect = DH * DD; } break;
        case VOXT_DEC_W3: if (dec_l) { dst = e->dec[layer].w13 + DH * DD; expect = DH * DD; } break;
        case VOXT_DEC_W2: if (dec_l) { dst = e->dec[layer].w2; expect = DD * DH; } break;
        default: break;
    }
    if (!dst || expect != n) {
        char b[160]; snprintf(b, sizeof b, "vox_hip_upload_bf16: tensor %d layer %d: got %zu elems, expected %zu", tensor, layer, n, expect);
        g_err = b; fprintf(stderr, "%s\n", b);
        return -1;
    }
    if (e->use_staged_upload && n * 2 >= ((size_t)4 << 20)) {
        if (!e->up) e->up = new Uploader();
        if (e->up->start() && e->up->copy(dst, src, n * 2) == 0 && e->up->fence(e->stream) == 0) return 0;
        // staging unavailable (stream / event / pinned allocation refused, or a copy failed): release whatever start() got
        // hold of - a retry per tensor would leak it again and again - and use plain copies for the rest of this engine's life
        (void)hipGetLastError();
        e->up->stop(); delete e->up; e->up = nullptr;
        e->use_staged_upload = false;
        fprintf(stderr, "vox_hip: staged weight upload unavailable; plain hipMemcpy from here on\n");
        HC(esync(e));
    }
    HC(hipMemcpy(dst, src, n * 2, hipMemcpyHostToDevice));
    return 0;
}

// All staged uploads have reached HBM; the staging threads and pinned slots are released (vox_load calls this last).
extern "C" int vox_hip_upload_done(vox_hip_engine_t *e) {
    if (!e) return -1;
    if (e->up) {
        HC(hipSetDevice(e->device));
        const int rc = e->up->flush();
        e->up->stop(); delete e->up; e->up = nullptr;
        if (rc) { g_err = "vox_hip_upload_done: staged copies failed"; return -1; }
    }
    return 0;
}

extern "C" int vox_hip_upload_f32(vox_hip_engine_t *e, int tensor, int layer, const float *src, size_t n) {
    if (!e || !src) return -1;
    HC(hipSetDevice(e->device));
    const vox_hip_dims_t &d = e->d;
    const size_t ED = d.enc_dim, EQ = e->enc_qd, DD = d.dec_dim;
    float *dst = nullptr; size_t expect = 0;
    const bool enc_l = layer >= 0 && layer < d.enc_layers, dec_l = layer >= 0 && layer < d.dec_layers;
    switch (tensor) {
        case VOXT_CONV0_B: dst = e->conv0_b; expect = ED; break;
        case VOXT_CONV1_B: dst = e->conv1_b; expect = ED; break;
        case VOXT_ENC_FINAL_NORM: dst = e->enc_final_norm; expect = ED; break;
        case VOXT_DEC_FINAL_NORM: dst = e->dec_final_norm; expect = DD; break;
        case VOXT_ENC_BQ: if (enc_l) { dst = e->enc[layer].bqkv; expect = EQ; } break;
        case VOXT_ENC_BV: if (enc_l) { dst = e->enc[layer].bqkv + 2 * EQ; expect = EQ; } break;
        case VOXT_ENC_BO: if (enc_l) { dst = e->enc[layer].bo; expect = ED; } break;
        case VOXT_ENC_B2: if (enc_l) { dst = e->enc[layer].b2; expect = ED; } break;
        case VOXT_ENC_ATTN_NORM: if (enc_l) { dst = e->enc[layer].n1; expect = ED; } break;
        case VOXT_ENC_FFN_NORM: if (enc_l) { dst = e->enc[layer].n2; expect = ED; } break;
        case VOXT_DEC_ATTN_NORM: if (dec_l) { dst = e->dec[layer].n1; expect = DD; } break;
        case VOXT_DEC_FFN_NORM: if (dec_l) { dst = e->dec[layer].n2; expect = DD; } break;
        case VOXT_DEC_ADA_SCALE: if (dec_l) { dst = e->dec[layer].ada; expect = DD; } break;
        default: break;
    }
    if (!dst || expect != n) {
        char b[160]; snprintf(b, sizeof b, "vox_hip_upload_f32: tensor %d layer %d: got %zu elems, expected %zu", tensor, layer, n, expect);
        g_err = b; fprintf(stderr, "%s\n", b);
        return -1;
    }
    HC(esync(e));
    HC(hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int vox_hip_upload_mel_tables(vox_hip_engine_t *e, const float *filters, const float *hann,
                                         const float *dft_cos, const float *dft_sin) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    const int MB = e->d.mel_bins;
    std::vector<float> cT((size_t)MEL_NFFT * MEL_NFREQ), sT((size_t)MEL_NFFT * MEL_NFREQ), fT((size_t)MEL_NFREQ * MB);
    for (int k = 0; k < MEL_NFREQ; k++)
        for (int n = 0; n < MEL_NFFT; n++) {
            cT[(size_t)n * MEL_NFREQ + k] = dft_cos[(size_t)k * MEL_NFFT + n];
            sT[(size_t)n * MEL_NFREQ + k] = dft_sin[(size_t)k * MEL_NFFT + n];
        }
    for (int m = 0; m < MB; m++)
        for (int k = 0; k < MEL_NFREQ; k++) fT[(size_t)k * MB + m] = filters[(size_t)m * MEL_NFREQ + k];
    HC(hipMemcpy(e->hann, hann, MEL_NFFT * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(e->cosT, cT.data(), cT.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(e->sinT, sT.data(), sT.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(e->filtT, fT.data(), fT.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------
// large-M transformer layer (encoder chunk rows / decoder prefill rows)
// ------------------------------------------------------------------------------------
struct RowsCfg {
    int D, QD, KVD, H, heads, kv_heads, hd, window;
    float eps;
    bool is_enc;
};

static int ensure_rows_scratch(vox_hip_engine *e, int n, const RowsCfg &c) {
    const size_t N3 = (size_t)c.QD + 2 * c.KVD;
    if (ensure(e, e->sxn, (size_t)n * c.D * 4)) return -1;
    if (ensure(e, e->sqkv, (size_t)n * N3 * 4)) return -1;
    if (ensure(e, e->sattn, (size_t)n * c.QD * 4)) return -1;
    if (ensure(e, e->sgu, (size_t)n * 2 * c.H * 4)) return -1;
    if (ensure(e, e->sh, (size_t)n * c.H * 4)) return -1;
    if (ensure(e, e->srope, (size_t)n * c.hd * 4)) return -1;
    return 0;
}

// x: [n, D] device, updated in place.  pos0 = logical position of row 0.
static int run_layer_rows(vox_hip_engine *e, float *x, int n, int pos0, const RowsCfg &c,
                          const uint16_t *wqkv, const float *bqkv, const uint16_t *wo, const float *bo,
                          const uint16_t *w13, const uint16_t *w2, const float *b2,
                          const float *n1, const float *n2, const float *ada,
                          float *kring, float *vring, int ring_cap,
                          const float *next_n1 = nullptr, uint16_t **xplanes = nullptr) {
    // next_n1 / xplanes (round 6, the encoder's layer loop): *xplanes != null on entry = the planes of this layer's normalised input, left
    // there by the previous layer's W2 reduce pass; on return *xplanes = the planes of (x, next_n1) if this layer's W2 launch could apply the
    // next layer's attention_norm in its reduce pass, null otherwise.
    float *xn = (float *)e->sxn.p, *qkv = (float *)e->sqkv.p, *attn = (float *)e->sattn.p;
    float *gu = (float *)e->sgu.p, *h = (float *)e->sh.p, *tab = (float *)e->srope.p;
    const int N3 = c.QD + 2 * c.KVD;
    hipStream_t s = e->stream;
    // Large chunks: the activations of every GEMM are written by their producer as bf16 planes (exact 3-term split, once)
    // and the GEMMs run on k_gemm_planes; small ones keep the f32-activation kernels.
    const bool planes = e->use_planes && e->use_mfma && e->use_bf16x3 && n >= 64 && c.D % GP_K == 0 && c.QD % GP_K == 0 && c.H % GP_K == 0;
    uint16_t *P = nullptr;
    if (planes) {
        const size_t pmax = (size_t)3 * n * std::max(std::max(c.D, c.QD), c.H);
        if (ensure(e, e->splanes, 2 * pmax * 2)) return -1;          // two plane sets: the W1;W3 launch reads one and writes the other
        P = (uint16_t *)e->splanes.p;
    }
    // (only for chunks that fill the chip with tiles: the epilogue variants have no split-K, and a 68-row pass ran its QKV launch
    // as 96 workgroups x 40 sequential K slices: 56 us against ~26 for split-K + reduce + RoPE)
    bool ring_in_epi = false;
    const bool fuse_epi = planes && e->use_epi && n >= 512 && (c.QD + c.KVD) % 2 == 0 && c.H % 64 == 0;
    // 1. attention_norm   2. merged QKV projection (+ q/v bias on the encoder, voxtral_encoder.c:542-544)
    uint16_t *const xin = xplanes ? *xplanes : nullptr;
    if (xplanes) *xplanes = nullptr;
    uint16_t *const Pset0 = P, *const Pset1 = P ? P + (size_t)3 * n * std::max(std::max(c.D, c.QD), c.H) : nullptr;
    if (planes) {
        if (xin && (xin == Pset0 || xin == Pset1)) P = xin;      // (normalised by the previous layer's W2 reduce)
        else hipLaunchKernelGGL(k_rmsnorm_planes, dim3(n), dim3(256), 0, s, P, (size_t)n * c.D, (const float *)x, c.D, n1, (const float *)nullptr, c.D, c.eps);
        if (fuse_epi) {
            GemmArgs x{}; x.rope_tab = tab; x.rope_cols = c.QD + c.KVD; x.head_dim = c.hd;
            // (round 6) the epilogue also files the rows the next chunk will need in the K / V rings - where that cannot overwrite a row THIS
            // chunk's attention still reads from the ring: no older position in the window (pos0 == 0), or room for both
            ring_in_epi = c.is_enc && (pos0 == 0 || n <= ring_cap - c.window) && !vox_disabled("enc_fuse");
            if (ring_in_epi) {
                const int keep = std::min(n, c.window);
                x.kring = kring; x.vring = vring; x.ring_cap = ring_cap; x.ring_kvd = c.KVD; x.ring_col0 = c.QD; x.ring_row0 = n - keep; x.ring_pos0 = pos0 + n - keep;
            }
            if (launch_gemm_planes(e, P, (size_t)n * c.D, c.D, wqkv, qkv, N3, n, N3, c.D, bqkv, nullptr, 0, ACT_NONE, GP_EPI_ROPE, &x)) return -1;
        } else if (launch_gemm_planes(e, P, (size_t)n * c.D, c.D, wqkv, qkv, N3, n, N3, c.D, bqkv, nullptr, 0, ACT_NONE)) return -1;
    } else {
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, s, xn, c.D, x, c.D, n1, (const float *)nullptr, c.D, c.eps);
        if (launch_gemm(e, xn, c.D, wqkv, qkv, N3, n, N3, c.D, bqkv, nullptr, 0, ACT_NONE)) return -1;
    }
    // 3. RoPE on q and k columns (table built once per chunk by the caller) - unless the QKV launch did it in its epilogue
    if (!fuse_epi) hipLaunchKernelGGL(k_rope_apply, dim3(grid1d((size_t)n * (c.QD + c.KVD) / 2)), dim3(256), 0, s,
                       qkv, N3, n, c.QD + c.KVD, c.hd, tab);
    // 4. attention over [window tail in the ring] + [this chunk]
    const float scale = 1.0f / sqrtf((float)c.hd);
    bool attn_planes = false;
    if (c.is_enc) {
        AttnArgs a{};
        a.out = attn; a.ldo = c.QD; a.q = qkv; a.ldq = N3; a.n_q = n; a.qpos0 = pos0;
        a.kB = qkv + c.QD; a.vB = qkv + c.QD + c.KVD; a.ldB = N3; a.posB0 = pos0; a.last_key = pos0 + n - 1;
        a.kA = kring; a.vA = vring; a.capA = ring_cap; a.ldA = c.KVD;
        a.n_heads = c.heads; a.n_kv_heads = c.kv_heads; a.scale = scale; a.window = c.window; a.st = nullptr;
        {
            const int qt = (n + 127) / 128, blocks = qt * c.heads;
            const int span = std::min(pos0 + n, c.window + std::min(n, 128));
            int ks = 1;
            if (blocks < 256) ks = std::max(1, std::min((512 + blocks - 1) / blocks, (span + 63) / 64));
            if (ks > 1) {
                if (ensure(e, e->spart_o, (size_t)n * c.heads * ks * c.hd * 4)) return -1;
                if (ensure(e, e->spart_ml, (size_t)n * c.heads * ks * 2 * 4)) return -1;
                a.part_o = (float *)e->spart_o.p; a.part_ml = (float *)e->spart_ml.p;
            }
            a.xcd_map = 1;
            // (round 6) one key slice: the MFMA kernel writes its output straight as the Wo launch's planes (no k_split_planes pass)
            attn_planes = planes && ks == 1 && e->use_attn_mfma && c.hd == 64 && c.heads == c.kv_heads && !vox_disabled("enc_fuse");
            if (attn_planes) { a.out_planes = P; a.out_plane = (size_t)n * c.QD; }
            if (e->use_attn_mfma && c.hd == 64 && c.heads == c.kv_heads)
                hipLaunchKernelGGL(k_attn_enc_bf16, dim3(qt, c.heads, ks), dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL((k_attn_rows<64>), dim3(qt, c.heads, ks), dim3(128), 0, s, a);
            if (ks > 1)
                hipLaunchKernelGGL((k_attn_combine<64>), dim3(c.heads, n), dim3(64), 0, s, attn, c.QD,
                                   (const float *)a.part_o, (const float *)a.part_ml, c.heads, ks);
        }
        // keep the last min(n, window) rows for the next chunk
        const int keep = std::min(n, c.window);
        if (!ring_in_epi)
            hipLaunchKernelGGL(k_ring_append, dim3(grid1d((size_t)keep * c.KVD / 4)), dim3(256), 0, s,
                               kring, vring, ring_cap, c.KVD, qkv, N3, c.QD, c.QD + c.KVD, n - keep, keep, pos0 + n - keep);
    } else {
        hipLaunchKernelGGL(k_ring_append, dim3(grid1d((size_t)n * c.KVD / 4)), dim3(256), 0, s,
                           kring, vring, ring_cap, c.KVD, qkv, N3, c.QD, c.QD + c.KVD, 0, n, pos0);
        const int max_len = std::min(pos0 + n, c.window);
        const int nsplit = (max_len + DEC_SPLIT_KEYS - 1) / DEC_SPLIT_KEYS;
        AttnArgs a{};
        a.out = attn; a.ldo = c.QD; a.q = qkv; a.ldq = N3; a.n_q = n; a.qpos0 = pos0;
        a.kB = nullptr; a.vB = nullptr; a.ldB = 0; a.posB0 = INT_MAX; a.last_key = pos0 + n - 1;
        a.kA = kring; a.vA = vring; a.capA = ring_cap; a.ldA = c.KVD;
        a.n_heads = c.heads; a.n_kv_heads = c.kv_heads; a.scale = scale; a.window = c.window; a.st = nullptr;
        a.split_keys = DEC_SPLIT_KEYS;
        if (nsplit > 1) {
            if (ensure(e, e->spart_o, (size_t)n * c.heads * nsplit * c.hd * 4)) return -1;
            if (ensure(e, e->spart_ml, (size_t)n * c.heads * nsplit * 2 * 4)) return -1;
            a.part_o = (float *)e->spart_o.p; a.part_ml = (float *)e->spart_ml.p;
        }
        if (e->use_dpp)
            hipLaunchKernelGGL((k_attn_dec<128, 4, true>), dim3(c.kv_heads, nsplit, n), dim3(256), 0, s, a, nsplit);
        else
            hipLaunchKernelGGL((k_attn_dec<128, 4, false>), dim3(c.kv_heads, nsplit, n), dim3(256), 0, s, a, nsplit);
        if (nsplit > 1)
            hipLaunchKernelGGL((k_attn_combine<128>), dim3(c.heads, n), dim3(128), 0, s, attn, c.QD,
                               (const float *)a.part_o, (const float *)a.part_ml, c.heads, nsplit);
    }
    if (planes) {
        // 5. x += attn.Wo^T (+bo)   6. ffn_norm (+ ada)   7. SwiGLU: merged W1;W3 GEMM, gate, W2 (+b2) + residual
        if (!attn_planes)
            hipLaunchKernelGGL(k_split_planes, dim3(grid1d((size_t)n * c.QD / 4)), dim3(256), 0, s, P, (size_t)n * c.QD, (const float *)attn, c.QD, n, c.QD);
        // (round 6) a split-K Wo launch reduces, adds the residual and applies the ffn_norm in one pass; its planes go to the second plane set
        // (the launch itself still reads the first one)
        uint16_t *Pn = P == Pset0 ? Pset1 : Pset0;
        const NormFuse nf{Pn, (size_t)n * c.D, n2, ada, c.eps};
        bool normed = false;
        if (launch_gemm_planes(e, P, (size_t)n * c.QD, c.QD, wo, x, c.D, n, c.D, c.QD, bo, x, c.D, ACT_NONE, GP_EPI_STD, nullptr,
                               vox_disabled("enc_fuse") ? nullptr : &nf, &normed)) return -1;
        if (normed) std::swap(P, Pn);
        else hipLaunchKernelGGL(k_rmsnorm_planes, dim3(n), dim3(256), 0, s, P, (size_t)n * c.D, (const float *)x, c.D, n2, ada, c.D, c.eps);
        if (fuse_epi) {
            uint16_t *P2 = Pn;           // (the plane set that does NOT hold the normalised rows)
            GemmArgs xa{}; xa.Yp = P2; xa.yp_plane = (size_t)n * c.H;
            if (launch_gemm_planes(e, P, (size_t)n * c.D, c.D, w13, nullptr, 0, n, c.H, c.D, nullptr, nullptr, 0, ACT_NONE, GP_EPI_SWIGLU, &xa)) return -1;
            // (round 6) the W2 launch's reduce pass also applies the NEXT layer's attention_norm; its planes go where this layer's
            // normalised rows were (the W1;W3 launch is done with them)
            const NormFuse nfn{P, (size_t)n * c.D, next_n1, nullptr, c.eps};
            bool nnormed = false;
            const bool try_next = xplanes && next_n1 && !vox_disabled("enc_fuse");
            if (launch_gemm_planes(e, P2, (size_t)n * c.H, c.H, w2, x, c.D, n, c.D, c.H, b2, x, c.D, ACT_NONE, GP_EPI_STD, nullptr,
                                   try_next ? &nfn : nullptr, &nnormed)) return -1;
            if (xplanes) *xplanes = nnormed ? P : nullptr;
            return 0;
        }
        if (launch_gemm_planes(e, P, (size_t)n * c.D, c.D, w13, gu, 2 * c.H, n, 2 * c.H, c.D, nullptr, nullptr, 0, ACT_NONE)) return -1;
        hipLaunchKernelGGL(k_silu_mul_planes, dim3(grid1d((size_t)n * c.H / 4)), dim3(256), 0, s, P, (size_t)n * c.H, (const float *)gu, n, c.H);
        if (launch_gemm_planes(e, P, (size_t)n * c.H, c.H, w2, x, c.D, n, c.D, c.H, b2, x, c.D, ACT_NONE)) return -1;
        return 0;
    }
    // 5. x += attn.Wo^T (+bo)
    if (launch_gemm(e, attn, c.QD, wo, x, c.D, n, c.D, c.QD, bo, x, c.D, ACT_NONE)) return -1;
    // 6. ffn_norm (+ ada scaling on the decoder)
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, s, xn, c.D, x, c.D, n2, ada, c.D, c.eps);
    // 7. SwiGLU: merged W1;W3 GEMM, gate, W2 (+b2) + residual
    if (launch_gemm(e, xn, c.D, w13, gu, 2 * c.H, n, 2 * c.H, c.D, nullptr, nullptr, 0, ACT_NONE)) return -1;
    hipLaunchKernelGGL(k_silu_mul, dim3(grid1d((size_t)n * c.H / 4)), dim3(256), 0, s, h, gu, n, c.H);
    if (launch_gemm(e, h, c.H, w2, x, c.D, n, c.D, c.H, b2, x, c.D, ACT_NONE)) return -1;
    return 0;
}

// k_rows_finish: one float4 column group per thread (whole waves, at most 1024 threads)
static inline int rf_threads(int D) { return std::min(1024, std::max(64, ((D / 4 + 63) / 64) * 64)); }

static RowsCfg enc_cfg(const vox_hip_engine *e) {
    const vox_hip_dims_t &d = e->d;
    return RowsCfg{d.enc_dim, e->enc_qd, e->enc_qd, d.enc_hidden, d.enc_heads, d.enc_heads, d.enc_head_dim,
                   d.enc_window, d.enc_eps, true};
}
static RowsCfg dec_cfg(const vox_hip_engine *e) {
    const vox_hip_dims_t &d = e->d;
    return RowsCfg{d.dec_dim, e->dec_qd, e->dec_kvd, d.dec_hidden, d.dec_heads, d.dec_kv_heads, d.dec_head_dim,
                   d.dec_window, d.dec_eps, false};
}

// Encoder transformer on device rows x[n, enc_dim] (in place), then final norm into out.
// Encoder attention of a chunk: window tail in the ring + this chunk's K/V in the merged QKV buffer.
static int enc_attention(vox_hip_engine *e, const RowsCfg &c, float *qkv, float *attn, int n, int pos0, float *kring, float *vring, int ring_cap) {
    const int N3 = c.QD + 2 * c.KVD;
    hipStream_t s = e->stream;
    AttnArgs a{};
    a.out = attn; a.ldo = c.QD; a.q = qkv; a.ldq = N3; a.n_q = n; a.qpos0 = pos0;
    a.kB = qkv + c.QD; a.vB = qkv + c.QD + c.KVD; a.ldB = N3; a.posB0 = pos0; a.last_key = pos0 + n - 1;
    a.kA = kring; a.vA = vring; a.capA = ring_cap; a.ldA = c.KVD;
    a.n_heads = c.heads; a.n_kv_heads = c.kv_heads; a.scale = 1.0f / sqrtf((float)c.hd); a.window = c.window; a.st = nullptr;
    if (n <= 32 && c.hd == 64 && c.heads == c.kv_heads && e->use_attn_small) {
        // streaming-size chunk: (head, 64-key slice) workgroups on plain FMAs + the usual combine
        const int lo = std::max(0, pos0 - c.window + 1), ks = (pos0 + n - 1 - lo) / 64 + 1;
        if (ensure(e, e->spart_o, (size_t)n * c.heads * ks * c.hd * 4)) return -1;
        if (ensure(e, e->spart_ml, (size_t)n * c.heads * ks * 2 * 4)) return -1;
        a.part_o = (float *)e->spart_o.p; a.part_ml = (float *)e->spart_ml.p;
        // (round 5: the key slices of a head are merged by whichever of its workgroups arrives last - no k_attn_combine launch)
        // Up to 16 rows the key slices of a head are merged by whichever of its workgroups arrives last (no k_attn_combine launch);
        // beyond, the merge of 25 - 32 rows by ONE workgroup per head takes longer than the launch it saves.  Same box, us per encoder
        // layer at 1 / 8 / 16 / 25 / 32 rows: 55.3 / 58.5 / 63.6 / 70.9 / 75.8 merged against 58.4 / 61.5 / 64.9 / 69.9 / 74.0 with the
        // launch (profiles/r05_enc_rows_ab.txt).
        const bool merge_here = e->attn_merge && n <= 16;
        a.arrive = merge_here ? e->d_attn_arrive : nullptr;
        if (e->use_dpp) hipLaunchKernelGGL((k_attn_small<true, 8>), dim3(c.heads, ks), dim3(512), 0, s, a, lo);
        else hipLaunchKernelGGL((k_attn_small<false, 8>), dim3(c.heads, ks), dim3(512), 0, s, a, lo);
        if (!merge_here)
            hipLaunchKernelGGL((k_attn_combine<64>), dim3(c.heads, n), dim3(64), 0, s, attn, c.QD,
                               (const float *)a.part_o, (const float *)a.part_ml, c.heads, ks);
        return 0;
    }
    const int qt = (n + 127) / 128, blocks = qt * c.heads;
    const int span = std::min(pos0 + n, c.window + std::min(n, 128));
    int ks = 1;
    if (blocks < 256) ks = std::max(1, std::min((512 + blocks - 1) / blocks, (span + 63) / 64));
    if (ks > 1) {
        if (ensure(e, e->spart_o, (size_t)n * c.heads * ks * c.hd * 4)) return -1;
        if (ensure(e, e->spart_ml, (size_t)n * c.heads * ks * 2 * 4)) return -1;
        a.part_o = (float *)e->spart_o.p; a.part_ml = (float *)e->spart_ml.p;
    }
    a.xcd_map = 1;
    if (e->use_attn_mfma && c.hd == 64 && c.heads == c.kv_heads)
        hipLaunchKernelGGL(k_attn_enc_bf16, dim3(qt, c.heads, ks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_attn_rows<64>), dim3(qt, c.heads, ks), dim3(128), 0, s, a);
    if (ks > 1)
        hipLaunchKernelGGL((k_attn_combine<64>), dim3(c.heads, n), dim3(64), 0, s, attn, c.QD,
                           (const float *)a.part_o, (const float *)a.part_ml, c.heads, ks);
    return 0;
}

// Streaming-size chunks (n <= 32 rows): the weight-streaming path of vox_skinny.h, 7-8 launches per layer.
static bool skinny_ok(const vox_hip_engine *e, int n, const RowsCfg &c) {
    return e->use_skinny && e->use_mfma && n >= 1 && n <= 32 && c.kv_heads == c.heads && c.hd == 64 &&
           c.D % 64 == 0 && c.QD % 64 == 0 && c.H % 64 == 0 && c.D % 32 == 0 && (c.QD + 2 * c.KVD) % 32 == 0 &&
           c.D <= 64 * SK_MAXC * SK_WPB &&      // the unsplit GEMMs (qkv, w1;w3) have K = D: at most SK_MAXC chunks per wave
           c.QD <= 64 * SK_MAXC * SK_WPB * 16 && c.H <= 64 * SK_MAXC * SK_WPB * 16;   // wo / w2: skinny_split() stops at 16 K splits
}
static int skinny_split(int K) { return std::max(1, std::min(16, (K / 64) / SK_WPB)); }      // one chunk per wave when K allows

static int launch_rowsgemm(vox_hip_engine *e, const uint16_t *Xp, size_t xp_plane, const float *X, int ldx, int n,
                           const uint16_t *W, int N, int K, float *partial);
static size_t rg_partial_bytes(int n, int N, int K);

static int encoder_rows_skinny(vox_hip_engine *e, float *x, int n, float *out) {
    const RowsCfg c = enc_cfg(e);
    const int N3 = c.QD + 2 * c.KVD, L = e->d.enc_layers, pos0 = e->enc_pos;
    float *xn = (float *)e->sxn.p, *qkv = (float *)e->sqkv.p, *attn = (float *)e->sattn.p, *tab = (float *)e->srope.p;
    hipStream_t s = e->stream;
    const int so = skinny_split(c.QD), s2 = skinny_split(c.H);
    // (measured and not kept: w2 on k_rowsgemm - 9.1 vs 10.2 us for the launch, but its 20 K splits cost k_rows_finish 7.4 instead of
    //  5.9 us; an L2 prefetch of the next GEMM launch's weight tiles by workgroups appended to the latency-bound launches - the GEMM
    //  launches get 0.0 - 0.6 us shorter, the carriers longer, 67.7 -> 69.9 us per layer: these launches are not waiting for HBM.
    //  profiles/NOTES.md)
    if (ensure(e, e->ssplitk, (size_t)std::max(so, s2) * n * c.D * 4)) return -1;
    // bf16 planes of the normalised rows [3][n][D] and of the gated hidden rows [3][n][H] (sgu is free on this path)
    if (ensure(e, e->sgu, (size_t)3 * n * (c.D + c.H) * 2)) return -1;
    uint16_t *xnp = (uint16_t *)e->sgu.p, *hp = xnp + (size_t)3 * n * c.D;
    float *part = (float *)e->ssplitk.p;
    const size_t lds1 = (size_t)SK_WPB * 4096, lds2 = (size_t)SK_WPB * 2 * 4096;
    if (e->enc_tl_on && !e->d_enc_tl && hipMalloc((void **)&e->d_enc_tl, 4 * 1024 * TL_STRIDE * 8) == hipSuccess)
        hipMemset(e->d_enc_tl, 0, 4 * 1024 * TL_STRIDE * 8);
    auto tlp = [&](int l, int k) { return (e->d_enc_tl && l == L / 2) ? e->d_enc_tl + (size_t)k * 1024 * TL_STRIDE : nullptr; };
    if (L > 0)      // attention_norm of layer 0 (no partials, no bias: x is left as it is)
        hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)nullptr, 0, n, c.D, (const float *)nullptr,
                           (const float *)e->enc[0].n1, c.eps, xn, c.D, xnp, (const float *)nullptr);
    for (int l = 0; l < L; l++) {
        EncLayer &Ly = e->enc[l];
        {   // attention_norm(x) . [wq; wk; wv]^T + bias, RoPE, K/V into the merged buffer and the rings
            SkinnyArgs a{};
            a.Xp = xnp; a.xp_plane = (size_t)n * c.D; a.n = n; a.W = Ly.wqkv; a.N = N3; a.K = c.D; a.bias = Ly.bqkv; a.Y = qkv; a.ldy = N3;
            a.rope_cols = c.QD + c.KVD; a.head_dim = c.hd; a.rope_tab = tab; a.kring = Ly.kring; a.vring = Ly.vring;
            a.ring_cap = e->enc_ring_cap; a.kv_dim = c.KVD; a.pos0 = pos0; a.q_cols = c.QD; a.tl = tlp(l, 0);
            hipLaunchKernelGGL((k_skinny<SK_QKV, 1, true>), dim3(N3 / 32, 1), dim3(64 * SK_WPB), lds1, s, a);
        }
        if (enc_attention(e, c, qkv, attn, n, pos0, Ly.kring, Ly.vring, e->enc_ring_cap)) return -1;
        {   // wo as K-split partials, then x += . + bo and ffn_norm in one launch
            SkinnyArgs a{};
            a.X = attn; a.ldx = c.QD; a.n = n; a.W = Ly.wo; a.N = c.D; a.K = c.QD; a.partial = part; a.tl = tlp(l, 1);
            hipLaunchKernelGGL((k_skinny<SK_PARTIAL, 1, false>), dim3(c.D / 32, so), dim3(64 * SK_WPB), lds1, s, a);
            hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)part, so, n, c.D, (const float *)Ly.bo,
                               (const float *)Ly.n2, c.eps, xn, c.D, xnp, (const float *)nullptr);
        }
        {   // silu(xn w1^T) * (xn w3^T), written as bf16 planes for the w2 launch
            SkinnyArgs a{};
            a.Xp = xnp; a.xp_plane = (size_t)n * c.D; a.n = n; a.W = Ly.w13; a.W2 = Ly.w13 + (size_t)c.H * c.D; a.N = c.H; a.K = c.D;
            a.Yp = hp; a.yp_plane = (size_t)n * c.H; a.tl = tlp(l, 2);
            hipLaunchKernelGGL((k_skinny<SK_SWIGLU, 1, true>), dim3(c.H / 32, 1), dim3(64 * SK_WPB), lds2, s, a);
        }
        {   // w2 partials, then x += . + b2 and the next norm (next layer's attention_norm, or the final norm into `out`)
            SkinnyArgs a{};
            a.Xp = hp; a.xp_plane = (size_t)n * c.H; a.n = n; a.W = Ly.w2; a.N = c.D; a.K = c.H; a.partial = part; a.tl = tlp(l, 3);
            hipLaunchKernelGGL((k_skinny<SK_PARTIAL, 1, true>), dim3(c.D / 32, s2), dim3(64 * SK_WPB), lds1, s, a);
            const bool last = l + 1 == L;
            hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)part, s2, n, c.D, (const float *)Ly.b2,
                               (const float *)(last ? e->enc_final_norm : e->enc[l + 1].n1), c.eps, last ? out : xn, c.D,
                               last ? (uint16_t *)nullptr : xnp, (const float *)nullptr);
        }
    }
    if (L == 0)
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, s, out, c.D, x, c.D, e->enc_final_norm, (const float *)nullptr, c.D, c.eps);
    e->enc_pos += n;
    LAUNCH_CHECK("encoder chunk launches (skinny path)");
    return 0;
}

// ------------------------------------------------------------------------------------
// 1 .. 128 rows on k_rowsgemm (vox_rowsgemm.h): weight tiles across the waves of a workgroup, the activation K range in LDS,
// K split over blockIdx.y, raw partial sums consumed by k_qkv_finish / k_rows_finish / k_swiglu_finish.
// ------------------------------------------------------------------------------------
struct RgPlan { int wpb, cpw, cw, S, nb, ntile; size_t lds; };
static RgPlan rg_plan(int n, int N, int K, bool f32x = false) {
    RgPlan p{};
    const int mt = (n + 31) / 32, nchunks = K / 64;
    p.wpb = N >= 2048 ? 8 : 4;
    // two weight tiles per wave (every activation fragment read from LDS feeds two MFMAs) for the wide GEMMs of <= 64 rows
    p.ntile = (mt <= 2 && N >= 4096) ? 2 : 1;
    p.nb = (N + 32 * p.ntile * p.wpb - 1) / (32 * p.ntile * p.wpb);
    // chunks per round: two LDS stages of 3 planes x 32 mt rows x cpw x 128 B must fit (mt * cpw <= 6: 144 KB)
    p.cpw = mt <= 3 ? 2 : 1;
    // register budget of the 512-thread variants (256 VGPRs): two weight register sets of NB x cpw x 4 fragments + 16 NB mt
    // accumulators (+ the f32 rows of the next round) - one chunk per round where two would spill
    if (p.wpb == 8 && (p.ntile == 2 || f32x || n > 64)) p.cpw = 1;
    p.lds = (size_t)2 * 3 * 32 * mt * p.cpw * 128;
    // one workgroup per CU (the stages take up to 144 KB of its LDS): as many K splits as fill the chip without a second wave
    // of workgroups - 288 workgroups on 256 CUs ran 75 us where 216 ran 54 (gpurun_out/p6)
    const int S = std::max(1, std::min(nchunks, 256 / p.nb));
    int cw = (nchunks + S - 1) / S;
    cw = ((cw + p.cpw - 1) / p.cpw) * p.cpw;
    p.cw = cw; p.S = (nchunks + cw - 1) / cw;
    return p;
}
static bool rowsgemm_ok(const vox_hip_engine *e, int n, const RowsCfg &c) {
    return e->use_rowsgemm && e->use_mfma && n >= 1 && n <= 128 && c.D % 64 == 0 && c.QD % 64 == 0 && c.H % 64 == 0 &&
           (c.QD + 2 * c.KVD) % 4 == 0 && c.hd % 4 == 0 && c.KVD % 4 == 0 && c.QD % 4 == 0;
}
// partial[S][n][N] = x . W^T split over K; x as planes (Xp) or f32 rows (X).  Returns S (the number of partial slices) or -1.
static int launch_rowsgemm(vox_hip_engine *e, const uint16_t *Xp, size_t xp_plane, const float *X, int ldx, int n,
                           const uint16_t *W, int N, int K, float *partial) {
    const RgPlan p = rg_plan(n, N, K, Xp == nullptr);
    RowsGemmArgs a{};
    a.Xp = Xp; a.xp_plane = xp_plane; a.X = X; a.ldx = ldx; a.n = n; a.mt = (n + 31) / 32; a.W = W; a.N = N; a.K = K; a.cw = p.cw;
    a.partial = partial;
    const dim3 grid(p.nb, p.S), block(64 * p.wpb);
    hipStream_t s = e->stream;
#define RG_LAUNCH(WPB, CPW, NBT, MTM)                                                                                    \
    do {                                                                                                                 \
        if (Xp) hipLaunchKernelGGL((k_rowsgemm<WPB, CPW, RG_X_PLANES, NBT, MTM>), grid, block, p.lds, s, a);             \
        else hipLaunchKernelGGL((k_rowsgemm<WPB, CPW, RG_X_F32, NBT, MTM>), grid, block, p.lds, s, a);                   \
    } while (0)
    const bool small = n <= 64;                                  // accumulator budget of 4 (instead of 8) 16-row tiles
    if (p.ntile == 2) {
        if (p.wpb == 8) { if (p.cpw == 2) RG_LAUNCH(8, 2, 2, 4); else RG_LAUNCH(8, 1, 2, 4); }
        else { if (p.cpw == 2) RG_LAUNCH(4, 2, 2, 4); else RG_LAUNCH(4, 1, 2, 4); }
    } else if (small) {
        if (p.wpb == 8) { if (p.cpw == 2) RG_LAUNCH(8, 2, 1, 4); else RG_LAUNCH(8, 1, 1, 4); }
        else { if (p.cpw == 2) RG_LAUNCH(4, 2, 1, 4); else RG_LAUNCH(4, 1, 1, 4); }
    } else {
        if (p.wpb == 8) { if (p.cpw == 2) RG_LAUNCH(8, 2, 1, 8); else RG_LAUNCH(8, 1, 1, 8); }
        else { if (p.cpw == 2) RG_LAUNCH(4, 2, 1, 8); else RG_LAUNCH(4, 1, 1, 8); }
    }
#undef RG_LAUNCH
    return p.S;
}
static size_t rg_partial_bytes(int n, int N, int K) {       // (the f32-activation plan never has fewer chunks per workgroup: its S is the larger one)
    return (size_t)std::max(rg_plan(n, N, K, false).S, rg_plan(n, N, K, true).S) * n * N * 4;
}

// fp8 mode (BASELINE config 5): partial[S][n][N] = x . W8^T on the fp8 MFMA (k_rowsgemm_f8, vox_rowsgemm_f8.h), n <= 64 rows of f32
// activations, W8 = row-scaled e4m3.  Returns S or -1.  The activations' pre-scale: normalised rows, attention outputs and gated
// hidden rows of this model stay far below 224 / 2 in magnitude, and e4m3 keeps its 4 significant bits down to 2^-6 / 2.
constexpr int RGF8_WPB = 8, RGF8_CPW = 2;
static int rgf8_splits(int N, int K) {
    const int nb = (N + 32 * RGF8_WPB - 1) / (32 * RGF8_WPB), nchunks = K / 64;
    const int S = std::max(1, std::min(nchunks / RGF8_CPW, 256 / nb));
    int cw = (nchunks + S - 1) / S;
    cw = ((cw + RGF8_CPW - 1) / RGF8_CPW) * RGF8_CPW;
    return (nchunks + cw - 1) / cw;
}
static size_t rgf8_partial_bytes(int n, int N, int K) { return (size_t)rgf8_splits(N, K) * n * N * 4; }
static int launch_rowsgemm_f8(vox_hip_engine *e, const float *X, int ldx, int n, const uint8_t *W8, const float *wscale, int N, int K, float *partial) {
    if (n < 1 || n > 64 || K % 64 || ldx % 4) { g_err = "rowsgemm_f8: needs 1 <= n <= 64, K % 64 == 0"; return -1; }
    const int nb = (N + 32 * RGF8_WPB - 1) / (32 * RGF8_WPB), nchunks = K / 64;
    const int S0 = std::max(1, std::min(nchunks / RGF8_CPW, 256 / nb));
    int cw = (nchunks + S0 - 1) / S0;
    cw = ((cw + RGF8_CPW - 1) / RGF8_CPW) * RGF8_CPW;
    const int S = (nchunks + cw - 1) / cw;
    RowsGemmF8Args a{};
    a.X = X; a.ldx = ldx; a.n = n; a.W = W8; a.wscale = wscale; a.N = N; a.K = K; a.cw = cw; a.prescale = 2.0f; a.partial = partial; a.clamped = e->d_f8_clamped;
    const dim3 grid(nb, S), block(64 * RGF8_WPB);
    switch ((n + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_rowsgemm_f8<RGF8_WPB, RGF8_CPW, 1>), grid, block, 0, e->stream, a); break;
        case 2: hipLaunchKernelGGL((k_rowsgemm_f8<RGF8_WPB, RGF8_CPW, 2>), grid, block, 0, e->stream, a); break;
        case 3: hipLaunchKernelGGL((k_rowsgemm_f8<RGF8_WPB, RGF8_CPW, 3>), grid, block, 0, e->stream, a); break;
        default: hipLaunchKernelGGL((k_rowsgemm_f8<RGF8_WPB, RGF8_CPW, 4>), grid, block, 0, e->stream, a); break;
    }
    return S;
}

// Decoder attention of `n` prefill rows over the KV ring (the rows' own K/V are in the ring already).
static int dec_attention_rows(vox_hip_engine *e, const RowsCfg &c, float *qkv, float *attn, int n, int pos0, float *kring, float *vring, int ring_cap) {
    hipStream_t s = e->stream;
    const int N3 = c.QD + 2 * c.KVD;
    const int max_len = std::min(pos0 + n, c.window);
    const int nsplit = (max_len + DEC_SPLIT_KEYS - 1) / DEC_SPLIT_KEYS;
    AttnArgs a{};
    a.out = attn; a.ldo = c.QD; a.q = qkv; a.ldq = N3; a.n_q = n; a.qpos0 = pos0;
    a.kB = nullptr; a.vB = nullptr; a.ldB = 0; a.posB0 = INT_MAX; a.last_key = pos0 + n - 1;
    a.kA = kring; a.vA = vring; a.capA = ring_cap; a.ldA = c.KVD;
    a.n_heads = c.heads; a.n_kv_heads = c.kv_heads; a.scale = 1.0f / sqrtf((float)c.hd); a.window = c.window; a.st = nullptr;
    a.split_keys = DEC_SPLIT_KEYS;
    if (nsplit > 1) {
        if (ensure(e, e->spart_o, (size_t)n * c.heads * nsplit * c.hd * 4)) return -1;
        if (ensure(e, e->spart_ml, (size_t)n * c.heads * nsplit * 2 * 4)) return -1;
        a.part_o = (float *)e->spart_o.p; a.part_ml = (float *)e->spart_ml.p;
    }
    if (e->use_dpp) hipLaunchKernelGGL((k_attn_dec<128, 4, true>), dim3(c.kv_heads, nsplit, n), dim3(256), 0, s, a, nsplit);
    else hipLaunchKernelGGL((k_attn_dec<128, 4, false>), dim3(c.kv_heads, nsplit, n), dim3(256), 0, s, a, nsplit);
    if (nsplit > 1)
        hipLaunchKernelGGL((k_attn_combine<128>), dim3(c.heads, n), dim3(128), 0, s, attn, c.QD,
                           (const float *)a.part_o, (const float *)a.part_ml, c.heads, nsplit);
    return 0;
}

// All layers of one stack on n <= 128 rows x[n][D] (in place).  Encoder: final norm into `out` (f32 rows); decoder prefill:
// out = nullptr, only the KV rings matter (voxtral_decoder.c:410-558).  The RoPE table of the chunk is in e->srope.
// Per layer: qkv GEMM, qkv finish (bias, RoPE, KV append), attention (+ combine), wo GEMM, finish (residual + ffn_norm (+ ada)
// -> planes), w1;w3 GEMM, SwiGLU finish (-> planes), w2 GEMM, finish (residual + the next norm -> planes) = 9 - 10 launches.
static int rows_mid_layers(vox_hip_engine *e, float *x, int n, int pos0, const RowsCfg &c, bool is_enc, float *out) {
    const int N3 = c.QD + 2 * c.KVD, L = is_enc ? e->d.enc_layers : e->d.dec_layers;
    hipStream_t s = e->stream;
    size_t pb = std::max(std::max(rg_partial_bytes(n, N3, c.D), rg_partial_bytes(n, c.D, c.QD)),
                         std::max(rg_partial_bytes(n, 2 * c.H, c.D), rg_partial_bytes(n, c.D, c.H)));
    if (ensure(e, e->ssplitk, pb)) return -1;
    if (ensure(e, e->sgu, (size_t)3 * n * (c.D + c.H) * 2)) return -1;       // bf16 planes of the normalised rows and of the gated hidden rows
    if (ensure(e, e->sqkv, (size_t)n * N3 * 4) || ensure(e, e->sattn, (size_t)n * c.QD * 4)) return -1;
    // fp8 mode (BASELINE config 5): the decoder prefill reads the row-scaled e4m3 copies of its matrices and multiplies on the fp8 MFMA
    // (k_rowsgemm_f8); the normalised rows and the gated hidden rows then travel as f32 (the kernel splits them into e4m3 terms)
    const bool f8 = !is_enc && e->use_fp8 && e->use_mfma && !e->fp8_prefill_bf16 && n <= 64 && e->dec[0].wqkv8;
    if (f8) {
        pb = std::max(std::max(rgf8_partial_bytes(n, N3, c.D), rgf8_partial_bytes(n, c.D, c.QD)),
                      std::max(rgf8_partial_bytes(n, 2 * c.H, c.D), rgf8_partial_bytes(n, c.D, c.H)));
        if (ensure(e, e->ssplitk, pb) || ensure(e, e->sxn, (size_t)n * c.D * 4) || ensure(e, e->sgu, (size_t)n * c.H * 4)) return -1;
    }
    float *xnf = f8 ? (float *)e->sxn.p : nullptr, *hf = f8 ? (float *)e->sgu.p : nullptr;
    uint16_t *xnp = f8 ? (uint16_t *)nullptr : (uint16_t *)e->sgu.p, *hp = f8 ? (uint16_t *)nullptr : xnp + (size_t)3 * n * c.D;
    float *part = (float *)e->ssplitk.p, *qkv = (float *)e->sqkv.p, *attn = (float *)e->sattn.p, *tab = (float *)e->srope.p;
    const int ring_cap = is_enc ? e->enc_ring_cap : e->dec_ring_cap;
    // the chunk's K/V rows may go to their ring slots before attention when they cannot overwrite a row the window still needs
    const bool append_early = !is_enc || n <= ring_cap - c.window;
    auto norm_of = [&](int l, int which) -> const float * {      // which: 0 attention_norm, 1 ffn_norm
        if (is_enc) return which ? e->enc[l].n2 : e->enc[l].n1;
        return which ? e->dec[l].n2 : e->dec[l].n1;
    };
    if (L > 0)
        hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)nullptr, 0, n, c.D, (const float *)nullptr,
                           norm_of(0, 0), c.eps, xnf, c.D, xnp, (const float *)nullptr);
    for (int l = 0; l < L; l++) {
        const uint16_t *wqkv = is_enc ? e->enc[l].wqkv : e->dec[l].wqkv, *wo = is_enc ? e->enc[l].wo : e->dec[l].wo;
        const uint16_t *w13 = is_enc ? e->enc[l].w13 : e->dec[l].w13, *w2 = is_enc ? e->enc[l].w2 : e->dec[l].w2;
        float *kring = is_enc ? e->enc[l].kring : e->dec[l].kring, *vring = is_enc ? e->enc[l].vring : e->dec[l].vring;
        const float *bqkv = is_enc ? e->enc[l].bqkv : nullptr, *bo = is_enc ? e->enc[l].bo : nullptr, *b2 = is_enc ? e->enc[l].b2 : nullptr;
        const float *ada = is_enc ? nullptr : e->dec[l].ada;
        int S = f8 ? launch_rowsgemm_f8(e, xnf, c.D, n, e->dec[l].wqkv8, e->dec[l].sqkv, N3, c.D, part)
                   : launch_rowsgemm(e, xnp, (size_t)n * c.D, nullptr, 0, n, wqkv, N3, c.D, part);
        if (S < 0) return -1;
        hipLaunchKernelGGL(k_qkv_finish, dim3(grid1d((size_t)n * N3 / 4)), dim3(256), 0, s, qkv, N3, (const float *)part, S, n, bqkv,
                           (const float *)tab, c.QD + c.KVD, c.hd, append_early ? kring : (float *)nullptr, vring, ring_cap, c.KVD, pos0, c.QD);
        if (is_enc) {
            if (enc_attention(e, c, qkv, attn, n, pos0, kring, vring, ring_cap)) return -1;
            if (!append_early) {
                const int keep = std::min(n, c.window);
                hipLaunchKernelGGL(k_ring_append, dim3(grid1d((size_t)keep * c.KVD / 4)), dim3(256), 0, s,
                                   kring, vring, ring_cap, c.KVD, qkv, N3, c.QD, c.QD + c.KVD, n - keep, keep, pos0 + n - keep);
            }
        } else if (dec_attention_rows(e, c, qkv, attn, n, pos0, kring, vring, ring_cap)) return -1;
        S = f8 ? launch_rowsgemm_f8(e, attn, c.QD, n, e->dec[l].wo8, e->dec[l].so, c.D, c.QD, part)
               : launch_rowsgemm(e, nullptr, 0, attn, c.QD, n, wo, c.D, c.QD, part);
        if (S < 0) return -1;
        hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)part, S, n, c.D, bo,
                           norm_of(l, 1), c.eps, xnf, c.D, xnp, ada);
        S = f8 ? launch_rowsgemm_f8(e, xnf, c.D, n, e->dec[l].w138, e->dec[l].s13, 2 * c.H, c.D, part)
               : launch_rowsgemm(e, xnp, (size_t)n * c.D, nullptr, 0, n, w13, 2 * c.H, c.D, part);
        if (S < 0) return -1;
        hipLaunchKernelGGL(k_swiglu_finish, dim3(grid1d((size_t)n * c.H / 4)), dim3(256), 0, s, hp, (size_t)n * c.H, (const float *)part, S, n, c.H, hf);
        S = f8 ? launch_rowsgemm_f8(e, hf, c.H, n, e->dec[l].w28, e->dec[l].s2, c.D, c.H, part)
               : launch_rowsgemm(e, hp, (size_t)n * c.H, nullptr, 0, n, w2, c.D, c.H, part);
        if (S < 0) return -1;
        const bool last = l + 1 == L;
        const float *next_norm = last ? (is_enc ? e->enc_final_norm : (const float *)nullptr) : norm_of(l + 1, 0);
        hipLaunchKernelGGL(k_rows_finish, dim3(n), dim3(rf_threads(c.D)), 0, s, x, c.D, (const float *)part, S, n, c.D, b2,
                           next_norm, c.eps, last ? out : xnf, c.D, last ? (uint16_t *)nullptr : xnp, (const float *)nullptr);
    }
    if (L == 0 && out)
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, s, out, c.D, x, c.D, e->enc_final_norm, (const float *)nullptr, c.D, c.eps);
    LAUNCH_CHECK("rows-gemm layer launches");
    return 0;
}

// ------------------------------------------------------------------------------------
// Round 6: a streaming-size chunk (n <= 32 rows) through all encoder layers as ONE persistent launch (k_enc_stack,
// vox_encstack.h) + the final norm.  x is NOT modified (a flagged chunk is repeated on the 8-launch path from the same rows).
// ------------------------------------------------------------------------------------
static bool enc_stack_usable(const vox_hip_engine *e, int n) {
    return e->enc_stack_ok && e->enc_stack_rearm <= 0 && n >= 1 && n <= 32 && e->d.enc_layers >= 1 &&
           e->d.enc_window + 31 <= ES_NSL * 128 && e->enc_ring_cap >= e->d.enc_window + 32;
}
static int enc_stack_init(vox_hip_engine *e) {
    if (e->enc_stack_ready) return 0;
    const int L = e->d.enc_layers;
    int rc = 0;
    rc |= dalloc(e, &e->d_es_tab, (size_t)L);
    rc |= dalloc(e, &e->d_es_xa, (size_t)32 * ES_D); rc |= dalloc(e, &e->d_es_xb, (size_t)32 * ES_D);
    rc |= dalloc(e, &e->d_es_ssq, (size_t)2 * 32 * ES_CB); rc |= dalloc(e, &e->d_es_q, (size_t)32 * ES_QD);
    rc |= dalloc(e, &e->d_es_po, (size_t)ES_HEADS * ES_NSL * 32 * 64); rc |= dalloc(e, &e->d_es_pml, (size_t)ES_HEADS * ES_NSL * 32 * 2);
    rc |= dalloc(e, &e->d_es_wop, (size_t)ES_HEADS * 32 * ES_D); rc |= dalloc(e, &e->d_es_w2p, (size_t)ES_KG5 * 32 * ES_D);
    rc |= dalloc(e, &e->d_es_apl, (size_t)ES_KS_D * 3 * 2 * 512); rc |= dalloc(e, &e->d_es_hpl, (size_t)ES_KS_H * 3 * 2 * 512);
    rc |= dalloc(e, &e->d_es_flags, (size_t)ES_WGS); rc |= dalloc(e, &e->d_es_err, (size_t)16);
    if (rc) return -1;
    if (hipHostMalloc((void **)&e->h_es_err, 16 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
    memset(e->h_es_err, 0, 16 * sizeof(unsigned));
    std::vector<EncStackLayer> tab((size_t)L);
    for (int l = 0; l < L; l++) {
        EncLayer &Y = e->enc[l];
        tab[l] = EncStackLayer{Y.wqkv, Y.wo, Y.w13, Y.w13 + (size_t)ES_H * ES_D, Y.w2, Y.bqkv, Y.bo, Y.b2, Y.n1, Y.n2, Y.kring, Y.vring};
    }
    HC(hipMemcpy(e->d_es_tab, tab.data(), tab.size() * sizeof(EncStackLayer), hipMemcpyHostToDevice));
    HC(hipMemset(e->d_es_apl, 0, (size_t)ES_KS_D * 3 * 2 * 1024)); HC(hipMemset(e->d_es_hpl, 0, (size_t)ES_KS_H * 3 * 2 * 1024));
    HC(hipMemset(e->d_es_flags, 0, ES_WGS * 4)); HC(hipMemset(e->d_es_err, 0, 16 * 4));
    HC(hipMemset(e->d_es_xa, 0, (size_t)32 * ES_D * 4)); HC(hipMemset(e->d_es_xb, 0, (size_t)32 * ES_D * 4));
    if (e->enc_tl_on && hipMalloc((void **)&e->d_es_tl, (size_t)ES_WGS * ES_TL_STRIDE * 8) == hipSuccess) hipMemset(e->d_es_tl, 0, (size_t)ES_WGS * ES_TL_STRIDE * 8);
    const void *fns[] = {(const void *)k_enc_stack<1, false>, (const void *)k_enc_stack<2, false>, (const void *)k_enc_stack<1, true>, (const void *)k_enc_stack<2, true>};
    for (const void *f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ES_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); return -1; }
    e->enc_epoch = 0;
    e->enc_stack_ready = true;
    return 0;
}
static int encoder_rows_stack(vox_hip_engine *e, const float *x, int n, float *out) {
    if (enc_stack_init(e)) return -1;
    hipStream_t s = e->stream;
    if (e->enc_epoch > 0xF0000000u) { HC(hipMemsetAsync(e->d_es_flags, 0, ES_WGS * 4, s)); e->enc_epoch = 0; }
    EncStackArgs a{};
    a.layers = e->d_es_tab; a.n_layers = e->d.enc_layers; a.n = n; a.pos0 = e->enc_pos; a.ring_cap = e->enc_ring_cap; a.window = e->d.enc_window;
    a.eps = e->d.enc_eps; a.scale = 1.0f / sqrtf((float)ES_HD); a.x_in = x; a.rope_tab = (const float *)e->srope.p;
    a.xa = e->d_es_xa; a.xb = e->d_es_xb; a.aplanes = e->d_es_apl; a.ssq = e->d_es_ssq; a.qbuf = e->d_es_q; a.part_o = e->d_es_po; a.part_ml = e->d_es_pml;
    a.wo_part = e->d_es_wop; a.hplanes = e->d_es_hpl; a.w2_part = e->d_es_w2p; a.flags = e->d_es_flags; a.epoch = e->enc_epoch;
    a.err = e->d_es_err; a.spin_limit = 500000ull; a.tl = e->d_es_tl; a.tl_layer = e->d.enc_layers / 2;
    e->enc_epoch += 256u * (unsigned)((ES_PHASES * e->d.enc_layers + 1 + 255) / 256);
    if (e->d_es_tl) {
        if (n <= 16) hipLaunchKernelGGL((k_enc_stack<1, true>), dim3(ES_WGS), dim3(ES_THREADS), ES_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((k_enc_stack<2, true>), dim3(ES_WGS), dim3(ES_THREADS), ES_LDS_BYTES, s, a);
    } else {
        if (n <= 16) hipLaunchKernelGGL((k_enc_stack<1, false>), dim3(ES_WGS), dim3(ES_THREADS), ES_LDS_BYTES, s, a);
        else hipLaunchKernelGGL((k_enc_stack<2, false>), dim3(ES_WGS), dim3(ES_THREADS), ES_LDS_BYTES, s, a);
    }
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, s, out, ES_D, (const float *)e->d_es_xa, ES_D, e->enc_final_norm, (const float *)nullptr, ES_D, e->d.enc_eps);
    e->enc_pos += n;
    e->enc_stack_pending = true; e->enc_stack_launches++;
    LAUNCH_CHECK("encoder chunk launches (stack kernel)");
    return 0;
}
// Enqueue the read-back of the stack kernel's error words (before the caller's host wait), and judge them after it.
static void enc_stack_fetch(vox_hip_engine *e) {
    if (e->enc_stack_pending) hipMemcpyAsync(e->h_es_err, e->d_es_err, 16 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream);
}
constexpr long ENC_STACK_REARM_CHUNKS = 64;
// After the host wait: 1 = a hand-off of the stack kernel timed out in a chunk enqueued since the last check (its results are void:
// the caller repeats the chunk, which now takes the 8-launch path), 0 = clean.
static int enc_stack_failed(vox_hip_engine *e) {
    if (!e->enc_stack_pending) return 0;
    e->enc_stack_pending = false;
    const unsigned *w = e->h_es_err;
    e->spin_hole_max = std::max(e->spin_hole_max, w[8]); e->spin_holes += w[9];
    if (w[8] | w[9]) (void)hipMemsetAsync(e->d_es_err + 8, 0, 2 * sizeof(unsigned), e->stream);
    if (e->enc_stack_inject) { e->enc_stack_inject = false; e->h_es_err[0] = 99u; }
    if (!w[0]) return 0;
    e->enc_stack_failures++;
    e->enc_stack_rearm = (ENC_STACK_REARM_CHUNKS << std::min(e->enc_stack_failures - 1, 6)) + 1;      // (+ 1: the repeat of this chunk counts one down)
    fprintf(stderr, "vox_hip: ERROR the encoder stack kernel timed out in a hand-off (code %u; its 256 workgroups were not co-resident?): workgroup %u "
                    "(XCD %u) waited for flag value %u, %.1f us of active waiting, longest gap between two polls %.1f us; repeating the chunk on the "
                    "launch-per-GEMM path and staying there for %ld chunks\n", w[0], w[1], w[2] & 15u, w[6], w[4] / 100.0, w[3] / 100.0, e->enc_stack_rearm - 1);
    (void)hipMemset(e->d_es_err, 0, 8 * sizeof(unsigned));
    memset(e->h_es_err, 0, 16 * sizeof(unsigned));
    return 1;
}

static int encoder_rows_dev(vox_hip_engine *e, float *x, int n, float *out) {
    const RowsCfg c = enc_cfg(e);
    if (ensure_rows_scratch(e, n, c)) return -1;
    hipLaunchKernelGGL(k_rope_table, dim3(grid1d((size_t)n * c.hd / 2)), dim3(256), 0, e->stream,
                       (float *)e->srope.p, e->enc_inv_freq, e->enc_pos, n, c.hd / 2);
    if (rowsgemm_ok(e, n, c) && (n > 32 || !skinny_ok(e, n, c))) {
        if (rows_mid_layers(e, x, n, e->enc_pos, c, true, out)) return -1;
        e->enc_pos += n;
        return 0;
    }
    if (skinny_ok(e, n, c)) {
        if (e->enc_stack_rearm > 0) e->enc_stack_rearm--;              // a suspension of the stack kernel counts down in chunks
        else if (e->enc_stack_now && enc_stack_usable(e, n)) return encoder_rows_stack(e, x, n, out);
        return encoder_rows_skinny(e, x, n, out);
    }
    uint16_t *xplanes = nullptr;          // planes of the next layer's normalised input, when the previous layer's W2 reduce pass left them
    for (int l = 0; l < e->d.enc_layers; l++) {
        EncLayer &L = e->enc[l];
        if (run_layer_rows(e, x, n, e->enc_pos, c, L.wqkv, L.bqkv, L.wo, L.bo, L.w13, L.w2, L.b2, L.n1, L.n2,
                           nullptr, L.kring, L.vring, e->enc_ring_cap, l + 1 < e->d.enc_layers ? e->enc[l + 1].n1 : nullptr, &xplanes)) return -1;
    }
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, e->stream, out, c.D, x, c.D, e->enc_final_norm,
                       (const float *)nullptr, c.D, c.eps);
    e->enc_pos += n;
    LAUNCH_CHECK("encoder chunk launches");
    return 0;
}

// y[M][N] = act(x[M][K] . W^T + bias) for the small GEMMs around the encoder stack (conv stem as im2col GEMMs, adapter):
// at streaming sizes (M = 6 .. 50 rows, 10 - 31 MB of weights each) the 128 x 128 tiles + split-K took 18 - 52 us per launch;
// <= 32 rows go through k_rowsgemm (f32 rows split in the kernel) + the same fixed-order reduce with the fused epilogue.
static int gemm_small_rows(vox_hip_engine *e, const float *X, int ldx, const uint16_t *W, float *Y, int ldy, int M, int N, int K,
                           const float *bias, int act) {
    // (<= 32 rows only: conv1 13.9 -> 10.7 us, adapter0 19.6 -> 14.9 us at 25 / 6 rows; at the flush pass's 34 - 68 rows the whole
    // encode got 0.14 ms SLOWER on the 2-layer model - gpurun_out/p14 - so those stay on the 128 x 128 tiles)
    if (e->use_rowsgemm && e->use_mfma && M >= 1 && M <= 32 && K % 64 == 0 && ldx % 4 == 0) {
        if (ensure(e, e->ssplitk, rg_partial_bytes(M, N, K))) return -1;
        const int S = launch_rowsgemm(e, nullptr, 0, X, ldx, M, W, N, K, (float *)e->ssplitk.p);
        GemmArgs a{nullptr, 0, W, Y, ldy, M, N, K, bias, nullptr, 0, act, S, 0, (float *)e->ssplitk.p};
        hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, a);
        return 0;
    }
    return launch_gemm(e, X, ldx, W, Y, ldy, M, N, K, bias, nullptr, 0, act);
}

// Adapter on device: in [m*4, enc_dim] contiguous == [m, 4*enc_dim]; out [m, dec_dim].
static int adapter_dev(vox_hip_engine *e, const float *in, int m, float *out) {
    const int DD = e->d.dec_dim, K0 = e->d.enc_dim * 4;
    if (ensure(e, e->smid, (size_t)m * DD * 4)) return -1;
    float *mid = (float *)e->smid.p;
    if (gemm_small_rows(e, in, K0, e->adapter0, mid, DD, m, DD, K0, nullptr, ACT_GELU)) return -1;
    if (gemm_small_rows(e, mid, DD, e->adapter1, out, DD, m, DD, DD, nullptr, ACT_NONE)) return -1;
    LAUNCH_CHECK("adapter launches");
    return 0;
}

// ------------------------------------------------------------------------------------
// mel + conv stem (device-resident boundary state)
//
// conv_in0: [2 + cap, mel_bins]   rows 0,1 = last two mel frames of the previous chunk
//                                 (zeros at stream start = causal left pad, voxtral.c:567-575),
//                                 rows 2.. = queued frames not yet consumed.
// conv_in1: [2 + cap, enc_dim]    row 0 = c0[2p-1] (last conv0 frame consumed by conv1, zero at
//                                 start), row 1 = the odd conv0 frame carried over (if c0_carry),
//                                 then this chunk's conv0 frames.
// In global indices: c0[f] = gelu(W0.[mel[f-2],mel[f-1],mel[f]] + b0),
//                    x[p]  = gelu(W1.[c0[2p-1],c0[2p],c0[2p+1]] + b1)   (voxtral.c:537-715)
// ------------------------------------------------------------------------------------
extern "C" int vox_hip_mel_frames(vox_hip_engine_t *e, const float *samples, int n_frames, float *out_mel, int to_queue) {
    if (!e || n_frames <= 0) return e ? 0 : -1;
    HC(hipSetDevice(e->device));
    const int MB = e->d.mel_bins;
    const size_t ns = (size_t)(n_frames - 1) * MEL_HOP + MEL_NFFT;
    if (ensure(e, e->ssamples, ns * 4)) return -1;
    // A streaming feed's samples (33 KB at -I 0.5) go through a pinned staging slot: the copy is asynchronous, the caller's buffer is free when
    // this returns, and the host goes on to enqueue the conv stem while the mel kernel runs (round 6; before: a host wait per feed).  A slot
    // is reused four feeds later, behind its own event.  Large transfers (a clip in one feed) keep the blocking form.
    const bool staged = to_queue && !out_mel && ns * 4 <= vox_hip_engine::SMP_SLOT_BYTES && e->smp_pin[0];
    int slot = -1;
    if (staged) {
        slot = e->smp_next; e->smp_next = (e->smp_next + 1) % vox_hip_engine::SMP_SLOTS;
        if (e->smp_used[slot]) HC(hipEventSynchronize(e->smp_ev[slot]));
        memcpy(e->smp_pin[slot], samples, ns * 4);
        HC(hipMemcpyAsync(e->ssamples.p, e->smp_pin[slot], ns * 4, hipMemcpyHostToDevice, e->stream));
        HC(hipEventRecord(e->smp_ev[slot], e->stream));
        e->smp_used[slot] = true;
    } else {
        HC(hipMemcpyAsync(e->ssamples.p, samples, ns * 4, hipMemcpyHostToDevice, e->stream));
    }
    float *dst;
    if (to_queue) {
        if (ensure_keep(e, e->conv_in0, (size_t)(2 + e->mel_q + n_frames) * MB * 4, (size_t)(2 + e->mel_q) * MB * 4)) return -1;
        dst = (float *)e->conv_in0.p + (size_t)(2 + e->mel_q) * MB;
    } else {
        if (ensure(e, e->stmp_out, (size_t)n_frames * MB * 4)) return -1;
        dst = (float *)e->stmp_out.p;
    }
    hipLaunchKernelGGL(k_mel_frames, dim3(n_frames), dim3(256), 0, e->stream, dst, MB, (const float *)e->ssamples.p,
                       e->hann, e->cosT, e->sinT, e->filtT);
    if (out_mel) HC(hipMemcpyAsync(out_mel, dst, (size_t)n_frames * MB * 4, hipMemcpyDeviceToHost, e->stream));
    if (!staged) HC(esync(e));   // the host sample buffer may be reused by the caller
    if (to_queue) e->mel_q += n_frames;
    return 0;
}

// Runs the conv stem over the first n frames of the mel queue. Output rows (device) are
// written to xout [rows, enc_dim]; returns rows.
static int conv_stem_dev(vox_hip_engine *e, int n, float **xout) {
    const vox_hip_dims_t &d = e->d;
    const int MB = d.mel_bins, ED = d.enc_dim;
    hipStream_t s = e->stream;
    *xout = nullptr;
    if (n <= 0) return 0;
    if (apply_enc_fences(e)) return -1;
    float *in0 = (float *)e->conv_in0.p;
    if (ensure_keep(e, e->conv_in1, (size_t)(2 + n + 1) * ED * 4, (size_t)2 * ED * 4)) return -1;
    float *in1 = (float *)e->conv_in1.p;
    // conv0 as GEMM over im2col rows
    if (ensure(e, e->sim2col, (size_t)n * std::max(MB, ED) * 3 * 4)) return -1;
    float *col = (float *)e->sim2col.p;
    hipLaunchKernelGGL(k_im2col3, dim3(grid1d((size_t)n * MB * 3)), dim3(256), 0, s, col, (const float *)in0, n, MB, 1);
    float *c0_new = in1 + (size_t)(1 + e->c0_carry) * ED;
    if (gemm_small_rows(e, col, MB * 3, e->conv0_w, c0_new, ED, n, ED, MB * 3, e->conv0_b, ACT_GELU)) return -1;
    // roll the mel history: rows 0,1 <- last two frames seen. n == 1 reproduces the
    // reference's tail update, which zeroes the older slot (voxtral.c:604-609, tc == 1).
    if (n >= 2) {
        HC(hipMemcpyAsync(in0, in0 + (size_t)n * MB, (size_t)2 * MB * 4, hipMemcpyDeviceToDevice, s));
    } else {
        HC(hipMemsetAsync(in0, 0, (size_t)MB * 4, s));
        HC(hipMemcpyAsync(in0 + MB, in0 + (size_t)2 * MB, (size_t)MB * 4, hipMemcpyDeviceToDevice, s));
    }
    // remaining queued frames (if any) slide to the front of the queue
    const int left = e->mel_q - n;
    if (left > 0) {
        // non-overlapping when left <= n; otherwise go through scratch
        if (left <= n) {
            HC(hipMemcpyAsync(in0 + (size_t)2 * MB, in0 + (size_t)(2 + n) * MB, (size_t)left * MB * 4, hipMemcpyDeviceToDevice, s));
        } else {
            if (ensure(e, e->stmp_in, (size_t)left * MB * 4)) return -1;
            HC(hipMemcpyAsync(e->stmp_in.p, in0 + (size_t)(2 + n) * MB, (size_t)left * MB * 4, hipMemcpyDeviceToDevice, s));
            HC(hipMemcpyAsync(in0 + (size_t)2 * MB, e->stmp_in.p, (size_t)left * MB * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    e->mel_q = left;

    // conv1 (stride 2) over [hist | carry | new]
    const int pending = e->c0_carry + n;
    const int nq = pending / 2;
    if (nq > 0) {
        hipLaunchKernelGGL(k_im2col3, dim3(grid1d((size_t)nq * ED * 3)), dim3(256), 0, s, col, (const float *)in1, nq, ED, 2);
        if (ensure(e, e->sx, (size_t)nq * ED * 4)) return -1;
        float *x = (float *)e->sx.p;
        if (gemm_small_rows(e, col, ED * 3, e->conv1_w, x, ED, nq, ED, ED * 3, e->conv1_b, ACT_GELU)) return -1;
        *xout = x;
        // history row <- last consumed conv0 frame; carry row <- odd leftover
        HC(hipMemcpyAsync(in1, in1 + (size_t)(2 * nq) * ED, (size_t)ED * 4, hipMemcpyDeviceToDevice, s));
        if (pending & 1)
            HC(hipMemcpyAsync(in1 + ED, in1 + (size_t)(2 * nq + 1) * ED, (size_t)ED * 4, hipMemcpyDeviceToDevice, s));
    }
    // (nq == 0: the single pending frame already sits in the carry slot)
    e->c0_carry = pending & 1;
    return nq;
}

extern "C" int vox_hip_conv_stem(vox_hip_engine_t *e, const float *mel_new, int n_mel, float *out, int out_cap_rows) {
    if (!e || n_mel <= 0) return e ? 0 : -1;
    HC(hipSetDevice(e->device));
    const int MB = e->d.mel_bins, ED = e->d.enc_dim;
    if (ensure_keep(e, e->conv_in0, (size_t)(2 + e->mel_q + n_mel) * MB * 4, (size_t)(2 + e->mel_q) * MB * 4)) return -1;
    HC(hipMemcpyAsync((float *)e->conv_in0.p + (size_t)(2 + e->mel_q) * MB, mel_new, (size_t)n_mel * MB * 4,
                      hipMemcpyHostToDevice, e->stream));
    e->mel_q += n_mel;
    float *x = nullptr;
    const int rows = conv_stem_dev(e, e->mel_q, &x);
    if (rows < 0) return -1;
    if (out && rows > 0) {
        if (rows > out_cap_rows) { g_err = "vox_hip_conv_stem: output buffer too small"; return -1; }
        HC(hipMemcpyAsync(out, x, (size_t)rows * ED * 4, hipMemcpyDeviceToHost, e->stream));
    }
    HC(esync(e));
    return rows;
}

// The reference's BATCH conv stem (vox_encoder_forward -> vox_causal_conv1d, voxtral_encoder.c:135-176,
// voxtral_kernels.c:293-340) right-pads an odd number of conv0 frames with one zero frame, so it
// emits ceil(L/2) rows where the stream path keeps the unpaired frame waiting.  This computes that
// last row from the carried frame and a zero partner; out_row [enc_dim] (host).  Returns 1 if a row
// was produced, 0 if nothing was pending, -1 on error.
extern "C" int vox_hip_conv_stem_pad_odd(vox_hip_engine_t *e, float *out_row) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    if (!e->c0_carry) return 0;
    const int ED = e->d.enc_dim;
    hipStream_t s = e->stream;
    if (ensure_keep(e, e->conv_in1, (size_t)3 * ED * 4, (size_t)2 * ED * 4)) return -1;
    float *in1 = (float *)e->conv_in1.p;
    HC(hipMemsetAsync(in1 + (size_t)2 * ED, 0, (size_t)ED * 4, s));
    if (ensure(e, e->sim2col, (size_t)ED * 3 * 4)) return -1;
    if (ensure(e, e->sx, (size_t)ED * 4)) return -1;
    float *col = (float *)e->sim2col.p, *x = (float *)e->sx.p;
    hipLaunchKernelGGL(k_im2col3, dim3(grid1d((size_t)ED * 3)), dim3(256), 0, s, col, (const float *)in1, 1, ED, 2);
    if (launch_gemm(e, col, ED * 3, e->conv1_w, x, ED, 1, ED, ED * 3, e->conv1_b, nullptr, 0, ACT_GELU)) return -1;
    HC(hipMemcpyAsync(in1, in1 + (size_t)2 * ED, (size_t)ED * 4, hipMemcpyDeviceToDevice, s));   // history <- the zero frame
    e->c0_carry = 0;
    if (out_row) HC(hipMemcpyAsync(out_row, x, (size_t)ED * 4, hipMemcpyDeviceToHost, s));
    HC(esync(e));
    return 1;
}

// ------------------------------------------------------------------------------------
// stage-level entry points on host buffers
// ------------------------------------------------------------------------------------
extern "C" int vox_hip_encoder_chunk(vox_hip_engine_t *e, const float *x_new, int new_len, float *out) {
    if (!e || new_len <= 0) return -1;
    HC(hipSetDevice(e->device));
    if (apply_enc_fences(e)) return -1;
    const int ED = e->d.enc_dim;
    if (ensure(e, e->stmp_in, (size_t)new_len * ED * 4)) return -1;
    if (ensure(e, e->stmp_out, (size_t)new_len * ED * 4)) return -1;
    const int pos_before = e->enc_pos;
    for (int attempt = 0; attempt < 2; attempt++) {
        // (the paths below the stack kernel work in place: the rows are uploaded again for a repeat)
        HC(hipMemcpyAsync(e->stmp_in.p, x_new, (size_t)new_len * ED * 4, hipMemcpyHostToDevice, e->stream));
        e->enc_stack_now = attempt == 0;
        const int erc = encoder_rows_dev(e, (float *)e->stmp_in.p, new_len, (float *)e->stmp_out.p);
        e->enc_stack_now = false;
        if (erc) return -1;
        HC(hipMemcpyAsync(out, e->stmp_out.p, (size_t)new_len * ED * 4, hipMemcpyDeviceToHost, e->stream));
        enc_stack_fetch(e);
        HC(esync(e));
        if (!enc_stack_failed(e)) break;
        e->enc_pos = pos_before;          // a hand-off of the stack kernel timed out: the same rows again, on the launch-per-GEMM path
    }
    return 0;
}

extern "C" int vox_hip_adapter(vox_hip_engine_t *e, const float *enc_out, int enc_len, float *out) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    const int ED = e->d.enc_dim, DD = e->d.dec_dim;
    const int m = enc_len / 4;
    if (m <= 0) return 0;
    if (ensure(e, e->stmp_in, (size_t)m * 4 * ED * 4)) return -1;
    if (ensure(e, e->stmp_out, (size_t)m * DD * 4)) return -1;
    HC(hipMemcpyAsync(e->stmp_in.p, enc_out, (size_t)m * 4 * ED * 4, hipMemcpyHostToDevice, e->stream));
    if (adapter_dev(e, (const float *)e->stmp_in.p, m, (float *)e->stmp_out.p)) return -1;
    HC(hipMemcpyAsync(out, e->stmp_out.p, (size_t)m * DD * 4, hipMemcpyDeviceToHost, e->stream));
    HC(esync(e));
    return m;
}

// ------------------------------------------------------------------------------------
// adapter buffer management
// ------------------------------------------------------------------------------------
static int adapter_reserve(vox_hip_engine *e, int64_t extra_rows) {
    const int DD = e->d.dec_dim;
    int64_t phys = e->adapter_total - e->adapter_row0;
    if (phys + extra_rows <= e->adapter_cap) return 0;
    if (drain_fences(e)) return -1;          // rows are about to move: nobody may still be writing into the old buffer
    // drop rows the decoder has already consumed (stream_adapter_compact, voxtral.c:718-731)
    const int64_t dead = std::min(e->adapter_consumed - e->adapter_row0, phys);
    if (dead > 0) {
        const int64_t live = phys - dead;
        HC(esync(e));
        if (live > 0) {
            if (ensure(e, e->stmp_in, (size_t)live * DD * 4)) return -1;
            HC(hipMemcpy(e->stmp_in.p, e->adapter + (size_t)dead * DD, (size_t)live * DD * 4, hipMemcpyDeviceToDevice));
            HC(hipMemcpy(e->adapter, e->stmp_in.p, (size_t)live * DD * 4, hipMemcpyDeviceToDevice));
        }
        e->adapter_row0 += dead;
        phys = live;
        if (phys + extra_rows <= e->adapter_cap) return 0;
    }
    int64_t ncap = e->adapter_cap;
    while (ncap < phys + extra_rows) ncap *= 2;
    float *np = nullptr;
    HC(esync(e));
    HC(hipMalloc((void **)&np, (size_t)ncap * DD * 4));
    if (phys > 0) HC(hipMemcpy(np, e->adapter, (size_t)phys * DD * 4, hipMemcpyDeviceToDevice));
    HC(hipFree(e->adapter));
    e->mem_used += (size_t)(ncap - e->adapter_cap) * DD * 4;
    e->adapter = np; e->adapter_cap = ncap;
    return 0;
}

extern "C" int64_t vox_hip_adapter_rows(const vox_hip_engine_t *e) { return e ? e->adapter_total : 0; }
extern "C" void *vox_hip_adapter_devptr(vox_hip_engine_t *e, int64_t *cap_rows) {
    if (!e) return nullptr;
    if (cap_rows) *cap_rows = e->adapter_cap;
    return e->adapter;
}
extern "C" int vox_hip_adapter_read(vox_hip_engine_t *e, int64_t first_row, int n_rows, float *out) {
    if (!e || n_rows <= 0) return -1;
    HC(hipSetDevice(e->device));
    if (first_row < e->adapter_row0 || first_row + n_rows > e->adapter_total) { g_err = "vox_hip_adapter_read: rows not resident"; return -1; }
    if (apply_row_fences(e, first_row + n_rows - 1)) return -1;
    HC(esync(e));
    HC(hipMemcpy(out, e->adapter + (size_t)(first_row - e->adapter_row0) * e->d.dec_dim,
                 (size_t)n_rows * e->d.dec_dim * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int vox_hip_adapter_append(vox_hip_engine_t *e, const float *rows, int n_rows) {
    if (!e || n_rows <= 0) return -1;
    HC(hipSetDevice(e->device));
    if (adapter_reserve(e, n_rows)) return -1;
    HC(hipMemcpy(e->adapter + (size_t)(e->adapter_total - e->adapter_row0) * e->d.dec_dim, rows,
                 (size_t)n_rows * e->d.dec_dim * 4, hipMemcpyHostToDevice));
    e->adapter_total += n_rows;
    return 0;
}

// ------------------------------------------------------------------------------------
// fused streaming encode: mel queue -> conv stem -> encoder -> 4x alignment -> adapter
// enc_out buffer: [3 + cap, enc_dim]; rows (3-enc_res)..2 hold the carried rows so that
// [carried | new] is contiguous (voxtral.c:824-890).
// ------------------------------------------------------------------------------------
extern "C" int vox_hip_stream_encode(vox_hip_engine_t *e, int n_mel, int *conv_rows, int *enc_residual) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    if (n_mel > e->mel_q) { g_err = "vox_hip_stream_encode: more frames requested than queued"; return -1; }
    const int ED = e->d.enc_dim, DD = e->d.dec_dim;
    hipStream_t s = e->stream;
    HC(hipEventRecord(e->ev0, s));
    float *x = nullptr;
    const int nq = conv_stem_dev(e, n_mel, &x);
    if (nq < 0) return -1;
    if (conv_rows) *conv_rows = nq;
    int new_tokens = 0;
    // (round 6) the stack kernel may flag a chunk (a hand-off timed out): everything from the encoder stack on is then repeated on the
    // launch-per-GEMM path from the same conv-stem rows, with the stream state rewound to this point
    const int saved_pos = e->enc_pos, saved_res = e->enc_res; const int64_t saved_total = e->adapter_total;
    bool waited = false;
    for (int attempt = 0; attempt < 2 && nq > 0; attempt++) {
        new_tokens = 0;
        if (ensure_keep(e, e->enc_out, (size_t)(3 + nq) * ED * 4, (size_t)3 * ED * 4)) return -1;
        float *eo = (float *)e->enc_out.p;
        // the rows carried for the 4x alignment (rows 0 .. 2) are overwritten at the end of the bracket: keep a copy while a repeat is possible
        const bool may_repeat = attempt == 0 && skinny_ok(e, nq, enc_cfg(e)) && enc_stack_usable(e, nq) && !(rowsgemm_ok(e, nq, enc_cfg(e)) && nq > 32);
        if (may_repeat && saved_res > 0) {
            if (ensure(e, e->es_carry, (size_t)3 * ED * 4)) return -1;
            HC(hipMemcpyAsync(e->es_carry.p, eo, (size_t)3 * ED * 4, hipMemcpyDeviceToDevice, s));
        }
        if (attempt == 1 && saved_res > 0) HC(hipMemcpyAsync(eo, e->es_carry.p, (size_t)3 * ED * 4, hipMemcpyDeviceToDevice, s));
        e->enc_stack_now = attempt == 0;
        const int erc = encoder_rows_dev(e, x, nq, eo + (size_t)3 * ED);
        e->enc_stack_now = false;
        if (erc) return -1;
        const int total = e->enc_res + nq;
        const int usable = (total / 4) * 4, leftover = total - usable;
        float *first = eo + (size_t)(3 - e->enc_res) * ED;
        if (usable > 0) {
            const int m = usable / 4;
            if (adapter_reserve(e, m)) return -1;
            float *dst = e->adapter + (size_t)(e->adapter_total - e->adapter_row0) * DD;
            if (adapter_dev(e, first, m, dst)) return -1;
            e->adapter_total += m;
            new_tokens = m;
        }
        // carry the trailing rows: move them to rows (3-leftover)..2, ascending order
        for (int i = 0; i < leftover; i++) {
            float *src = first + (size_t)(usable + i) * ED;
            float *dst = eo + (size_t)(3 - leftover + i) * ED;
            if (src != dst) HC(hipMemcpyAsync(dst, src, (size_t)ED * 4, hipMemcpyDeviceToDevice, s));
        }
        e->enc_res = leftover;
        if (!e->enc_stack_pending) break;
        HC(hipEventRecord(e->ev1, s));           // (the timing event in front of the ONE host wait of this call)
        enc_stack_fetch(e);
        HC(esync(e));
        waited = true;
        if (!enc_stack_failed(e)) break;
        waited = false;
        e->enc_pos = saved_pos; e->enc_res = saved_res; e->adapter_total = saved_total;       // (the carried rows 0 .. 2 of enc_out were only read)
    }
    if (enc_residual) *enc_residual = e->enc_res;
    if (!waited) {
        HC(hipEventRecord(e->ev1, s));
        HC(esync(e));
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e->ev0, e->ev1);
    e->timing.encode_ms += ms;
    return new_tokens;
}

// ------------------------------------------------------------------------------------
// Multi-GPU encoder sharding (option E1: exact context parallelism).
// Each rank owns a contiguous range of encoder positions.  All ranks walk the 32 layers;
// before layer l a rank imports the layer-l K/V of the window-1 positions preceding its
// range (sent by its left neighbour right after that neighbour finished layer l) into its
// position-indexed ring, so attention sees exactly what a single GPU would.  The exchange
// itself (RCCL send/recv over xGMI) is done by the caller on device pointers.
// ------------------------------------------------------------------------------------

extern "C" int vox_hip_shard_begin(vox_hip_engine_t *e, int n_mel, int discard_rows, int pos0) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    if (n_mel > e->mel_q) { g_err = "vox_hip_shard_begin: more frames requested than queued"; return -1; }
    float *x = nullptr;
    const int nq = conv_stem_dev(e, n_mel, &x);
    if (nq < 0) return -1;
    const int n = nq - discard_rows;
    if (n <= 0) { g_err = "vox_hip_shard_begin: empty shard"; return -1; }
    e->shard_x = x + (size_t)discard_rows * e->d.enc_dim;
    e->shard_n = n;
    e->enc_pos = pos0;
    const RowsCfg c = enc_cfg(e);
    if (ensure_rows_scratch(e, n, c)) return -1;
    hipLaunchKernelGGL(k_rope_table, dim3(grid1d((size_t)n * c.hd / 2)), dim3(256), 0, e->stream,
                       (float *)e->srope.p, e->enc_inv_freq, pos0, n, c.hd / 2);
    return n;
}

extern "C" int vox_hip_shard_layer(vox_hip_engine_t *e, int layer) {
    if (!e || layer < 0 || layer >= e->d.enc_layers || !e->shard_x) return -1;
    HC(hipSetDevice(e->device));
    const RowsCfg c = enc_cfg(e);
    EncLayer &L = e->enc[layer];
    return run_layer_rows(e, e->shard_x, e->shard_n, e->enc_pos, c, L.wqkv, L.bqkv, L.wo, L.bo, L.w13, L.w2, L.b2,
                          L.n1, L.n2, nullptr, L.kring, L.vring, e->enc_ring_cap);
}

// K then V rows of positions [pos_first, pos_first+n) of `layer`: dst [2][n][kv_dim] (device).  The copies are only
// ENQUEUED on the engine stream (a consumer ordered behind that stream - RCCL on the same stream, an event - needs no
// host wait); vox_hip_shard_kv_export is the same followed by a host synchronisation (host-staged transports: gloo).
extern "C" int vox_hip_shard_kv_export_async(vox_hip_engine_t *e, int layer, int pos_first, int n, void *dst_dev) {
    if (!e || layer < 0 || layer >= e->d.enc_layers || n <= 0) return -1;
    HC(hipSetDevice(e->device));
    const int kvd = e->enc_qd, cap = e->enc_ring_cap;
    float *dst = (float *)dst_dev;
    for (int i = 0; i < n;) {            // contiguous runs inside the ring
        const int slot = (pos_first + i) % cap;
        const int run = std::min(n - i, cap - slot);
        HC(hipMemcpyAsync(dst + (size_t)i * kvd, e->enc[layer].kring + (size_t)slot * kvd, (size_t)run * kvd * 4, hipMemcpyDeviceToDevice, e->stream));
        HC(hipMemcpyAsync(dst + (size_t)(n + i) * kvd, e->enc[layer].vring + (size_t)slot * kvd, (size_t)run * kvd * 4, hipMemcpyDeviceToDevice, e->stream));
        i += run;
    }
    return 0;
}
extern "C" int vox_hip_shard_kv_export(vox_hip_engine_t *e, int layer, int pos_first, int n, void *dst_dev) {
    if (vox_hip_shard_kv_export_async(e, layer, pos_first, n, dst_dev)) return -1;
    HC(esync(e));
    return 0;
}

extern "C" int vox_hip_shard_kv_import(vox_hip_engine_t *e, int layer, int pos_first, int n, const void *src_dev) {
    if (!e || layer < 0 || layer >= e->d.enc_layers || n <= 0) return -1;
    HC(hipSetDevice(e->device));
    const int kvd = e->enc_qd, cap = e->enc_ring_cap;
    if (n > cap) { g_err = "vox_hip_shard_kv_import: more rows than the ring holds"; return -1; }
    const float *src = (const float *)src_dev;
    for (int i = 0; i < n;) {
        const int slot = (pos_first + i) % cap;
        const int run = std::min(n - i, cap - slot);
        HC(hipMemcpyAsync(e->enc[layer].kring + (size_t)slot * kvd, src + (size_t)i * kvd, (size_t)run * kvd * 4, hipMemcpyDeviceToDevice, e->stream));
        HC(hipMemcpyAsync(e->enc[layer].vring + (size_t)slot * kvd, src + (size_t)(n + i) * kvd, (size_t)run * kvd * 4, hipMemcpyDeviceToDevice, e->stream));
        i += run;
    }
    return 0;
}

// Final norm + adapter over the shard rows (shard_n must be a multiple of 4). Writes
// [shard_n/4, dec_dim] rows to dst_dev (device) and returns the row count.
extern "C" int vox_hip_shard_end_async(vox_hip_engine_t *e, void *dst_dev) {
    if (!e || !e->shard_x || !dst_dev) return -1;
    HC(hipSetDevice(e->device));
    const int n = e->shard_n, ED = e->d.enc_dim;
    if (n % 4) { g_err = "vox_hip_shard_end: shard rows must be a multiple of 4"; return -1; }
    if (ensure(e, e->stmp_out, (size_t)n * ED * 4)) return -1;
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, e->stream, (float *)e->stmp_out.p, ED, e->shard_x, ED,
                       e->enc_final_norm, (const float *)nullptr, ED, e->d.enc_eps);
    if (adapter_dev(e, (const float *)e->stmp_out.p, n / 4, (float *)dst_dev)) return -1;
    e->shard_x = nullptr; e->shard_n = 0;
    return n / 4;
}
extern "C" int vox_hip_shard_end(vox_hip_engine_t *e, void *dst_dev) {
    const int m = vox_hip_shard_end_async(e, dst_dev);
    if (m < 0) return -1;
    HC(esync(e));
    return m;
}

// Append adapter rows that already live in device memory (gathered over xGMI).
extern "C" int vox_hip_adapter_append_dev_async(vox_hip_engine_t *e, const void *rows_dev, int n_rows) {
    if (!e || n_rows <= 0) return -1;
    HC(hipSetDevice(e->device));
    if (adapter_reserve(e, n_rows)) return -1;           // (waits only when the buffer has to grow or compact)
    HC(hipMemcpyAsync(e->adapter + (size_t)(e->adapter_total - e->adapter_row0) * e->d.dec_dim, rows_dev,
                      (size_t)n_rows * e->d.dec_dim * 4, hipMemcpyDeviceToDevice, e->stream));
    e->adapter_total += n_rows;
    return 0;
}
extern "C" int vox_hip_adapter_append_dev(vox_hip_engine_t *e, const void *rows_dev, int n_rows) {
    if (vox_hip_adapter_append_dev_async(e, rows_dev, n_rows)) return -1;
    HC(esync(e));
    return 0;
}

extern "C" void *vox_hip_device_alloc(vox_hip_engine_t *e, size_t bytes) {
    if (!e) return nullptr;
    if (hipSetDevice(e->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return p;
}
extern "C" void vox_hip_device_free(vox_hip_engine_t *e, void *p) { if (e && p) { hipSetDevice(e->device); hipFree(p); } }
extern "C" int vox_hip_memcpy(vox_hip_engine_t *e, void *dst, const void *src, size_t bytes, int kind) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    HC(hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------
static int decoder_prefill_dev(vox_hip_engine *e, float *x, int n) {
    const RowsCfg c = dec_cfg(e);
    for (int off = 0; off < n;) {
        // prompts of up to 128 rows (the stream's 38) run as one k_rowsgemm chunk, longer ones in 512-row GEMM chunks
        const int m = (n - off <= 128 && e->use_rowsgemm) ? n - off : std::min(PREFILL_CHUNK, n - off);
        if (ensure_rows_scratch(e, m, c)) return -1;
        hipLaunchKernelGGL(k_rope_table, dim3(grid1d((size_t)m * c.hd / 2)), dim3(256), 0, e->stream,
                           (float *)e->srope.p, e->dec_inv_freq, e->dec_pos, m, c.hd / 2);
        float *xc = x + (size_t)off * c.D;
        if (rowsgemm_ok(e, m, c)) {
            if (rows_mid_layers(e, xc, m, e->dec_pos, c, false, nullptr)) return -1;
            e->dec_pos += m; off += m;
            continue;
        }
        for (int l = 0; l < e->d.dec_layers; l++) {
            DecLayer &L = e->dec[l];
            if (run_layer_rows(e, xc, m, e->dec_pos, c, L.wqkv, nullptr, L.wo, nullptr, L.w13, L.w2, nullptr,
                               L.n1, L.n2, L.ada, L.kring, L.vring, e->dec_ring_cap)) return -1;
        }
        e->dec_pos += m; off += m;
    }
    LAUNCH_CHECK("decoder prefill launches");
    return 0;
}

static int f8_prefill_clamped(vox_hip_engine *e);
extern "C" int vox_hip_decoder_prefill(vox_hip_engine_t *e, const float *embeds, int seq_len) {
    if (!e || seq_len <= 0) return -1;
    HC(hipSetDevice(e->device));
    const int DD = e->d.dec_dim;
    if (ensure(e, e->sx, (size_t)seq_len * DD * 4)) return -1;
    const int pos0 = e->dec_pos;
    for (int attempt = 0; attempt < 2; attempt++) {
        HC(hipMemcpyAsync(e->sx.p, embeds, (size_t)seq_len * DD * 4, hipMemcpyHostToDevice, e->stream));      // (the rows pass works in place)
        if (decoder_prefill_dev(e, (float *)e->sx.p, seq_len)) return -1;
        HC(esync(e));
        if (!f8_prefill_clamped(e)) break;
        e->dec_pos = pos0;
    }
    return 0;
}

// Keys per attention block for a decode step at KV length kv_len: short contexts are latency
// bound, so they get small slices (64 keys = one 16-key trip per wave); long contexts get
// bigger ones so that the number of partials stays <= 64.
static int dec_split_keys(int kv_len) {
    if (kv_len <= 512) return 64;
    if (kv_len <= 3072) return 128;        // measured: 2500 keys 1.875 ms (128) vs 1.915 ms (256), 4000 keys 1.987 vs 1.979
    if (kv_len <= 8192) return 256;
    return 512;
}

template <int PRO, int EPI, int RPW, int CPL, int KS, int MINW, bool W8 = false>
static void launch_gemv3(vox_hip_engine *e, const GemvArgs &a) {
    constexpr int K = CPL * KS * 64 * (W8 ? 16 : 8);
    const int rows_per_block = (4 / KS) * RPW;
    const int grid = (a.N + rows_per_block - 1) / rows_per_block;
    size_t fl = K + 64;
    if (PRO == PRO_RMS || PRO == PRO_EMBED_RMS) fl += 2 * (size_t)K;
    if (PRO == PRO_ATTN) fl += 256;
    hipLaunchKernelGGL((k_gemv3<PRO, EPI, RPW, CPL, KS, MINW, W8>), dim3(grid), dim3(256), fl * sizeof(float), e->stream, a);
}

// The end of a decoder pass over one row x[dec_dim] of the stack's output: final norm -> tied-embedding logits -> per-block argmax
// (voxtral_decoder.c:694-704), then the argmax finish, which also advances the decoder cursor.
static void enqueue_logits_tail(vox_hip_engine *e, const float *xin, float *logits_dst, int eos, int advance, bool fast) {
    const vox_hip_dims_t &d = e->d;
    const int DD = d.dec_dim;
    hipStream_t s = e->stream;
    GemvArgs a{};
    a.W = (e->sim_on && e->sim_lm) ? e->tok_emb_s : e->tok_emb; a.x = xin; a.norm_w = e->dec_final_norm; a.eps = d.dec_eps; a.y = logits_dst;
    a.N = d.vocab; a.K = DD; a.blk_val = e->blk_val; a.blk_idx = e->blk_idx;
    // VOX_HIP_DISABLE=fp8_lmhead (agreement study): fp8 mode with the LM head on the bf16 embedding
    if (fast && e->use_fp8 && !e->fp8_lmhead_bf16) {
        a.W = reinterpret_cast<const uint16_t *>(e->tok_emb8); a.wscale = e->stok;
        hipLaunchKernelGGL((k_gemv<PRO_RMS, EPI_LOGITS, 4, true>), dim3(e->logits_grid), dim3(256),
                           ((size_t)DD + 16) * sizeof(float), s, a);
    } else
        launch_gemv<PRO_RMS, EPI_LOGITS, 4>(e, a, e->logits_grid);
    prof_mark(e, PK_LOGITS);
    hipLaunchKernelGGL(k_argmax_finish, dim3(1), dim3(256), 0, s, (const float *)e->blk_val, (const int *)e->blk_idx,
                       e->logits_grid, e->d_st, e->d_tokens, eos, advance);
    prof_mark(e, PK_ARGMAX);
}

// Enqueue one decode step. kv_pos = logical position of this token (host mirror of st->pos).
static int enqueue_step(vox_hip_engine *e, int kv_pos, bool build_embed, float *logits_dst, int eos, int advance) {
    const vox_hip_dims_t &d = e->d;
    const int DD = d.dec_dim, DQ = e->dec_qd, DKV = e->dec_kvd, DH = d.dec_hidden, HD = d.dec_head_dim;
    hipStream_t s = e->stream;
    // production kernels are specialised for the 4B shapes; anything else takes the generic ones
    const bool fast = e->use_fast && DD == 3072 && DQ == 4096 && DKV == 1024 && DH == 9216;
    if (!fast) {
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(256), 0, s, (const DecState *)e->d_st, (const float *)e->dec_inv_freq,
                           HD / 2, e->dec_rope, e->dx, (const float *)e->adapter, (const uint16_t *)e->tok_emb, DD, build_embed ? 1 : 0);
        prof_mark(e, PK_BEGIN);
    }
    // Hand-off granules are tagged with launch epochs (up to 2 per layer and step).  Long before the 32-bit counter runs out (~50 h of
    // decoding) every granule buffer is zeroed in stream order and the counter restarts: a slot that only some step shapes write (key
    // slices 9 .. 32, the fp8 forms) must not meet its own old tag again.
    if (e->fuse_epoch > 0xFFF00000u && e->d_gq) {
        HC(hipMemsetAsync(e->d_gq, 0, (size_t)DF_GROUPS * DF_GQ * 8, s)); HC(hipMemsetAsync(e->d_gp, 0, (size_t)DF_GROUPS * DF_BPG * DF_GP * 8, s));
        HC(hipMemsetAsync(e->d_gh, 0, (size_t)FFN_H * 8, s));
        if (e->d_gx) { HC(hipMemsetAsync(e->d_gx, 0, (size_t)DF_D * 8, s)); HC(hipMemsetAsync(e->d_gw, 0, (size_t)8 * DF_D * 8, s)); HC(hipMemsetAsync(e->d_gxp, 0, (size_t)DF_D * 8, s)); }
        e->fuse_epoch = 0;
    }
    const int kv_len = std::min(kv_pos + 1, d.dec_window);
    const int split_keys = fast ? dec_split_keys(kv_len) : DEC_SPLIT_KEYS;
    const int nsplit = (kv_len + split_keys - 1) / split_keys;
    const bool fuse_combine = fast && nsplit <= 8;
    const float scale = 1.0f / sqrtf((float)HD);
    const bool fused = fast && e->use_fused && e->use_dpp;      // the FFN kernels reduce with DPP row sums (self-tested at start-up).  fp8 mode: qkv / wo stay on the bf16 matrices (k_dec_attn_fused), w1;w3 and w2 stream the fp8 copies
    // fused path: key slices of the attention stage (<= 32 per KV head, multiples of 64 keys)
    int tap_i = -1;
    for (size_t i = 0; i < e->tap_pos.size(); i++) if (e->tap_pos[i] == kv_pos) tap_i = (int)i;
    // what the 12-wave shape needs apart from the slice count (bf16 form; fp8 form)
    const bool pf_ok12 = e->pf_units == 0 || (e->pf_when == 3 && e->pf_member_units == 0);
    const bool static12 = fused && e->merge12 > 0 && e->use_ffn && !e->use_fp8 && !e->sim_on && (e->skip_kinds & ~(1u << PK_W2)) == 0 && tap_i < 0 && pf_ok12;
    const bool static12_f8 = fused && e->merge12 == 2 && e->use_fp8 && !e->fp8_attn_bf16 && !e->sim_on && (e->skip_kinds & ~(1u << PK_SWIGLU)) == 0 && tap_i < 0 && pf_ok12;
    int f_split = 64, f_ns = 1;
    if (fused) {
        while ((kv_len + f_split - 1) / f_split > DF_BPG) f_split += 64;
        // the merged launches (k_dec_stack / k_ffn_attn12 / k_w2x_attn12) take up to 8 key slices: beyond 512 keys, up to merge12_maxkeys, of more than
        // one tile each - only where such a launch will actually run (debug taps, A/B rungs and the simulated-fp8 study keep the 8-wave
        // kernel's own schedule: 64-key slices)
        if (e->merge12 == 2 && (static12 || static12_f8) && kv_len > 512 && kv_len <= e->merge12_maxkeys) f_split = ((kv_len + 7) / 8 + 63) / 64 * 64;
        f_ns = (kv_len + f_split - 1) / f_split;
    }
    // Fused path: the residual stream ping-pongs between two buffers.  k_gemv_w13x's block 0 writes x' = x + sum(wo partials)
    // while the other 255 workgroups may not have fetched x yet - in place that is a race that only bites when workgroups
    // of one launch start far apart (two models sharing the GPU: found by test_two_decoders_sharing_the_gpu_stay_correct).
    float *xin = e->dx, *xalt = e->dx2;
    auto tap = [&](int slot, const float *src) {      // slot 2l: layer l's input, 2l + 1: after its attention block, 2L: the stack's output
        if (tap_i < 0) return;
        hipMemcpyAsync(e->d_taps + ((size_t)tap_i * (2 * d.dec_layers + 1) + slot) * DD, src, (size_t)DD * 4, hipMemcpyDeviceToDevice, s);
    };
    const int tl_layer = 13;          // VOX_HIP_FUSE_TL: the mid-stack layer whose blocks are stamped
    // the 12-wave shape (k_attn12 / k_ffn_attn12): <= 8 key slices (up to merge12_maxkeys keys), 9 .. 32 in its LONG form (round 5); bf16, DPP, no debug hooks
    const bool long12 = f_ns > 8;
    const bool shape12 = static12 && (f_ns <= 8 || e->merge12_long);
    // fp8 mode: the W2 launch of layer l and the (fp8) attention block of layer l + 1 as one launch, same regime
    const bool shape12_f8 = static12_f8 && f_ns <= 8;
    bool attn_done = false;           // this layer's attention block ran at the end of the previous layer's launch (k_ffn_attn12)
    // k_dec_stack: every block of the step's layers in ONE launch - with the embedding gather (embed = 1: attention(0) included) or behind
    // layer 0's own attention launch (embed = 0: the first step after a prefill, whose x is in memory)
    const bool stack_here = shape12 && !long12 && e->merge12 == 2 && e->use_stack && d.dec_layers > 1;
    auto launch_stack = [&](int embed) -> int {
        std::vector<DecStackLayer> tab((size_t)d.dec_layers);
        for (int k = 0; k < d.dec_layers; k++) {
            DecLayer &K = e->dec[k];
            tab[k] = DecStackLayer{K.wqkv, K.wo, K.w13, K.w13 + (size_t)DH * DD, K.w2, K.n1, K.n2, K.ada, K.kring, K.vring};
        }
        if (e->h_stack_tab.size() != tab.size() || memcmp(e->h_stack_tab.data(), tab.data(), tab.size() * sizeof(DecStackLayer)) != 0) {
            HC(hipMemcpyAsync(e->d_stack_tab, tab.data(), tab.size() * sizeof(DecStackLayer), hipMemcpyHostToDevice, s));
            HC(esync(e));                         // (rare: the first step, or after the rings / ada vectors moved) the source is a local; counted like every host wait
            e->h_stack_tab = tab;
        }
        DecStackArgs sa{};
        sa.layers = e->d_stack_tab; sa.n_layers = d.dec_layers; sa.eps = d.dec_eps; sa.inv_freq = e->dec_inv_freq;
        sa.kv_cap = e->dec_ring_cap; sa.pos = kv_pos; sa.window = d.dec_window; sa.scale = scale;
        sa.x0 = xin; sa.wo_part = e->d_wo_part; sa.x_out = xalt;
        sa.embed = embed; sa.adapter = e->adapter; sa.tok_emb = e->tok_emb; sa.st = e->d_st;
        sa.gq = e->d_gq; sa.gp = e->d_gp; sa.gh = e->d_gh; sa.gx = e->d_gx; sa.gw = e->d_gw; sa.gxp = e->d_gxp;
        if (embed && ++e->fuse_epoch == 0) e->fuse_epoch = 1;          // (embed = 0: layer 0's attention launch took this epoch already)
        sa.epoch0 = e->fuse_epoch; sa.split_keys = f_split; sa.nsplit = f_ns;
        sa.err = e->d_fuse_err; sa.spin_limit = 500000ull;
        e->fuse_epoch += (unsigned)d.dec_layers;                 // (layer l of the launch tags with epoch0 + l; the counter restarts at the top of a step)
        sa.tl = e->d_fuse_tl; sa.tl_layer = tl_layer;
        if (e->skip_kinds & (1u << PK_W2)) {}        // (timing experiment, kind 6: the step without this launch)
        else if (e->d_fuse_tl) hipLaunchKernelGGL(k_dec_stack<true>, dim3(256), dim3(FFN_THREADS), FA12_LDS_BYTES, s, sa);      // VOX_HIP_FUSE_TL: the instrumented build
        else hipLaunchKernelGGL(k_dec_stack<false>, dim3(256), dim3(FFN_THREADS), FA12_LDS_BYTES, s, sa);
        prof_mark(e, PK_W2);
        std::swap(xin, xalt);
        return 0;
    };
    for (int l = 0; l < d.dec_layers; l++) {
        DecLayer &L = e->dec[l];
        if (fused) {
            if (l == 0 && stack_here && build_embed && !(e->skip_kinds & (1u << PK_QKV))) {
                if (launch_stack(1)) return -1;
                break;
            }
            if (!attn_done && !(e->skip_kinds & (1u << PK_QKV))) {
                // attention_norm -> wq/wk/wv -> RoPE -> KV append -> attention -> wo (K-split partials): one launch
                DecFuseArgs a{};
                a.wqkv = e->sim_on ? L.wqkv_s : L.wqkv; a.wo = e->sim_on ? L.wo_s : L.wo; a.x = xin; a.norm_w = L.n1; a.eps = d.dec_eps; a.inv_freq = e->dec_inv_freq;
                a.kring = L.kring; a.vring = L.vring; a.kv_cap = e->dec_ring_cap; a.pos = kv_pos; a.window = d.dec_window; a.scale = scale;
                a.adapter = e->adapter; a.tok_emb = e->tok_emb; a.st = e->d_st; a.x_out = xin;
                a.gq = e->d_gq; a.gp = e->d_gp; a.wo_part = e->d_wo_part;
                if (++e->fuse_epoch == 0) e->fuse_epoch = 1;
                a.epoch = e->fuse_epoch; a.split_keys = f_split; a.nsplit = f_ns;
                a.err = e->d_fuse_err; a.spin_limit = 500000ull;         // 5 ms at the 100 MHz wall clock (a hand-off takes microseconds)
                a.tl = (l == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl : nullptr;
                if (e->pf_units > 0) {        // the next launch's (k_gemv_w13x) first bytes, see DfPrefetch
                    a.pf.w = e->use_fp8 ? reinterpret_cast<const unsigned char *>(L.w138) : reinterpret_cast<const unsigned char *>(L.w13);
                    a.pf.row_bytes = e->use_fp8 ? DD : 2 * DD; a.pf.rows_m = DH;
                    a.pf.units = std::min(e->pf_units, 72 * (e->use_fp8 ? 3 : 6)); a.pf.member_units = e->pf_member_units; a.pf.when = e->pf_when;
                }
                const bool emb = (l == 0 && build_embed);
                // VOX_HIP_DISABLE=fp8_attn (A/B): the round-3 state, qkv / wo on the bf16 matrices
                if (e->use_fp8 && e->use_dpp && !e->fp8_attn_bf16) {
                    // fp8 mode (BASELINE config 5): the projection and Wo matrices stream their row-scaled e4m3 copies too
                    a.wqkv = reinterpret_cast<const uint16_t *>(L.wqkv8); a.wo = reinterpret_cast<const uint16_t *>(L.wo8);
                    a.sqkv = L.sqkv; a.so = L.so;
                    if (emb) hipLaunchKernelGGL((k_dec_attn_fused<true, true, true>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                    else hipLaunchKernelGGL((k_dec_attn_fused<false, true, true>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                } else if (shape12 && !emb) {
                    if (long12) hipLaunchKernelGGL(k_attn12<true>, dim3(DF_BLOCKS), dim3(DA12_THREADS), DA12_LDS_BYTES, s, a);
                    else hipLaunchKernelGGL(k_attn12<false>, dim3(DF_BLOCKS), dim3(DA12_THREADS), DA12_LDS_BYTES, s, a);
                } else if (e->use_dpp) {
                    if (emb) hipLaunchKernelGGL((k_dec_attn_fused<true, true>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                    else hipLaunchKernelGGL((k_dec_attn_fused<false, true>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                } else {
                    if (emb) hipLaunchKernelGGL((k_dec_attn_fused<true, false>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                    else hipLaunchKernelGGL((k_dec_attn_fused<false, false>), dim3(DF_BLOCKS), dim3(DF_THREADS), DF_LDS_BYTES, s, a);
                }
                prof_mark(e, PK_QKV);
            }
            attn_done = false;
            tap(2 * l, xin);              // (layer 0 of a stream step builds x inside the launch above and writes it to xin)
            // (fp8 mode keeps the two launches: with half the W2 bytes the 72 KB sweep and the dot products at the end are no longer
            //  hidden - measured 1.1055 against 1.0271 ms per step, gpurun_out/r4k)
            if (l == 0 && stack_here) {       // (layer 0's attention block ran as a launch of its own: the first step after a prefill)
                if (launch_stack(0)) return -1;
                break;
            }
            const bool merged_here = shape12 && e->merge12 == 2 && l + 1 < d.dec_layers;
            if (merged_here && (e->skip_kinds & (1u << PK_W2))) {      // timing experiment: the step without its k_ffn_attn12 launches
                if (++e->fuse_epoch == 0) e->fuse_epoch = 1;
                attn_done = true;
                std::swap(xin, xalt);
                continue;
            }
            if (e->use_ffn && !e->use_fp8 && !e->sim_on && !(e->skip_kinds & (1u << PK_SWIGLU)) && ((shape12 && e->merge12 == 2) || !(e->skip_kinds & (1u << PK_W2)))) {
                // the whole FFN block as one launch (k_ffn_fused): x' -> ffn_norm -> silu(W1 x) * (W3 x) -> in-kernel hand-off of h -> x' + W2 h
                FfnArgs a{};
                a.w1 = L.w13; a.w3 = L.w13 + (size_t)DH * DD; a.w2 = L.w2; a.x = xin; a.wo_part = e->d_wo_part; a.norm_w = L.n2; a.ada = L.ada;
                a.eps = d.dec_eps; a.x_out = xalt; a.xprime_out = tap_i >= 0 ? e->d_xprime : nullptr;
                a.gh = e->d_gh; a.epoch = e->fuse_epoch; a.err = e->d_fuse_err; a.spin_limit = 500000ull;

                a.tl = (l == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl + TL_STRIDE * 1024 : nullptr;
                if (merged_here) {
                    // k_ffn_attn12: this FFN block and the NEXT layer's attention block in one launch; x'' goes over in granules
                    DecLayer &N = e->dec[l + 1];
                    DecFuseArgs b{};
                    b.wqkv = N.wqkv; b.wo = N.wo; b.x = xalt; b.norm_w = N.n1; b.eps = d.dec_eps; b.inv_freq = e->dec_inv_freq;
                    b.kring = N.kring; b.vring = N.vring; b.kv_cap = e->dec_ring_cap; b.pos = kv_pos; b.window = d.dec_window; b.scale = scale;
                    b.gq = e->d_gq; b.gp = e->d_gp; b.wo_part = e->d_wo_part;
                    if (++e->fuse_epoch == 0) e->fuse_epoch = 1;
                    b.epoch = e->fuse_epoch; b.split_keys = f_split; b.nsplit = f_ns;
                    b.err = e->d_fuse_err; b.spin_limit = 500000ull;
                    b.tl = (l + 1 == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl : nullptr;
                    if (e->pf_units > 0) {
                        b.pf.w = reinterpret_cast<const unsigned char *>(N.w13); b.pf.row_bytes = 2 * DD; b.pf.rows_m = DH;
                        b.pf.units = std::min(e->pf_units, 72 * 6); b.pf.member_units = 0; b.pf.when = 3;
                    }
                    if (long12) hipLaunchKernelGGL(k_ffn_attn12<true>, dim3(256), dim3(FFN_THREADS), FA12_LDS_BYTES, s, a, b, e->d_gx);
                    else hipLaunchKernelGGL(k_ffn_attn12<false>, dim3(256), dim3(FFN_THREADS), FA12_LDS_BYTES, s, a, b, e->d_gx);
                    attn_done = true;
                    prof_mark(e, PK_W2);          // (the per-kernel table lists the merged launches in the slot the fused FFN path leaves empty)
                } else {
                    hipLaunchKernelGGL(k_ffn_fused, dim3(256), dim3(FFN_THREADS), FFN_LDS_BYTES, s, a);
                    prof_mark(e, PK_SWIGLU);
                }
                tap(2 * l + 1, e->d_xprime);
                std::swap(xin, xalt);
                continue;
            }
            if (!(e->skip_kinds & (1u << PK_SWIGLU))) {
                // x += sum of the wo partials -> ffn_norm * (1 + ada) -> silu(W1 x) * (W3 x)
                W13xArgs a{};
                a.w1 = e->sim_on ? L.w13_s : L.w13; a.w3 = a.w1 + (size_t)DH * DD; a.x = xin; a.wo_part = e->d_wo_part; a.norm_w = L.n2; a.ada = L.ada;
                a.eps = d.dec_eps; a.x_out = xalt; a.h = e->dh;
                a.tl = (l == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl + TL_STRIDE * 1024 : nullptr;
                if (e->use_fp8) {
                    a.w1 = reinterpret_cast<const uint16_t *>(L.w138); a.w3 = reinterpret_cast<const uint16_t *>(L.w138 + (size_t)DH * DD);
                    a.s1 = L.s13; a.s3 = L.s13 + DH;
                    hipLaunchKernelGGL(k_gemv_w13x<true>, dim3(256), dim3(W13X_THREADS), W13X_LDS_BYTES, s, a);
                } else {
                    hipLaunchKernelGGL(k_gemv_w13x<false>, dim3(256), dim3(W13X_THREADS), W13X_LDS_BYTES, s, a);
                }
                prof_mark(e, PK_SWIGLU);
            }
            tap(2 * l + 1, xalt);         // x' = x + attention block, written by k_gemv_w13x's block 0
            if (!(e->skip_kinds & (1u << PK_W2))) {
                {
                    W2xArgs a{};
                    a.w2 = e->sim_on ? L.w2_s : L.w2; a.h = e->dh; a.x = xalt;        // x' += h . W2^T, in place (one wave per row)
                    a.tl = (l == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl + 2 * TL_STRIDE * 1024 : nullptr;
                    if (e->use_fp8 && shape12_f8 && l + 1 < d.dec_layers) {
                        a.w2 = reinterpret_cast<const uint16_t *>(L.w28); a.s2 = L.s2;
                        DecLayer &N = e->dec[l + 1];
                        DecFuseArgs b{};
                        b.wqkv = reinterpret_cast<const uint16_t *>(N.wqkv8); b.wo = reinterpret_cast<const uint16_t *>(N.wo8); b.sqkv = N.sqkv; b.so = N.so;
                        b.x = xalt; b.norm_w = N.n1; b.eps = d.dec_eps; b.inv_freq = e->dec_inv_freq;
                        b.kring = N.kring; b.vring = N.vring; b.kv_cap = e->dec_ring_cap; b.pos = kv_pos; b.window = d.dec_window; b.scale = scale;
                        b.gq = e->d_gq; b.gp = e->d_gp; b.wo_part = e->d_wo_part;
                        const unsigned x_epoch = e->fuse_epoch;
                        if (++e->fuse_epoch == 0) e->fuse_epoch = 1;
                        b.epoch = e->fuse_epoch; b.split_keys = f_split; b.nsplit = f_ns;
                        b.err = e->d_fuse_err; b.spin_limit = 500000ull;
                        b.tl = (l + 1 == tl_layer && e->d_fuse_tl) ? e->d_fuse_tl : nullptr;
                        if (e->pf_units > 0) {
                            b.pf.w = reinterpret_cast<const unsigned char *>(N.w138); b.pf.row_bytes = DD; b.pf.rows_m = DH;
                            b.pf.units = std::min(e->pf_units, 72 * 3); b.pf.member_units = 0; b.pf.when = 3;
                        }
                        hipLaunchKernelGGL(k_w2x_attn12, dim3(256), dim3(W2X_THREADS), DA12_LDS_BYTES, s, a, b, e->d_gx, x_epoch);
                        attn_done = true;
                    } else if (e->use_fp8) {
                        a.w2 = reinterpret_cast<const uint16_t *>(L.w28); a.s2 = L.s2;
                        hipLaunchKernelGGL(k_gemv_w2x<true>, dim3(256), dim3(W2X_THREADS), W2X_LDS_BYTES, s, a);
                    } else {
                        hipLaunchKernelGGL(k_gemv_w2x<false>, dim3(256), dim3(W2X_THREADS), W2X_LDS_BYTES, s, a);
                    }
                }
                prof_mark(e, PK_W2);
            }
            std::swap(xin, xalt);
            continue;
        }
        if (!(e->skip_kinds & (1u << PK_QKV)))
        {   // RMSNorm -> merged QKV GEMV -> RoPE -> KV append   (voxtral_decoder.c:656-665)
            GemvArgs a{};
            a.W = L.wqkv; a.x = e->dx; a.norm_w = L.n1; a.ada = nullptr; a.eps = d.dec_eps; a.y = e->dq;
            a.N = DQ + 2 * DKV; a.K = DD; a.q_rows = DQ; a.k_rows = DKV; a.head_dim = HD; a.rope = e->dec_rope;
            a.kcache = L.kring; a.vcache = L.vring; a.kv_cap = e->dec_ring_cap; a.kv_dim = DKV; a.st = e->d_st;
            a.inv_freq = e->dec_inv_freq; a.adapter = e->adapter; a.tok_emb = e->tok_emb; a.x_out = e->dx;
            a.pos_host = kv_pos;
            const bool f8 = fast && e->use_fp8;
            if (f8) { a.W = reinterpret_cast<const uint16_t *>(L.wqkv8); a.wscale = L.sqkv; }
            if (!fast) launch_gemv<PRO_RMS, EPI_QKV, 4>(e, a);
            else if (f8 && l == 0 && build_embed) launch_gemv3<PRO_EMBED_RMS, EPI_QKV, 2, 3, 1, 3, true>(e, a);
            else if (f8) launch_gemv3<PRO_RMS, EPI_QKV, 2, 3, 1, 3, true>(e, a);
            else if (l == 0 && build_embed) launch_gemv3<PRO_EMBED_RMS, EPI_QKV, 2, 6, 1, 3>(e, a);
            else launch_gemv3<PRO_RMS, EPI_QKV, 2, 6, 1, 3>(e, a);
            prof_mark(e, PK_QKV);
        }
        tap(2 * l, e->dx);
        if (!(e->skip_kinds & (1u << PK_ATTN)))
        {   // attention over the KV window (voxtral_decoder.c:667-673)
            AttnArgs a{};
            a.out = e->dattn; a.ldo = DQ; a.q = e->dq; a.ldq = DQ; a.n_q = 1; a.qpos0 = kv_pos;
            a.posB0 = INT_MAX; a.last_key = kv_pos; a.kA = L.kring; a.vA = L.vring; a.capA = e->dec_ring_cap; a.ldA = DKV;
            a.n_heads = d.dec_heads; a.n_kv_heads = d.dec_kv_heads; a.scale = scale; a.window = d.dec_window;
            // fast path: the position comes from the host (no dependent scalar load in front of the
            // K/V reads); the generic kernels of the other geometries keep reading DecState
            a.st = fast ? nullptr : e->d_st; a.split_keys = split_keys; a.part_o = e->dpart_o; a.part_ml = e->dpart_ml;
            a.force_partials = fuse_combine ? 1 : 0;
            if (e->use_dpp)
                hipLaunchKernelGGL((k_attn_dec<128, 4, true>), dim3(d.dec_kv_heads, nsplit, 1), dim3(256), 0, s, a, nsplit);
            else
                hipLaunchKernelGGL((k_attn_dec<128, 4, false>), dim3(d.dec_kv_heads, nsplit, 1), dim3(256), 0, s, a, nsplit);
            prof_mark(e, PK_ATTN);
            if (nsplit > 1 && !fuse_combine) {
                hipLaunchKernelGGL((k_attn_combine<128>), dim3(d.dec_heads, 1), dim3(128), 0, s, e->dattn, DQ,
                                   (const float *)e->dpart_o, (const float *)e->dpart_ml, d.dec_heads, nsplit);
                prof_mark(e, PK_COMBINE);
            }
        }
        if (!(e->skip_kinds & (1u << PK_WO)))
        {   // x += attn.Wo^T   (fast path: the split-K partials are merged in the prologue)
            GemvArgs a{};
            a.W = L.wo; a.x = e->dattn; a.y = e->dx; a.N = DD; a.K = DQ;
            a.part_o = e->dpart_o; a.part_ml = e->dpart_ml; a.nsplit = nsplit; a.attn_hd = HD;
            const bool f8 = fast && e->use_fp8;
            if (f8) { a.W = reinterpret_cast<const uint16_t *>(L.wo8); a.wscale = L.so; }
            if (!fast) launch_gemv<PRO_NONE, EPI_RESID, 2>(e, a);
            else if (f8 && fuse_combine) launch_gemv3<PRO_ATTN, EPI_RESID, 3, 4, 1, 1, true>(e, a);
            else if (f8) launch_gemv3<PRO_NONE, EPI_RESID, 3, 4, 1, 1, true>(e, a);
            else if (fuse_combine) launch_gemv3<PRO_ATTN, EPI_RESID, 3, 8, 1, 1>(e, a);
            else launch_gemv3<PRO_NONE, EPI_RESID, 3, 8, 1, 1>(e, a);
            prof_mark(e, PK_WO);
        }
        tap(2 * l + 1, e->dx);
        if (!(e->skip_kinds & (1u << PK_SWIGLU)))
        {   // RMSNorm * (1+ada) -> silu(W1 x) * (W3 x)
            GemvArgs a{};
            a.W = L.w13; a.W2 = L.w13 + (size_t)DH * DD; a.x = e->dx; a.norm_w = L.n2; a.ada = L.ada; a.eps = d.dec_eps;
            a.y = e->dh; a.N = DH; a.K = DD;
            const bool f8 = fast && e->use_fp8;
            if (f8) {
                a.W = reinterpret_cast<const uint16_t *>(L.w138); a.W2 = reinterpret_cast<const uint16_t *>(L.w138 + (size_t)DH * DD);
                a.wscale = L.s13; a.wscale2 = L.s13 + DH;
            }
            if (!fast) launch_gemv<PRO_RMS, EPI_SWIGLU, 2>(e, a);
            else if (f8) launch_gemv3<PRO_RMS, EPI_SWIGLU, 3, 3, 1, 3, true>(e, a);
            else launch_gemv3<PRO_RMS, EPI_SWIGLU, 3, 6, 1, 3>(e, a);
            prof_mark(e, PK_SWIGLU);
        }
        if (!(e->skip_kinds & (1u << PK_W2)))
        {   // x += h.W2^T
            GemvArgs a{};
            a.W = L.w2; a.x = e->dh; a.y = e->dx; a.N = DD; a.K = DH;
            const bool f8 = fast && e->use_fp8;
            if (f8) { a.W = reinterpret_cast<const uint16_t *>(L.w28); a.wscale = L.s2; }
            if (!fast) launch_gemv<PRO_NONE, EPI_RESID, 2>(e, a);
            else if (f8) launch_gemv3<PRO_NONE, EPI_RESID, 1, 9, 1, 3, true>(e, a);
            else launch_gemv3<PRO_NONE, EPI_RESID, 3, 9, 2, 2>(e, a);
            prof_mark(e, PK_W2);
        }
    }
    tap(2 * d.dec_layers, xin);
    enqueue_logits_tail(e, xin, logits_dst, eos, advance, fast);
    LAUNCH_CHECK("decode step launches");
    return 0;
}

static int set_state(vox_hip_engine *e, int pos, int token, int64_t adapter_phys_row) {
    // the source is a pinned slot (two, alternating), so nothing has to wait here: the copy is in stream order in front of the steps
    // that read the state, and the host goes on enqueueing them (round 6; before: a stack variable and a host wait per run)
    if (e->pin_st_inflight >= 2) HC(esync(e));
    DecState &st = e->pin->st_in[e->pin_st_next];
    e->pin_st_next ^= 1; e->pin_st_inflight++;
    st.pos = pos; st.token = token; st.n_out = 0; st.stop = 0; st.adapter_row = adapter_phys_row;
    HC(hipMemcpyAsync(e->d_st, &st, sizeof st, hipMemcpyHostToDevice, e->stream));
    return 0;
}

// After a synchronisation: did a hand-off of the fused decode kernel time out since the last check?  If so the results of
// everything enqueued since are void: switch to the launch-per-GEMV chain (loudly) and tell the caller to redo.  The switch
// is a suspension, not a verdict: a time-out means the 256 workgroups were not co-resident for a moment (another process or
// stream held CUs), so after FUSE_REARM_STEPS clean steps on the chain (doubling with every further failure, capped) the
// fused kernel is tried again - one transient contention event used to cost 0.2 ms per token for the engine's lifetime.
// VOX_HIP_DISABLE=rearm keeps the old sticky behaviour (A/B).
constexpr long FUSE_REARM_STEPS = 256;
static int fused_failed_words(vox_hip_engine *e, const unsigned *w);
static int fused_failed(vox_hip_engine *e) {
    if (!e->use_fused) return 0;
    unsigned w[16] = {0};
    if (hipMemcpy(w, e->d_fuse_err, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return fused_failed_words(e, w);
}
// (w = the 16 error words as read back after the host wait - by the caller's own asynchronous copy on the hot path)
static int fused_failed_words(vox_hip_engine *e, const unsigned *w) {
    if (!e->use_fused) return 0;
    e->spin_hole_max = std::max(e->spin_hole_max, w[8]); e->spin_holes += w[9];
    if (w[8] | w[9]) (void)hipMemset(e->d_fuse_err + 8, 0, 2 * sizeof(unsigned));
    const unsigned err = w[0];
    if (!err) return 0;
    e->use_fused = false; e->fuse_failures++;
    static const bool no_rearm = vox_disabled("rearm");
    e->fuse_rearm = no_rearm ? 0 : FUSE_REARM_STEPS << std::min(e->fuse_failures - 1, 6);
    // (ticks of the 100 MHz wall clock -> us)
    fprintf(stderr, "vox_hip: ERROR fused decode kernel timed out in hand-off %u (its 256 workgroups were not co-resident?) in the batch "
                    "of decoder steps from position %d on (hand-off epoch counter %u after it): workgroup %u (XCD %u, thread %u) waited for tag %u, "
                    "%.1f us of active waiting, longest gap between two polls %.1f us; holes > 1 ms seen by any spin so far: %llu (longest %.1f us); "
                    "repeating the work on the launch-per-GEMV chain%s\n",
            err, e->dec_pos, e->fuse_epoch, w[1], w[2] & 15u, w[7], w[6], w[4] / 100.0, w[3] / 100.0,
            (unsigned long long)e->spin_holes, e->spin_hole_max / 100.0,
            e->fuse_rearm ? " and staying there for a while before the fused kernel is tried again" : " and staying there");
    (void)hipMemset(e->d_fuse_err, 0, 8 * sizeof(unsigned));
    return 1;
}
// `steps` decode steps completed cleanly: count down a suspension of the fused kernel.
static void fused_rearm_tick(vox_hip_engine *e, int steps) {
    if (!e->fused_ok || e->use_fused || e->fuse_rearm <= 0 || steps <= 0) return;
    e->fuse_rearm -= steps;
    if (e->fuse_rearm <= 0) {
        e->fuse_rearm = 0; e->use_fused = true;
        fprintf(stderr, "vox_hip: fused decode kernel re-armed after %d time-out(s)\n", e->fuse_failures);
    }
}
// Debug taps of the decoder's residual stream (parity tests at the depth of the stack, not only at its logits): the decode
// steps that run at the listed KV positions copy x at the start of every layer, x after every attention block and x after
// the last layer - [2 L + 1][dec_dim] per position, in that order - into a device buffer, in stream order (no effect on the
// kernels).  vox_hip_debug_tap_read fetches [n][2 L + 1][dec_dim] and ends the tapping.  n <= 16.
extern "C" int vox_hip_debug_tap_config(vox_hip_engine_t *e, const int *positions, int n) {
    if (!e || n < 0 || n > 16) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    if (e->d_taps) { hipFree(e->d_taps); e->d_taps = nullptr; }
    e->tap_pos.clear();
    if (n == 0) return 0;
    const size_t elems = (size_t)n * (2 * e->d.dec_layers + 1) * e->d.dec_dim;
    HC(hipMalloc((void **)&e->d_taps, elems * 4));
    HC(hipMemset(e->d_taps, 0, elems * 4));
    e->tap_pos.assign(positions, positions + n);
    return 0;
}
extern "C" int vox_hip_debug_tap_read(vox_hip_engine_t *e, float *out) {
    if (!e || !e->d_taps || !out) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    HC(hipMemcpy(out, e->d_taps, e->tap_pos.size() * (2 * e->d.dec_layers + 1) * e->d.dec_dim * 4, hipMemcpyDeviceToHost));
    hipFree(e->d_taps); e->d_taps = nullptr; e->tap_pos.clear();
    return 0;
}

// fuse_failures so far / is the fused kernel live right now / steps left of a suspension (tests, bench)
extern "C" int vox_hip_fuse_stats(const vox_hip_engine_t *e, int *failures, int *armed, long *rearm_in) {
    if (!e) return -1;
    if (failures) *failures = e->fuse_failures;
    if (armed) *armed = e->use_fused ? 1 : 0;
    if (rearm_in) *rearm_in = e->fuse_rearm;
    return e->fused_ok ? 0 : 1;
}
// Diagnosis: how many holes (> 1 ms between two consecutive polls of a bounded spin: the queue was switched out, vox_decfuse.h
// "bounded spins") the decode launches have seen so far, and the longest one in microseconds.
extern "C" int vox_hip_spin_holes(vox_hip_engine_t *e, unsigned long long *count, double *longest_us) {
    if (!e || !e->d_fuse_err) return -1;
    unsigned w[2] = {0, 0};
    HC(hipSetDevice(e->device));
    HC(esync(e));
    HC(hipMemcpy(w, e->d_fuse_err + 8, sizeof w, hipMemcpyDeviceToHost));
    if (count) *count = e->spin_holes + w[1];
    if (longest_us) *longest_us = std::max(e->spin_hole_max, w[0]) / 100.0;
    return 0;
}
// fp8 mode's prefill on the fp8 MFMA (k_rowsgemm_f8): *fallbacks = how often a prefill had to be repeated on the bf16 matrices because an
// activation exceeded the e4m3 range after the prescale (the first time switches the fp8 MFMA prefill off for the engine); *clamped = the
// device counter right now (read and cleared: the kernel-level entry vox_hip_linear_bf16(impl 6) counts into it too).
extern "C" int vox_hip_fp8_prefill_stats(vox_hip_engine_t *e, int *fallbacks, unsigned *clamped) {
    if (!e || !e->d_f8_clamped) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    unsigned c = 0;
    HC(hipMemcpy(&c, e->d_f8_clamped, sizeof c, hipMemcpyDeviceToHost));
    if (c) HC(hipMemset(e->d_f8_clamped, 0, sizeof c));
    if (fallbacks) *fallbacks = e->fp8_prefill_fallbacks;
    if (clamped) *clamped = c;
    return 0;
}
// The encoder stack kernel (VOX_PATH_ENC_STACK): launches so far, hand-off time-outs so far, 1 if it is live now (0: suspended after a
// time-out, or not available on this engine).  Returns 0, -1 on error.
extern "C" int vox_hip_enc_stack_stats(const vox_hip_engine_t *e, long *launches, int *failures, int *armed) {
    if (!e) return -1;
    if (launches) *launches = (long)e->enc_stack_launches;
    if (failures) *failures = e->enc_stack_failures;
    if (armed) *armed = (e->enc_stack_ok && e->enc_stack_rearm <= 0) ? 1 : 0;
    return 0;
}
// Test hook: the next check after a chunk behaves as if a hand-off of the stack kernel had timed out (the chunk is repeated on the
// launch-per-GEMM path, the suspension runs).
extern "C" int vox_hip_debug_inject_enc_stack_timeout(vox_hip_engine_t *e) {
    if (!e || !e->enc_stack_ok) return -1;
    e->enc_stack_inject = true;
    return 0;
}
extern "C" int vox_hip_debug_set_handoff_epoch(vox_hip_engine_t *e, unsigned epoch, unsigned *old) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    if (old) *old = e->fuse_epoch;
    e->fuse_epoch = epoch;
    return 0;
}
// Test hook: make the next check after a synchronisation behave as if a hand-off had timed out (the batch is repeated on
// the chain, the suspension / re-arm logic runs).  No effect on engines without the fused kernel.
extern "C" int vox_hip_debug_inject_fuse_timeout(vox_hip_engine_t *e) {
    if (!e || !e->use_fused) return -1;
    HC(hipSetDevice(e->device));
    HC(esync(e));
    const unsigned code = 99u;
    HC(hipMemcpy(e->d_fuse_err, &code, sizeof code, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int vox_hip_decoder_step(vox_hip_engine_t *e, const float *embed, float *logits) {
    if (!e || !embed) return -1;
    HC(hipSetDevice(e->device));
    int tok = -1;
    for (int attempt = 0; attempt < 2; attempt++) {
        HC(hipMemcpyAsync(e->dx, embed, (size_t)e->d.dec_dim * 4, hipMemcpyHostToDevice, e->stream));
        if (set_state(e, e->dec_pos, 0, 0)) return -1;
        if (enqueue_step(e, e->dec_pos, false, e->dlogits, -1, 1)) return -1;
        HC(hipMemcpyAsync(&tok, e->d_tokens, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        if (logits) HC(hipMemcpyAsync(logits, e->dlogits, (size_t)e->d.vocab * 4, hipMemcpyDeviceToHost, e->stream));
        HC(esync(e));
        if (!fused_failed(e)) break;
    }
    fused_rearm_tick(e, 1);
    e->dec_pos += 1;
    return tok;
}

// fp8 mode: did the prefill that has just been waited for clamp an activation (k_rowsgemm_f8's fixed prescale covers |x| <= 112)?  Then its
// K/V and logits are not to be trusted: the fp8 MFMA prefill is switched off for this engine (VOX_PATH_FP8_MFMA disappears from
// vox_hip_active_paths) and the caller repeats the pass on the bf16 matrices.
static int f8_prefill_clamped(vox_hip_engine *e) {
    if (!e->use_fp8 || e->fp8_prefill_bf16 || !e->d_f8_clamped) return 0;
    unsigned c = 0;
    if (hipMemcpy(&c, e->d_f8_clamped, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (!c) return 0;
    (void)hipMemset(e->d_f8_clamped, 0, sizeof c);
    e->fp8_prefill_bf16 = true; e->fp8_prefill_fallbacks++;
    fprintf(stderr, "vox_hip: WARNING fp8 prefill: %u activation group(s) beyond the e4m3 range after the prescale (|x| > 112); repeating the prefill on the "
                    "bf16 matrices and keeping it there for this engine\n", c);
    return 1;
}
static int prefill_stream_once(vox_hip_engine_t *e, int64_t first_row, int n_prompt, int bos, int pad, float *logits);
extern "C" int vox_hip_decoder_prefill_stream(vox_hip_engine_t *e, int64_t first_row, int n_prompt, int bos, int pad, float *logits) {
    if (!e || n_prompt < 1) return -1;
    const int pos0 = e->dec_pos; const int64_t cons0 = e->adapter_consumed;
    int tok = prefill_stream_once(e, first_row, n_prompt, bos, pad, logits);
    if (tok >= 0 && f8_prefill_clamped(e)) {
        e->dec_pos = pos0; e->adapter_consumed = cons0;
        tok = prefill_stream_once(e, first_row, n_prompt, bos, pad, logits);
    }
    return tok;
}
static int prefill_stream_once(vox_hip_engine_t *e, int64_t first_row, int n_prompt, int bos, int pad, float *logits) {
    HC(hipSetDevice(e->device));
    if (first_row < e->adapter_row0 || first_row + n_prompt > e->adapter_total) { g_err = "prefill_stream: adapter rows not resident"; return -1; }
    const int DD = e->d.dec_dim;
    hipStream_t s = e->stream;
    if (apply_row_fences(e, first_row + n_prompt - 1)) return -1;
    HC(hipEventRecord(e->ev0, s));
    if (ensure(e, e->sx, (size_t)n_prompt * DD * 4)) return -1;
    float *x = (float *)e->sx.p;
    const float *arow = e->adapter + (size_t)(first_row - e->adapter_row0) * DD;
    hipLaunchKernelGGL(k_embed_prompt, dim3(grid1d((size_t)n_prompt * DD)), dim3(256), 0, s, x, arow,
                       (const uint16_t *)e->tok_emb, n_prompt, DD, bos, pad);
    // Round 5: the LAST prompt row goes through the rows pass too (its k_rowsgemm launches cost the same for 39 rows as for 38, and
    // the pass computes the last layer's FFN block anyway), and only the logits tail runs on its row of the stack's output - instead
    // of a whole decode step for it (1.25 ms of the 4.3 ms this call took).  Same arithmetic up to summation order; the reference does
    // prefill(n - 1) + forward(1) (voxtral.c:1005-1012).  Not with a debug tap on that position (the taps live in the step), not
    // beyond one k_rowsgemm chunk.  VOX_HIP_DISABLE=prefill_tail is the old sequence.
    bool tapped = false;
    for (size_t i = 0; i < e->tap_pos.size(); i++) if (e->tap_pos[i] == e->dec_pos + n_prompt - 1) tapped = true;
    const bool fast_geom = e->use_fast && DD == 3072 && e->dec_qd == 4096 && e->dec_kvd == 1024 && e->d.dec_hidden == 9216;
    if (n_prompt > 1 && n_prompt <= 128 && rowsgemm_ok(e, n_prompt, dec_cfg(e)) && !tapped && !e->sim_on && !e->skip_kinds &&
        !vox_disabled("prefill_tail")) {
        if (decoder_prefill_dev(e, x, n_prompt)) return -1;
        int tok = -1;
        if (set_state(e, e->dec_pos - 1, 0, 0)) return -1;
        enqueue_logits_tail(e, x + (size_t)(n_prompt - 1) * DD, e->dlogits, -1, 1, fast_geom);
        HC(hipMemcpyAsync(&tok, e->d_tokens, sizeof(int), hipMemcpyDeviceToHost, s));
        if (logits) HC(hipMemcpyAsync(logits, e->dlogits, (size_t)e->d.vocab * 4, hipMemcpyDeviceToHost, s));
        HC(hipEventRecord(e->ev1, s));
        HC(esync(e));
        float ms = 0.f; hipEventElapsedTime(&ms, e->ev0, e->ev1);
        e->timing.prefill_ms += ms;
        e->adapter_consumed = std::max(e->adapter_consumed, first_row + n_prompt);
        return tok;
    }
    if (n_prompt > 1 && decoder_prefill_dev(e, x, n_prompt - 1)) return -1;
    int tok = -1;
    for (int attempt = 0; attempt < 2; attempt++) {
        // the last prompt row is the first decode step's input (voxtral.c:1005-1012)
        HC(hipMemcpyAsync(e->dx, x + (size_t)(n_prompt - 1) * DD, (size_t)DD * 4, hipMemcpyDeviceToDevice, s));
        if (set_state(e, e->dec_pos, 0, 0)) return -1;
        if (enqueue_step(e, e->dec_pos, false, e->dlogits, -1, 1)) return -1;
        HC(hipMemcpyAsync(&tok, e->d_tokens, sizeof(int), hipMemcpyDeviceToHost, s));
        if (logits) HC(hipMemcpyAsync(logits, e->dlogits, (size_t)e->d.vocab * 4, hipMemcpyDeviceToHost, s));
        HC(hipEventRecord(e->ev1, s));
        HC(esync(e));
        if (!fused_failed(e)) break;
    }
    float ms = 0.f; hipEventElapsedTime(&ms, e->ev0, e->ev1);
    e->timing.prefill_ms += ms;
    e->dec_pos += 1;
    e->adapter_consumed = std::max(e->adapter_consumed, first_row + n_prompt);
    return tok;
}

extern "C" int vox_hip_decoder_run(vox_hip_engine_t *e, int64_t first_row, int n_steps, int prev_token, int eos_token,
                                   int *tokens_out, float *logits_out) {
    if (!e || n_steps <= 0 || !tokens_out) return -1;
    HC(hipSetDevice(e->device));
    if (first_row < e->adapter_row0 || first_row + n_steps > e->adapter_total) { g_err = "decoder_run: adapter rows not resident"; return -1; }
    hipStream_t s = e->stream;
    const size_t V = e->d.vocab;
    int done = 0;
    // the wait for the shard that holds the first row is another GPU's encoder time, not decode time: in front of ev0
    // (as prefill_stream does); waits for later shards inside the run overlap with decoding and stay where they are
    if (!e->row_fences.empty() && apply_row_fences(e, first_row)) return -1;
    float ms_total = 0.f;
    while (done < n_steps) {
        const int batch = std::min(n_steps - done, logits_out ? 64 : MAX_RUN_STEPS);
        HC(hipEventRecord(e->ev0, s));
        if (set_state(e, e->dec_pos, prev_token, first_row + done - e->adapter_row0)) return -1;
        float *lg = e->dlogits;
        if (logits_out) {
            if (ensure(e, e->stmp_out, (size_t)batch * V * 4)) return -1;
            lg = (float *)e->stmp_out.p;
        }
        for (int i = 0; i < batch; i++) {
            // a shard's adapter rows are waited for (on the stream) right in front of the first step that reads them
            if (!e->row_fences.empty() && apply_row_fences(e, first_row + done + i)) return -1;
            if (enqueue_step(e, e->dec_pos + i, true, logits_out ? lg + (size_t)i * V : lg, eos_token, 1)) return -1;
        }
        // ONE host wait per batch: the timing event, the state, the fused kernels' error words and the token ids all come back through
        // pinned memory in stream order behind the steps (round 6; before: a wait + three blocking copies + a second wait per run, ~0.15 ms
        // of a streaming feed)
        HC(hipEventRecord(e->ev1, s));
        HC(hipMemcpyAsync(&e->pin->st_out, e->d_st, sizeof(DecState), hipMemcpyDeviceToHost, s));
        const bool fused_live = e->use_fused;
        if (fused_live) HC(hipMemcpyAsync(e->pin->fuse_err, e->d_fuse_err, sizeof e->pin->fuse_err, hipMemcpyDeviceToHost, s));
        HC(hipMemcpyAsync(e->pin->tokens, e->d_tokens, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, s));
        HC(esync(e));
        if (fused_live && fused_failed_words(e, e->pin->fuse_err)) continue;          // the batch's results are void: run it again (now on the chain)
        { float ms = 0.f; hipEventElapsedTime(&ms, e->ev0, e->ev1); ms_total += ms; }
        const DecState st = e->pin->st_out;
        const int got = st.n_out;
        if (got > 0) memcpy(tokens_out + done, e->pin->tokens, (size_t)got * sizeof(int));
        if (logits_out && got > 0)
            HC(hipMemcpy(logits_out + (size_t)done * V, lg, (size_t)got * V * 4, hipMemcpyDeviceToHost));
        e->dec_pos += got;
        done += got;
        fused_rearm_tick(e, got);
        if (got > 0) prev_token = tokens_out[done - 1];
        if (st.stop || got < batch) break;
    }
    e->timing.decode_ms += ms_total;
    e->timing.decode_steps += done;
    e->adapter_consumed = std::max(e->adapter_consumed, first_row + done);
    return done;
}

// ------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------
extern "C" void vox_hip_reset_encoder(vox_hip_engine_t *e) {
    if (!e) return;
    hipSetDevice(e->device);
    (void)drain_fences(e);
    esync(e);
    e->enc_pos = 0; e->mel_q = 0; e->c0_carry = 0; e->enc_res = 0;
    // zero the causal-history rows (start-of-sequence padding)
    if (e->conv_in0.p) hipMemsetAsync(e->conv_in0.p, 0, (size_t)2 * e->d.mel_bins * 4, e->stream);
    if (e->conv_in1.p) hipMemsetAsync(e->conv_in1.p, 0, (size_t)2 * e->d.enc_dim * 4, e->stream);
    // k_attn_small's per-head arrival counters are reset by the last arriver only: after an aborted or faulted launch they would stay non-zero
    if (e->d_attn_arrive) hipMemsetAsync(e->d_attn_arrive, 0, (size_t)std::max(1, e->d.enc_heads) * 4, e->stream);
    esync(e);
}
// The same without waiting: the host-side counters are reset at once, the history rows are zeroed in stream order
// (everything already enqueued still sees the old rows, everything enqueued later the new ones).
extern "C" void vox_hip_reset_encoder_async(vox_hip_engine_t *e) {
    if (!e) return;
    hipSetDevice(e->device);
    (void)apply_enc_fences(e);
    e->enc_pos = 0; e->mel_q = 0; e->c0_carry = 0; e->enc_res = 0;
    if (e->conv_in0.p) hipMemsetAsync(e->conv_in0.p, 0, (size_t)2 * e->d.mel_bins * 4, e->stream);
    if (e->conv_in1.p) hipMemsetAsync(e->conv_in1.p, 0, (size_t)2 * e->d.enc_dim * 4, e->stream);
    if (e->d_attn_arrive) hipMemsetAsync(e->d_attn_arrive, 0, (size_t)std::max(1, e->d.enc_heads) * 4, e->stream);
}
// The engine's HIP stream (a hipStream_t) for callers that order their own work against it without a host wait
// (multi_gpu.py wraps it in torch.cuda.ExternalStream so that RCCL send / recv are stream-ordered with the shard kernels),
// and the number of host-side waits on that stream so far.
extern "C" void *vox_hip_stream_handle(vox_hip_engine_t *e) { return e ? (void *)e->stream : nullptr; }
extern "C" unsigned long long vox_hip_host_syncs(const vox_hip_engine_t *e) { return e ? e->n_host_syncs : 0; }
extern "C" void vox_hip_reset_decoder(vox_hip_engine_t *e) {
    if (!e) return;
    hipSetDevice(e->device);
    (void)drain_fences(e);
    esync(e);
    e->dec_pos = 0;
    e->adapter_total = 0; e->adapter_row0 = 0; e->adapter_consumed = 0;
}
extern "C" void vox_hip_reset_decoder_kv(vox_hip_engine_t *e) {
    if (!e) return;
    hipSetDevice(e->device);
    esync(e);
    e->dec_pos = 0;
}
extern "C" int vox_hip_decoder_kv_len(const vox_hip_engine_t *e) { return e ? e->dec_pos : 0; }
extern "C" int vox_hip_mel_queue_len(const vox_hip_engine_t *e) { return e ? e->mel_q : 0; }
extern "C" void vox_hip_sync(vox_hip_engine_t *e) { if (e) { hipSetDevice(e->device); (void)drain_fences(e); esync(e); } }
// 1 if the stream's encoder state sits on a token boundary (no conv0 frame waiting for its stride-2 partner, no encoder rows
// waiting for 4x alignment): the state a sharded chunk may start from (vox_multi.c)
extern "C" int vox_hip_encoder_aligned(const vox_hip_engine_t *e) { return e ? (e->c0_carry == 0 && e->enc_res == 0) : 0; }
extern "C" int vox_hip_encoder_pos(const vox_hip_engine_t *e) { return e ? e->enc_pos : 0; }
extern "C" int vox_hip_pending_fences(const vox_hip_engine_t *e) { return e ? (int)(e->row_fences.size() + e->enc_fences.size()) : 0; }
extern "C" void vox_hip_get_timing(const vox_hip_engine_t *e, vox_hip_timing_t *t) { if (e && t) *t = e->timing; }
extern "C" void vox_hip_reset_timing(vox_hip_engine_t *e) { if (e) e->timing = vox_hip_timing_t{}; }
// Encoder time spent outside vox_hip_stream_encode (a chunk sharded over several engines, host/vox_multi.c): the host
// measures it around a synchronisation of the stream engine and books it here so that vox_hip_get_timing stays the
// one place a caller reads phase times from.
extern "C" void vox_hip_add_encode_ms(vox_hip_engine_t *e, double ms) { if (e && ms > 0) e->timing.encode_ms += ms; }

// ------------------------------------------------------------------------------------
// kernel-level test / bench surface
// ------------------------------------------------------------------------------------
// y[i] = hi + mid + lo of the bf16 planes p[0 .. 2][i] (test surface: reads a planes-writing epilogue's result back as f32)
__global__ void k_planes_sum(float *y, const uint16_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (__uint_as_float((uint32_t)p[i] << 16) + __uint_as_float((uint32_t)p[n + i] << 16)) + __uint_as_float((uint32_t)p[2 * n + i] << 16);
}
extern "C" int vox_hip_linear_bf16(vox_hip_engine_t *e, float *y, const float *x, const uint16_t *w, const float *bias,
                                   int M, int K, int N, int impl) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    float *dx = nullptr, *dy = nullptr, *db = nullptr; uint16_t *dw = nullptr;
    HC(hipMalloc((void **)&dx, (size_t)M * K * 4)); HC(hipMalloc((void **)&dy, (size_t)M * N * 4));
    const size_t wrows = impl == 9 ? 2 * (size_t)N : (size_t)N;
    HC(hipMalloc((void **)&dw, wrows * K * 2));
    HC(hipMemcpy(dx, x, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dw, w, wrows * K * 2, hipMemcpyHostToDevice));
    if (bias) { HC(hipMalloc((void **)&db, (size_t)N * 4)); HC(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice)); }
    if (impl == 4 || impl == 5) {
        // test surface of k_rowsgemm (vox_rowsgemm.h): 4 = activations as bf16 planes, 5 = f32 rows; partials added by k_splitk_reduce
        if (M > 128 || K % 64) { g_err = "vox_hip_linear_bf16: impl 4 / 5 need M <= 128 and K % 64 == 0"; return -1; }
        uint16_t *dp = nullptr; float *part = nullptr;
        HC(hipMalloc((void **)&dp, (size_t)3 * M * K * 2));
        HC(hipMalloc((void **)&part, rg_partial_bytes(M, N, K)));
        hipLaunchKernelGGL(k_split_planes, dim3(grid1d((size_t)M * K / 4)), dim3(256), 0, e->stream, dp, (size_t)M * K, (const float *)dx, K, M, K);
        const int S = impl == 4 ? launch_rowsgemm(e, dp, (size_t)M * K, nullptr, 0, M, dw, N, K, part)
                                : launch_rowsgemm(e, nullptr, 0, dx, K, M, dw, N, K, part);
        GemmArgs a{nullptr, 0, dw, dy, N, M, N, K, db, nullptr, 0, ACT_NONE, S, 0, part};
        hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, a);
        HC(esync(e));
        hipFree(dp); hipFree(part);
    } else if (impl == 6 || impl == 7) {
        // test surface of the fp8 mode's M > 1 GEMM: the weights are quantised on the device as vox_hip_quantize_decoder_fp8 does
        // (k_quant_fp8_rows: e4m3, one f32 scale per row); impl 6 = k_rowsgemm_f8 (fp8 MFMA, activations as two e4m3 terms);
        // impl 7 = the SAME quantised weights dequantised to f32 and multiplied in f32 (the reference of 6: isolates what the
        // activation split and the MFMA add to the weights' quantisation error)
        if (M > 64 || K % 64) { g_err = "vox_hip_linear_bf16: impl 6 / 7 need M <= 64 and K % 64 == 0"; return -1; }
        uint8_t *dq = nullptr; float *dsc = nullptr, *part = nullptr;
        HC(hipMalloc((void **)&dq, (size_t)N * K)); HC(hipMalloc((void **)&dsc, (size_t)N * 4));
        hipLaunchKernelGGL(k_quant_fp8_rows, dim3(N), dim3(256), 0, e->stream, (const uint16_t *)dw, dq, dsc, K);
        if (impl == 6) {
            HC(hipMalloc((void **)&part, rgf8_partial_bytes(M, N, K)));
            const int S = launch_rowsgemm_f8(e, dx, K, M, dq, dsc, N, K, part);
            if (S < 0) return -1;
            GemmArgs a{nullptr, 0, dw, dy, N, M, N, K, db, nullptr, 0, ACT_NONE, S, 0, part};
            hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, a);
        } else {
            hipLaunchKernelGGL(k_fp8_ref_gemm, dim3(N), dim3(64), 0, e->stream, dy, (const float *)dx, (const uint8_t *)dq, (const float *)dsc, (const float *)db, M, N, K);
        }
        HC(esync(e));
        hipFree(dq); hipFree(dsc); if (part) hipFree(part);
    } else if (impl == 8 || impl == 9) {
        // test surface of k_gemm_planes (vox_gemm_planes.h), both tile widths (128 x 128; 128 x 256 from 600 wide tiles on):
        // 8 = y = x W^T + bias;  9 = the SwiGLU launch: w holds [w1; w3] (2 N rows), y[M][N] = silu(x w1^T) * (x w3^T), read back
        // as the sum of the three bf16 planes the kernel writes (exact: the planes are the 3-term split of the f32 value)
        if (K % GP_K || (impl == 9 && N % 64)) { g_err = "vox_hip_linear_bf16: impl 8 / 9 need K % 32 == 0 (9: N % 64 == 0)"; return -1; }
        uint16_t *dp = nullptr, *dh = nullptr;
        HC(hipMalloc((void **)&dp, (size_t)3 * M * K * 2));
        hipLaunchKernelGGL(k_split_planes, dim3(grid1d((size_t)M * K / 4)), dim3(256), 0, e->stream, dp, (size_t)M * K, (const float *)dx, K, M, K);
        if (impl == 8) {
            if (launch_gemm_planes(e, dp, (size_t)M * K, K, dw, dy, N, M, N, K, db, nullptr, 0, ACT_NONE)) return -1;
        } else {
            HC(hipMalloc((void **)&dh, (size_t)3 * M * N * 2));
            GemmArgs ex{}; ex.Yp = dh; ex.yp_plane = (size_t)M * N;
            if (launch_gemm_planes(e, dp, (size_t)M * K, K, dw, nullptr, 0, M, N, K, nullptr, nullptr, 0, ACT_NONE, GP_EPI_SWIGLU, &ex)) return -1;
            hipLaunchKernelGGL(k_planes_sum, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, dy, (const uint16_t *)dh, (size_t)M * N);
        }
        HC(esync(e));
        hipFree(dp); if (dh) hipFree(dh);
    } else if (linear_dev(e, dy, N, dx, K, dw, db, M, K, N, ACT_NONE, nullptr, 0, impl)) return -1;
    HC(esync(e));
    HC(hipGetLastError());
    HC(hipMemcpy(y, dy, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy); hipFree(dw); if (db) hipFree(db);
    return 0;
}

extern "C" int vox_hip_causal_attention(vox_hip_engine_t *e, float *out, const float *q, const float *k, const float *v,
                                        int seq_q, int seq_k, int n_heads, int n_kv_heads, int head_dim, float scale,
                                        int window, int q_offset) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    const size_t qn = (size_t)seq_q * n_heads * head_dim, kn = (size_t)seq_k * n_kv_heads * head_dim;
    float *dq, *dk, *dv, *dout;
    HC(hipMalloc((void **)&dq, qn * 4)); HC(hipMalloc((void **)&dout, qn * 4));
    HC(hipMalloc((void **)&dk, kn * 4)); HC(hipMalloc((void **)&dv, kn * 4));
    HC(hipMemcpy(dq, q, qn * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dk, k, kn * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dv, v, kn * 4, hipMemcpyHostToDevice));
    AttnArgs a{};
    a.out = dout; a.ldo = n_heads * head_dim; a.q = dq; a.ldq = n_heads * head_dim; a.n_q = seq_q; a.qpos0 = q_offset;
    a.kB = dk; a.vB = dv; a.ldB = n_kv_heads * head_dim; a.posB0 = 0; a.last_key = seq_k - 1;
    a.kA = dk; a.vA = dv; a.capA = 1 << 30; a.ldA = n_kv_heads * head_dim;
    a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.scale = scale; a.window = window > 0 ? window : (1 << 30);
    a.st = nullptr; a.split_keys = DEC_SPLIT_KEYS;
    float *po = nullptr, *pml = nullptr;
    int rc = 0;
    if (head_dim == 64 && e->use_attn_mfma && n_heads == n_kv_heads) {
        hipLaunchKernelGGL(k_attn_enc_bf16, dim3((seq_q + 127) / 128, n_heads), dim3(256), 0, e->stream, a);
    } else if (head_dim == 64 && n_heads == n_kv_heads) {
        hipLaunchKernelGGL((k_attn_rows<64>), dim3((seq_q + 127) / 128, n_heads), dim3(128), 0, e->stream, a);
    } else if (head_dim == 128 && n_heads == 4 * n_kv_heads) {
        const int max_len = std::min(q_offset + seq_q, a.window);
        const int nsplit = std::max(1, (std::min(max_len, seq_k) + DEC_SPLIT_KEYS - 1) / DEC_SPLIT_KEYS);
        if (nsplit > 1) {
            HC(hipMalloc((void **)&po, (size_t)seq_q * n_heads * nsplit * head_dim * 4));
            HC(hipMalloc((void **)&pml, (size_t)seq_q * n_heads * nsplit * 2 * 4));
            a.part_o = po; a.part_ml = pml;
        }
        if (e->use_dpp)
            hipLaunchKernelGGL((k_attn_dec<128, 4, true>), dim3(n_kv_heads, nsplit, seq_q), dim3(256), 0, e->stream, a, nsplit);
        else
            hipLaunchKernelGGL((k_attn_dec<128, 4, false>), dim3(n_kv_heads, nsplit, seq_q), dim3(256), 0, e->stream, a, nsplit);
        if (nsplit > 1)
            hipLaunchKernelGGL((k_attn_combine<128>), dim3(n_heads, seq_q), dim3(128), 0, e->stream, dout, a.ldo,
                               (const float *)po, (const float *)pml, n_heads, nsplit);
    } else if (head_dim <= 256 && n_kv_heads > 0 && n_heads % n_kv_heads == 0) {
        // any other geometry: the generic kernel of the kernel-level API (vox_kernel_api.h)
        hipLaunchKernelGGL(k_attn_generic, dim3(seq_q, n_heads), dim3(64), 0, e->stream, dout, (const float *)dq, (const float *)dk,
                           (const float *)dv, seq_q, seq_k, n_heads, n_kv_heads, head_dim, scale, window, q_offset);
    } else {
        g_err = "vox_hip_causal_attention: unsupported head geometry"; rc = -1;
    }
    if (!rc) {
        HC(esync(e));
        HC(hipGetLastError());
        HC(hipMemcpy(out, dout, qn * 4, hipMemcpyDeviceToHost));
    }
    hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); if (po) hipFree(po); if (pml) hipFree(pml);
    return rc;
}

// Times `iters` full decode steps (all layers + logits + argmax) on the resident weights at a
// given KV length, without touching the stream state seen by the host API. Returns seconds/step.
extern "C" double vox_hip_time_decoder_step(vox_hip_engine_t *e, int iters, int kv_len) {
    if (!e || iters <= 0) return -1.0;
    if (hipSetDevice(e->device) != hipSuccess) return -1.0;
    const int saved_pos = e->dec_pos;
    int pos = std::max(0, kv_len - 1);
    hipMemsetAsync(e->dx, 0, (size_t)e->d.dec_dim * 4, e->stream);
    // make sure one adapter row exists for the embedding gather
    if (e->adapter_total - e->adapter_row0 < 1) {
        hipMemsetAsync(e->adapter, 0, (size_t)e->d.dec_dim * 4, e->stream);
    }
    if (set_state(e, pos, 1, 0)) return -1.0;
    {
        for (int i = 0; i < 3; i++) enqueue_step(e, pos, true, e->dlogits, -1, 0);   // warm-up
        hipEventRecord(e->ev0, e->stream);
        for (int i = 0; i < iters; i++) enqueue_step(e, pos, true, e->dlogits, -1, 0);
        hipEventRecord(e->ev1, e->stream);
    }
    esync(e);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e->ev0, e->ev1);
    e->dec_pos = saved_pos;
    if (e->d_fuse_tl && getenv("VOX_HIP_FUSE_TL")) {      // per-workgroup timeline of the last step's layer-13 launches -> text file
        std::vector<unsigned long long> h((size_t)3 * 1024 * TL_STRIDE);
        FILE *f = fopen(getenv("VOX_HIP_FUSE_TL"), "w");
        if (f && hipMemcpy(h.data(), e->d_fuse_tl, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < 256; b++) if (h[(size_t)TL_STRIDE * b] && h[(size_t)TL_STRIDE * b] < t0) t0 = h[(size_t)TL_STRIDE * b];
            fprintf(f, "# kernel block start_us end_us xcc hw_id stamps...   (kernel 0 = k_dec_attn_fused, 1 = k_gemv_w13x, 2 = k_gemv3 W2; 100 MHz clock)\n");
            for (int k = 0; k < 3; k++)
                for (int b = 0; b < 1024; b++) {
                    const unsigned long long *r = &h[(size_t)(k * 1024 + b) * TL_STRIDE];
                    if (!r[0]) continue;
                    fprintf(f, "%d %d %.2f %.2f %u %u", k, b, (double)(r[0] - t0) / 100.0, (double)(r[1] - t0) / 100.0,
                            (unsigned)(r[2] >> 32), (unsigned)r[2]);
                    for (int q = 3; q < TL_STRIDE; q++) fprintf(f, " %.2f", r[q] ? (double)(r[q] - t0) / 100.0 : -1.0);
                    fprintf(f, "\n");
                }
        }
        if (f) fclose(f);
    }
    return (double)ms * 1e-3 / iters;
}

// Encoder stack on an n-row chunk behind ctx_rows positions, timed with HIP events (see vox_hip.h).  The K/V rings are zeroed
// first (attention over zero keys: finite values, the same memory traffic), x is a zero chunk (finite through every norm:
// RMSNorm of 0 is 0 * rsqrt(eps)); only the time matters here, the parity tests check the values.
extern "C" double vox_hip_time_encoder_rows(vox_hip_engine_t *e, int n_rows, int ctx_rows, int iters) {
    if (!e || n_rows <= 0 || iters <= 0 || ctx_rows < 0) return -1.0;
    if (hipSetDevice(e->device) != hipSuccess) return -1.0;
    vox_hip_reset_encoder(e);
    const int ED = e->d.enc_dim;
    if (ensure(e, e->stmp_in, (size_t)n_rows * ED * 4) || ensure(e, e->stmp_out, (size_t)n_rows * ED * 4)) return -1.0;
    for (auto &L : e->enc) {
        hipMemsetAsync(L.kring, 0, (size_t)e->enc_ring_cap * e->enc_qd * 4, e->stream);
        hipMemsetAsync(L.vring, 0, (size_t)e->enc_ring_cap * e->enc_qd * 4, e->stream);
    }
    float ms = 0.f;
    for (int it = -2; it < iters; it++) {
        if (it == 0) hipEventRecord(e->ev0, e->stream);
        hipMemsetAsync(e->stmp_in.p, 0, (size_t)n_rows * ED * 4, e->stream);
        e->enc_pos = ctx_rows;
        e->enc_stack_now = true;
        const int erc = encoder_rows_dev(e, (float *)e->stmp_in.p, n_rows, (float *)e->stmp_out.p);
        e->enc_stack_now = false;
        if (erc) { vox_hip_reset_encoder(e); return -1.0; }
    }
    hipEventRecord(e->ev1, e->stream);
    enc_stack_fetch(e);
    esync(e);
    if (enc_stack_failed(e)) { vox_hip_reset_encoder(e); return -1.0; }       // (a flagged pass is not a measurement)
    hipEventElapsedTime(&ms, e->ev0, e->ev1);
    if (e->d_enc_tl && getenv("VOX_HIP_ENC_TL")) {      // per-workgroup timeline of the last pass's mid-stack GEMM launches -> text file
        std::vector<unsigned long long> h((size_t)4 * 1024 * TL_STRIDE);
        FILE *f = fopen(getenv("VOX_HIP_ENC_TL"), "w");
        if (f && hipMemcpy(h.data(), e->d_enc_tl, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < (size_t)4 * 1024; i++) if (h[i * TL_STRIDE] && h[i * TL_STRIDE] < t0) t0 = h[i * TL_STRIDE];
            fprintf(f, "# kernel block start_us end_us xcc hw_id stamps(issued, chunk0 done, compute done, tiles in LDS, reduced)   kernel 0 qkv, 1 wo, 2 w1;w3, 3 w2\n");
            for (int k = 0; k < 4; k++)
                for (int b = 0; b < 1024; b++) {
                    const unsigned long long *r = &h[(size_t)(k * 1024 + b) * TL_STRIDE];
                    if (!r[0]) continue;
                    fprintf(f, "%d %d %.2f %.2f %u %u", k, b, (double)(r[0] - t0) / 100.0, (double)(r[1] - t0) / 100.0, (unsigned)(r[2] >> 32), (unsigned)r[2]);
                    for (int q = 3; q < 8; q++) fprintf(f, " %.2f", r[q] ? (double)(r[q] - t0) / 100.0 : -1.0);
                    fprintf(f, "\n");
                }
        }
        if (f) fclose(f);
    }
    if (e->d_es_tl && getenv("VOX_HIP_ENC_TL")) {       // the stack kernel's per-workgroup phase stamps of the mid-stack layer (last pass) -> <file>.stack
        std::vector<unsigned long long> h((size_t)ES_WGS * ES_TL_STRIDE);
        const std::string fn = std::string(getenv("VOX_HIP_ENC_TL")) + ".stack";
        FILE *f = fopen(fn.c_str(), "w");
        if (f && hipMemcpy(h.data(), e->d_es_tl, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(f, "# block kernel_start_us now_us xcc hw_id | per phase P1 P2 P3 F3 P4 P5 F5: wait_done_us body_done_us (relative to the first P1 wait_done; 100 MHz clock)\n");
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < ES_WGS; b++) if (h[(size_t)b * ES_TL_STRIDE + 3] && h[(size_t)b * ES_TL_STRIDE + 3] < t0) t0 = h[(size_t)b * ES_TL_STRIDE + 3];
            for (int b = 0; b < ES_WGS; b++) {
                const unsigned long long *r = &h[(size_t)b * ES_TL_STRIDE];
                if (!r[0]) continue;
                fprintf(f, "%d %.2f %.2f %u %u", b, ((double)r[0] - (double)t0) / 100.0, ((double)r[1] - (double)t0) / 100.0, (unsigned)(r[2] >> 32), (unsigned)r[2]);
                for (int q = 0; q < 2 * ES_PHASES; q++) fprintf(f, " %.2f", ((double)r[3 + q] - (double)t0) / 100.0);
                fprintf(f, "\n");
            }
        }
        if (f) fclose(f);
    }
    vox_hip_reset_encoder(e);
    return (double)ms * 1e-3 / iters;
}

// What a kernel boundary costs on this stack: n back-to-back launches of a kernel that only reads
// its argument and writes one word per block, with a GEMV-sized grid.  Returns seconds per launch.
__global__ __launch_bounds__(256) void k_boundary_probe(float *sink, int blocks_that_write) {
    if (threadIdx.x == 0 && (int)blockIdx.x < blocks_that_write) sink[blockIdx.x] = (float)blockIdx.x;
}
extern "C" double vox_hip_time_empty_launches(vox_hip_engine_t *e, int n, int grid) {
    if (!e || n <= 0 || grid <= 0) return -1.0;
    if (hipSetDevice(e->device) != hipSuccess) return -1.0;
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_boundary_probe, dim3(grid), dim3(256), 0, e->stream, e->dh, 16);
    hipEventRecord(e->ev0, e->stream);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_boundary_probe, dim3(grid), dim3(256), 0, e->stream, e->dh, 16);
    hipEventRecord(e->ev1, e->stream);
    esync(e);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e->ev0, e->ev1);
    return (double)ms * 1e-3 / n;
}
// The same chain captured once into a hipGraph and replayed: no host launch cost in the timed
// region, i.e. the GPU-side boundary alone.
extern "C" double vox_hip_time_empty_launches_graph(vox_hip_engine_t *e, int n, int grid) {
    if (!e || n <= 0 || grid <= 0) return -1.0;
    if (hipSetDevice(e->device) != hipSuccess) return -1.0;
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    esync(e);
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_boundary_probe, dim3(grid), dim3(256), 0, e->stream, e->dh, 16);
    if (hipStreamEndCapture(e->stream, &g) != hipSuccess || !g) { (void)hipGetLastError(); return -1.0; }
    double r = -1.0;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
        hipGraphLaunch(ge, e->stream);                         // warm-up replay
        hipEventRecord(e->ev0, e->stream);
        hipGraphLaunch(ge, e->stream);
        hipEventRecord(e->ev1, e->stream);
        esync(e);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e->ev0, e->ev1);
        r = (double)ms * 1e-3 / n;
        hipGraphExecDestroy(ge);
    } else (void)hipGetLastError();
    hipGraphDestroy(g);
    return r;
}

// In-situ cost of one kernel kind: seconds per step with and without its launches (same stream,
// same neighbours, no events in between); (full - skipped) / launches_per_step is the time the
// kernel adds to the chain, boundary included.  kind: 1 qkv, 2 attention, 4 wo, 5 swiglu, 6 w2.
extern "C" int vox_hip_time_decoder_step_without(vox_hip_engine_t *e, int iters, int kv_len, int kind,
                                                 double *full_s, double *skipped_s) {
    if (!e || kind <= 0 || kind >= PK_LOGITS) return -1;
    const double a = vox_hip_time_decoder_step(e, iters, kv_len);
    e->skip_kinds = 1u << kind;
    if (kind == PK_SWIGLU && e->use_fused && e->use_ffn && !e->use_fp8) e->skip_kinds |= 1u << PK_W2;      // k_ffn_fused is ONE launch: w1;w3 and w2 go together
    // (kind 6 alone, where vox_hip_merged_launches_per_step() > 0: the step without its k_ffn_attn12 launches)
    const double b = vox_hip_time_decoder_step(e, iters, kv_len);
    e->skip_kinds = 0;
    if (full_s) *full_s = a;
    if (skipped_s) *skipped_s = b;
    return (a > 0 && b > 0) ? 0 : -1;
}

static bool merged_static_ok(const vox_hip_engine *e) {
    const vox_hip_dims_t &d = e->d;
    const bool fast = e->use_fast && d.dec_dim == 3072 && e->dec_qd == 4096 && e->dec_kvd == 1024 && d.dec_hidden == 9216;
    return fast && e->use_fused && e->use_dpp && e->merge12 == 2 && e->use_ffn && !e->use_fp8 && !e->sim_on && (e->pf_units == 0 || (e->pf_when == 3 && e->pf_member_units == 0));
}
extern "C" int vox_hip_merged_launches_per_step(const vox_hip_engine_t *e, int kv_len) {
    if (!e || !merged_static_ok(e)) return 0;
    const int kl = std::min(std::max(kv_len, 1), e->d.dec_window);
    return (kl <= std::max(512, e->merge12_maxkeys) || e->merge12_long) ? e->d.dec_layers - 1 : 0;
}

// Layers whose blocks run inside the ONE k_dec_stack launch of a decode step at this KV length (0 = the stack kernel is not used there).
extern "C" int vox_hip_stack_layers(const vox_hip_engine_t *e, int kv_len) {
    if (!e || !merged_static_ok(e) || !e->use_stack || e->d.dec_layers < 2) return 0;
    const int kl = std::min(std::max(kv_len, 1), e->d.dec_window);
    return kl <= std::max(512, e->merge12_maxkeys) ? e->d.dec_layers : 0;
}

// Per-kernel average durations of the decode step, measured with HIP events recorded on the
// engine stream between consecutive launches (event i+1 - event i is attributed to the kernel
// launched in between).  avg_us[PK_COUNT], launches[PK_COUNT] (per step).  Returns seconds/step.
extern "C" double vox_hip_profile_decode(vox_hip_engine_t *e, int iters, int kv_len, double *avg_us, int *launches) {
    if (!e || iters <= 0) return -1.0;
    if (hipSetDevice(e->device) != hipSuccess) return -1.0;
    const int saved_pos = e->dec_pos;
    const int pos = std::max(0, kv_len - 1);
    hipMemsetAsync(e->dx, 0, (size_t)e->d.dec_dim * 4, e->stream);
    if (e->adapter_total - e->adapter_row0 < 1) hipMemsetAsync(e->adapter, 0, (size_t)e->d.dec_dim * 4, e->stream);
    if (set_state(e, pos, 1, 0)) return -1.0;
    for (int i = 0; i < 2; i++) enqueue_step(e, pos, true, e->dlogits, -1, 0);
    esync(e);
    double sum_us[PK_COUNT] = {0}; long cnt[PK_COUNT] = {0};
    double total = 0;
    for (int it = 0; it < iters; it++) {
        e->prof_on = true; e->prof_used = 0;
        prof_mark(e, -1);
        enqueue_step(e, pos, true, e->dlogits, -1, 0);
        e->prof_on = false;
        esync(e);
        for (size_t i = 1; i < e->prof_used; i++) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, e->prof_ev[i - 1], e->prof_ev[i]);
            const int k = e->prof_kind[i];
            if (k >= 0 && k < PK_COUNT) { sum_us[k] += ms * 1e3; cnt[k]++; }
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e->prof_ev[0], e->prof_ev[e->prof_used - 1]);
        total += ms;
    }
    for (int k = 0; k < PK_COUNT; k++) {
        if (avg_us) avg_us[k] = cnt[k] ? sum_us[k] / cnt[k] : 0.0;
        if (launches) launches[k] = (int)(cnt[k] / iters);
    }
    e->dec_pos = saved_pos;
    return total * 1e-3 / iters;
}

// BASELINE config 5: fp8 (e4m3, one f32 scale per output row) copies of the decoder matrices and
// of the tied embedding for the HBM-bound decode GEMVs (3.43 GB per token instead of 6.86 GB).
// Prefill and the encoder keep the bf16 weights.  Call after the bf16 uploads.  0 / -1.
extern "C" int vox_hip_quantize_decoder_fp8(vox_hip_engine_t *e) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    if (e->up && e->up->flush()) return -1;
    const vox_hip_dims_t &d = e->d;
    const int DD = d.dec_dim, DQ = e->dec_qd, DKV = e->dec_kvd, DH = d.dec_hidden;
    if (!(DD == 3072 && DQ == 4096 && DKV == 1024 && DH == 9216)) { g_err = "fp8 decode weights: 4B geometry only"; return -1; }
    auto quant = [&](const uint16_t *W, int N, int K, uint8_t **Q, float **S) -> int {
        if (dalloc(e, Q, (size_t)N * K) || dalloc(e, S, (size_t)N)) return -1;
        hipLaunchKernelGGL(k_quant_fp8_rows, dim3(N), dim3(256), 0, e->stream, W, *Q, *S, K);
        return 0;
    };
    for (int l = 0; l < d.dec_layers; l++) {
        DecLayer &L = e->dec[l];
        if (quant(L.wqkv, DQ + 2 * DKV, DD, &L.wqkv8, &L.sqkv) || quant(L.wo, DD, DQ, &L.wo8, &L.so) ||
            quant(L.w13, 2 * DH, DD, &L.w138, &L.s13) || quant(L.w2, DD, DH, &L.w28, &L.s2)) return -1;
    }
    if (quant(e->tok_emb, d.vocab, DD, &e->tok_emb8, &e->stok)) return -1;
    HC(esync(e));
    HC(hipGetLastError());
    e->use_fp8 = true;
    // Start-up cross-check of the fp8 MFMA family (round 6, advisor): k_rowsgemm_f8 on 38 rows x the first 256 quantised rows of layer 0's
    // wq;wk;wv against the same e4m3 weights dequantised and multiplied in f32 (k_fp8_ref_gemm).  A wrong fragment layout is an O(1)
    // error, the activation split an O(1e-3) one; on a mismatch the prefill stays on the bf16 matrices (VOX_PATH_FP8_MFMA is not reported).
    if (e->use_mfma && !e->fp8_prefill_bf16 && d.dec_layers > 0 && e->dec[0].wqkv8) {
        const int M = 38, N = 256, K = DD;
        std::vector<float> hx((size_t)M * K);
        unsigned lcg = 12345u;
        for (auto &v : hx) { lcg = lcg * 1664525u + 1013904223u; v = ((int)(lcg >> 8) % 20001 - 10000) * 3e-4f; }     // uniform in (-3, 3)
        float *dx = nullptr, *y0 = nullptr, *y1 = nullptr, *part = nullptr;
        bool ok = hipMalloc((void **)&dx, hx.size() * 4) == hipSuccess && hipMalloc((void **)&y0, (size_t)M * N * 4) == hipSuccess &&
                  hipMalloc((void **)&y1, (size_t)M * N * 4) == hipSuccess && hipMalloc((void **)&part, rgf8_partial_bytes(M, N, K)) == hipSuccess &&
                  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        double num = 0.0, den = 0.0;
        if (ok) {
            const int S = launch_rowsgemm_f8(e, dx, K, M, e->dec[0].wqkv8, e->dec[0].sqkv, N, K, part);
            GemmArgs ra{nullptr, 0, nullptr, y1, N, M, N, K, nullptr, nullptr, 0, ACT_NONE, S, 0, part};
            if (S > 0) hipLaunchKernelGGL(k_splitk_reduce, dim3(grid1d((size_t)M * N)), dim3(256), 0, e->stream, ra);
            hipLaunchKernelGGL(k_fp8_ref_gemm, dim3(N), dim3(64), 0, e->stream, y0, (const float *)dx, (const uint8_t *)e->dec[0].wqkv8,
                               (const float *)e->dec[0].sqkv, (const float *)nullptr, M, N, K);
            std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
            ok = S > 0 && esync(e) == hipSuccess && hipGetLastError() == hipSuccess &&
                 hipMemcpy(h0.data(), y0, h0.size() * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(h1.data(), y1, h1.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
            for (size_t i = 0; ok && i < h0.size(); i++) { const double dlt = (double)h1[i] - h0[i]; num += dlt * dlt; den += (double)h0[i] * h0[i]; }
            ok = ok && den > 0.0 && num <= 4e-4 * den;          // rms error below 2 % of rms(y)
        }
        (void)hipMemsetAsync(e->d_f8_clamped, 0, 16, e->stream);   // (the check's own activations are not a prefill)
        hipFree(dx); hipFree(y0); hipFree(y1); hipFree(part);
        if (!ok) {
            (void)hipGetLastError();
            e->fp8_prefill_bf16 = true;
            fprintf(stderr, "vox_hip: WARNING k_rowsgemm_f8 failed its start-up cross-check against the dequantised f32 product (relative rms error %.3g): "
                            "fp8 mode's prefill runs on the bf16 matrices\n", den > 0.0 ? sqrt(num / den) : -1.0);
        }
    }
    return 0;
}
extern "C" int vox_hip_weight_format(vox_hip_engine_t *e) { return e ? (e->use_fp8 ? 1 : 0) : -1; }

// Agreement study for BASELINE config 5 (tools/fp8_agreement.py): what would block-scaled fp8 weights do to the greedy ids?
// An e4m3 value times a POWER-OF-TWO scale is exactly a bf16 value, so a block-scaled fp8 GEMV with such scales is
// bit-for-bit the bf16 GEMV on the dequantised weights: copies of the decoder matrices (and optionally of the LM head) are
// rewritten as dequant(quant(w)) with one 2^k scale per `block` weights of a row (block 0 = per row, 32 = the MX granularity,
// 128 = SURVEY 7 step 9) and the decode step - fused path, bf16 kernels - streams those.  Prefill and encoder keep the
// originals, as in the real fp8 mode.  block < 0 switches the simulation off.  4B fused geometry only; not with fp8 mode.
__global__ __launch_bounds__(256) void k_sim_fp8_blocks(const uint16_t *W, uint16_t *Q, int K, int block) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const uint16_t *w = W + (size_t)row * K;
    uint16_t *q = Q + (size_t)row * K;
    const int B = block > 0 ? block : K;
    for (int b0 = 0; b0 < K; b0 += B) {
        float amax = 0.f;
        for (int k = tid; k < B; k += 256) amax = fmaxf(amax, fabsf(bf16_to_f32(w[b0 + k])));
        amax = wave_max(amax);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // smallest power of two with amax / scale <= 448 (the largest finite e4m3 value)
        float sc = 1.0f;
        if (amax > 0.f) { int ex; (void)frexpf(amax / 448.0f, &ex); sc = ldexpf(1.0f, ex); if (amax / ldexpf(1.0f, ex - 1) <= 448.0f) sc = ldexpf(1.0f, ex - 1); }
        const float inv = 1.0f / sc;
        for (int k2 = tid; k2 < B / 2; k2 += 256) {
            const float v0 = bf16_to_f32(w[b0 + 2 * k2]) * inv, v1 = bf16_to_f32(w[b0 + 2 * k2 + 1]) * inv;
            const int word = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false);
            const f32x2 d = __builtin_amdgcn_cvt_pk_f32_fp8(word, false);
            q[b0 + 2 * k2] = (uint16_t)(__float_as_uint(d.x * sc) >> 16);          // exact: 4 significant bits x 2^k
            q[b0 + 2 * k2 + 1] = (uint16_t)(__float_as_uint(d.y * sc) >> 16);
        }
    }
}
extern "C" int vox_hip_simulate_block_fp8(vox_hip_engine_t *e, int block, int lm_head) {
    if (!e) return -1;
    HC(hipSetDevice(e->device));
    if (block < 0) { HC(esync(e)); e->sim_on = false; return 0; }
    const vox_hip_dims_t &d = e->d;
    const int DD = d.dec_dim, DQ = e->dec_qd, DKV = e->dec_kvd, DH = d.dec_hidden;
    if (!e->fused_ok || e->use_fp8) { g_err = "simulate_block_fp8: needs the fused bf16 decode path"; return -1; }
    if (block != 0 && (block % 2 || DD % block || DQ % block || DH % block)) { g_err = "simulate_block_fp8: bad block size"; return -1; }
    if (e->up && e->up->flush()) return -1;
    HC(esync(e));
    auto sim = [&](const uint16_t *W, int N, int K, uint16_t **Q) -> int {
        if (!*Q && dalloc(e, Q, (size_t)N * K)) return -1;
        hipLaunchKernelGGL(k_sim_fp8_blocks, dim3(N), dim3(256), 0, e->stream, W, *Q, K, block);
        return 0;
    };
    for (int l = 0; l < d.dec_layers; l++) {
        DecLayer &L = e->dec[l];
        if (sim(L.wqkv, DQ + 2 * DKV, DD, &L.wqkv_s) || sim(L.wo, DD, DQ, &L.wo_s) || sim(L.w13, 2 * DH, DD, &L.w13_s) || sim(L.w2, DD, DH, &L.w2_s)) return -1;
    }
    if (lm_head && sim(e->tok_emb, d.vocab, DD, &e->tok_emb_s)) return -1;
    HC(esync(e));
    HC(hipGetLastError());
    e->sim_on = true; e->sim_lm = lm_head != 0;
    return 0;
}

// Experiment hook: time `iters` passes over the five decode kernels of ONE layer (233 MB of
// weights, which fit the 256 MB Infinity Cache) to see what the same launches cost when the
// weights are cache-resident instead of streamed from HBM.  Returns seconds per layer pass.
extern "C" double vox_hip_time_layer_repeat(vox_hip_engine_t *e, int iters, int kv_len, double *avg_us, int *launches) {
    if (!e || iters <= 0 || e->d.dec_layers < 1) return -1.0;
    const int saved_layers = e->d.dec_layers, saved_vocab = e->d.vocab, saved_grid = e->logits_grid;
    e->d.dec_layers = 1;                       // enqueue_step walks layers [0, dec_layers)
    e->d.vocab = 16; e->logits_grid = 1;       // shrink the logits pass to a stub (805 MB would flush the cache)
    const double r = vox_hip_profile_decode(e, iters, kv_len, avg_us, launches);
    e->d.dec_layers = saved_layers; e->d.vocab = saved_vocab; e->logits_grid = saved_grid;
    return r;
}

// ------------------------------------------------------------------------------------
// start-up self-tests: DPP row reduction and MFMA fragment layout, each against a plain
// reference kernel.  A mismatch is loud and switches to the slower-but-plain HIP variant;
// nothing ever falls back to the CPU.
// ------------------------------------------------------------------------------------
static int self_test(vox_hip_engine *e) {
    if (hipFuncSetAttribute((const void *)k_gemm_mfma_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM_X3_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        e->use_bf16x3 = false;
    }
    // (1) DPP
    float *d_a = nullptr, *d_b = nullptr;
    HC(hipMalloc((void **)&d_a, 64 * 4)); HC(hipMalloc((void **)&d_b, 64 * 4));
    hipLaunchKernelGGL(k_dpp_selftest, dim3(1), dim3(64), 0, e->stream, d_a, d_b);
    float ha[64], hb[64];
    HC(hipMemcpyAsync(ha, d_a, sizeof ha, hipMemcpyDeviceToHost, e->stream));
    HC(hipMemcpyAsync(hb, d_b, sizeof hb, hipMemcpyDeviceToHost, e->stream));
    HC(esync(e));
    bool ok = true;
    for (int i = 0; i < 64; i++) if (fabsf(ha[i] - hb[i]) > 1e-3f * fabsf(hb[i])) ok = false;
    int failed = 0;
    if (!ok) { fprintf(stderr, "vox_hip: DPP row reduction self-test FAILED\n"); e->use_dpp = false; failed++; }
    hipFree(d_a); hipFree(d_b);
    if (vox_disabled("dpp")) e->use_dpp = false;

    // (2) MFMA GEMMs (bf16x3 and f32-input) vs the scalar kernel on an asymmetric 160 x 192 x 128 problem
    const int M = 160, N = 192, K = 128;
    std::vector<float> hx((size_t)M * K), hy0((size_t)M * N), hy1((size_t)M * N), hy2((size_t)M * N), hbias(N);
    std::vector<uint16_t> hw((size_t)N * K);
    for (int m = 0; m < M; m++) for (int k = 0; k < K; k++) hx[(size_t)m * K + k] = 0.01f * (float)((m * 7 + k * 3) % 29 - 14) + 0.001f * m;
    for (int n = 0; n < N; n++) for (int k = 0; k < K; k++) {
        float v = 0.02f * (float)((n * 5 + k * 11) % 23 - 11) - 0.0005f * n;
        uint32_t u; memcpy(&u, &v, 4); hw[(size_t)n * K + k] = (uint16_t)(u >> 16);
    }
    for (int n = 0; n < N; n++) hbias[n] = 0.1f * (float)(n % 7);
    float *dx, *dy0, *dy1, *dy2, *dbias; uint16_t *dw;
    HC(hipMalloc((void **)&dx, hx.size() * 4)); HC(hipMalloc((void **)&dy0, hy0.size() * 4));
    HC(hipMalloc((void **)&dy1, hy1.size() * 4)); HC(hipMalloc((void **)&dy2, hy2.size() * 4));
    HC(hipMalloc((void **)&dw, hw.size() * 2)); HC(hipMalloc((void **)&dbias, N * 4));
    HC(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    HC(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    launch_gemm(e, dx, K, dw, dy0, N, M, N, K, dbias, nullptr, 0, ACT_NONE, 0);
    launch_gemm(e, dx, K, dw, dy1, N, M, N, K, dbias, nullptr, 0, ACT_NONE, 2);
    launch_gemm(e, dx, K, dw, dy2, N, M, N, K, dbias, nullptr, 0, ACT_NONE, 1);
    HC(esync(e));
    HC(hipMemcpy(hy0.data(), dy0, hy0.size() * 4, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hy1.data(), dy1, hy1.size() * 4, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hy2.data(), dy2, hy2.size() * 4, hipMemcpyDeviceToHost));
    double maxd0 = 0, maxd = 0;
    for (size_t i = 0; i < hy1.size(); i++) {
        maxd0 = std::max(maxd0, (double)fabsf(hy0[i] - hy2[i]));
        maxd = std::max(maxd, (double)fabsf(hy1[i] - hy2[i]));
    }
    if (!(maxd0 < 2e-5)) {
        fprintf(stderr, "vox_hip: bf16x3 MFMA GEMM self-test FAILED (max diff %g)\n", maxd0);
        e->use_bf16x3 = false; failed++;
    }
    if (!(maxd < 2e-5)) {
        fprintf(stderr, "vox_hip: f32-input MFMA GEMM self-test FAILED (max diff %g)\n", maxd);
        e->use_mfma = false; failed++;
    }
    if (vox_disabled("bf16x3")) e->use_bf16x3 = false;
    {   // (2b) the planes GEMM (pre-split activations, LDS-DMA pipeline) on the same problem, with and without split-K
        bool okp = hipFuncSetAttribute((const void *)k_gemm_planes<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(2)) == hipSuccess &&
                   hipFuncSetAttribute((const void *)k_gemm_planes<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(4)) == hipSuccess &&
                   hipFuncSetAttribute((const void *)k_gemm_planes<2, 2, GP_EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(2)) == hipSuccess &&
                   hipFuncSetAttribute((const void *)k_gemm_planes<2, 2, GP_EPI_ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(2)) == hipSuccess &&
                   hipFuncSetAttribute((const void *)k_gemm_planes<2, 4, GP_EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(4)) == hipSuccess &&
                   hipFuncSetAttribute((const void *)k_gemm_planes<2, 4, GP_EPI_ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gp_stage_bytes(4)) == hipSuccess;
        uint16_t *dp = nullptr;
        okp = okp && hipMalloc((void **)&dp, (size_t)3 * M * K * 2) == hipSuccess;
        if (okp) {
            hipLaunchKernelGGL(k_split_planes, dim3(grid1d((size_t)M * K / 4)), dim3(256), 0, e->stream, dp, (size_t)M * K, (const float *)dx, K, M, K);
            double worst = 0;
            for (int pass = 0; pass < 2 && okp; pass++) {
                const bool sk = e->use_splitk;
                if (pass == 1) e->use_splitk = false;
                okp = launch_gemm_planes(e, dp, (size_t)M * K, K, dw, dy0, N, M, N, K, dbias, nullptr, 0, ACT_NONE) == 0;
                e->use_splitk = sk;
                okp = okp && esync(e) == hipSuccess &&
                      hipMemcpy(hy0.data(), dy0, hy0.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
                for (size_t i = 0; okp && i < hy0.size(); i++) worst = std::max(worst, (double)fabsf(hy0[i] - hy2[i]));
            }
            if (!(worst < 2e-5)) okp = false;
            if (!okp) fprintf(stderr, "vox_hip: planes GEMM self-test FAILED (max diff %g)\n", worst);
        }
        if (dp) hipFree(dp);
        if (!okp) { (void)hipGetLastError(); e->use_planes = false; failed++; }
        if (vox_disabled("planes")) e->use_planes = false;
        if (vox_disabled("epi")) e->use_epi = false;                         // separate RoPE / SiLU launches
        if (vox_disabled("attn_small")) e->use_attn_small = false;
        if (vox_disabled("attn_merge")) e->attn_merge = false;
        if (vox_disabled("staged_upload")) e->use_staged_upload = false;
    }

    // (3) MFMA encoder attention vs the thread-per-query kernel: 200 queries, 2 heads, window 90
    {
        const int nq = 200, nh = 2, hd = 64, ld = nh * hd, win = 90;
        std::vector<float> hq((size_t)nq * ld), hk((size_t)nq * ld), hv((size_t)nq * ld), r1((size_t)nq * ld), r2((size_t)nq * ld);
        for (size_t i = 0; i < hq.size(); i++) {
            hq[i] = 0.05f * (float)((int)((i * 2654435761u) >> 20 & 63) - 31);
            hk[i] = 0.04f * (float)((int)((i * 40503u + 77u) >> 7 & 63) - 30);
            hv[i] = 0.03f * (float)((int)((i * 9973u + 5u) >> 3 & 127) - 64);
        }
        float *dq, *dk, *dv, *do1, *do2;
        HC(hipMalloc((void **)&dq, hq.size() * 4)); HC(hipMalloc((void **)&dk, hq.size() * 4)); HC(hipMalloc((void **)&dv, hq.size() * 4));
        HC(hipMalloc((void **)&do1, hq.size() * 4)); HC(hipMalloc((void **)&do2, hq.size() * 4));
        HC(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
        HC(hipMemcpy(dk, hk.data(), hq.size() * 4, hipMemcpyHostToDevice));
        HC(hipMemcpy(dv, hv.data(), hq.size() * 4, hipMemcpyHostToDevice));
        AttnArgs a{};
        a.ldo = ld; a.q = dq; a.ldq = ld; a.n_q = nq; a.qpos0 = 0; a.kB = dk; a.vB = dv; a.ldB = ld; a.posB0 = 0;
        a.last_key = nq - 1; a.kA = dk; a.vA = dv; a.capA = 1 << 30; a.ldA = ld; a.n_heads = nh; a.n_kv_heads = nh;
        a.scale = 0.125f; a.window = win; a.st = nullptr;
        a.out = do2;
        hipLaunchKernelGGL((k_attn_rows<64>), dim3((nq + 127) / 128, nh), dim3(128), 0, e->stream, a);
        a.out = do1;
        hipLaunchKernelGGL(k_attn_enc_bf16, dim3((nq + 127) / 128, nh), dim3(256), 0, e->stream, a);
        HC(esync(e));
        HC(hipMemcpy(r1.data(), do1, r1.size() * 4, hipMemcpyDeviceToHost));
        HC(hipMemcpy(r2.data(), do2, r2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (size_t i = 0; i < r1.size(); i++) md = std::max(md, (double)fabsf(r1[i] - r2[i]));
        if (!(md < 1e-4)) {
            fprintf(stderr, "vox_hip: bf16-split MFMA attention self-test FAILED (max diff %g)\n", md);
            e->use_attn_mfma = false; failed++;
        }
        hipFree(dq); hipFree(dk); hipFree(dv); hipFree(do1); hipFree(do2);
        if (vox_disabled("attn_mfma")) e->use_attn_mfma = false;
    }
    if (vox_disabled("mfma")) e->use_mfma = false;
    if (vox_disabled("fast")) e->use_fast = false;
    if (vox_disabled("splitk")) e->use_splitk = false;
    if (vox_disabled("skinny")) e->use_skinny = false;
    if (hipFuncSetAttribute((const void *)k_skinny<SK_SWIGLU, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_WPB * 2 * 4096) != hipSuccess) {
        (void)hipGetLastError();
        e->use_skinny = false;
    }
    {   // k_enc_stack (round 6): the 4B encoder's shapes on a 256-CU part, all CUs available to the stream; it falls back on k_skinny's launches
        hipDeviceProp_t prop;
        const vox_hip_dims_t &d = e->d;
        e->enc_stack_ok = e->use_skinny && e->use_mfma && e->use_dpp && !vox_disabled("enc_stack") && !getenv("VOX_HIP_CUMASK") &&
                          d.enc_dim == ES_D && e->enc_qd == ES_QD && d.enc_hidden == ES_H && d.enc_heads == ES_HEADS && d.enc_head_dim == ES_HD &&
                          hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount == ES_WGS;
    }
    {   // k_rowsgemm: up to 3 planes x 128 rows x 2 chunks (or 96 rows x 4 chunks) of activations in LDS
        const int rg_lds = 2 * 3 * 32 * 6 * 128;        // the largest request the launcher can make (mt * cpw <= 6)
#define RG_FN(WPB, CPW) (const void *)k_rowsgemm<WPB, CPW, RG_X_PLANES, 1, 4>, (const void *)k_rowsgemm<WPB, CPW, RG_X_F32, 1, 4>, \
                        (const void *)k_rowsgemm<WPB, CPW, RG_X_PLANES, 1, 8>, (const void *)k_rowsgemm<WPB, CPW, RG_X_F32, 1, 8>, \
                        (const void *)k_rowsgemm<WPB, CPW, RG_X_PLANES, 2, 4>, (const void *)k_rowsgemm<WPB, CPW, RG_X_F32, 2, 4>
        const void *fns[] = {RG_FN(8, 2), RG_FN(8, 1), RG_FN(4, 2), RG_FN(4, 1)};
#undef RG_FN
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, rg_lds) != hipSuccess) {
                (void)hipGetLastError();
                if (e->use_rowsgemm) fprintf(stderr, "vox_hip: WARNING k_rowsgemm cannot get its LDS (%zu bytes): 33 .. 128-row passes fall back to the planes GEMM / k_skinny\n", (size_t)rg_lds);
                e->use_rowsgemm = false;
            }
        // (5) k_rowsgemm against the scalar GEMM (round 4): both activation modes, one 16-row tile / five (the 8-tile accumulator
        // budget) / two weight tiles per wave, K split over workgroups - the variants decoder prefill, the encoder flush pass and the
        // small conv / adapter GEMMs run.  The kernel has one straight-line MFMA body per tile count because of a read-after-write
        // hazard the compiler does not track across branches (vox_rowsgemm.h): a miscompiled variant must not pass silently.
        if (e->use_rowsgemm && e->use_mfma) {
            struct { int n, N, K, planes; } cases[] = {{13, 256, 256, 0}, {70, 256, 256, 0}, {40, 4096, 128, 0}, {38, 512, 384, 1}};
            for (const auto &cs : cases) {
                const int n = cs.n, N2 = cs.N, K2 = cs.K;
                std::vector<float> x((size_t)n * K2), yref((size_t)n * N2);
                std::vector<uint16_t> w((size_t)N2 * K2);
                for (int m = 0; m < n; m++) for (int k = 0; k < K2; k++) x[(size_t)m * K2 + k] = 0.013f * (float)((m * 5 + k * 7) % 31 - 15) + 0.0007f * m;
                for (int r = 0; r < N2; r++) for (int k = 0; k < K2; k++) {
                    const float v = 0.017f * (float)((r * 3 + k * 13) % 19 - 9) - 0.0003f * (r % 64);
                    uint32_t u; memcpy(&u, &v, 4); w[(size_t)r * K2 + k] = (uint16_t)(u >> 16);
                }
                const size_t pbytes = rg_partial_bytes(n, N2, K2);
                float *tx = nullptr, *ty = nullptr, *tp = nullptr; uint16_t *tw = nullptr, *tpl = nullptr;
                HC(hipMalloc((void **)&tx, x.size() * 4)); HC(hipMalloc((void **)&ty, yref.size() * 4)); HC(hipMalloc((void **)&tp, pbytes));
                HC(hipMalloc((void **)&tw, w.size() * 2)); HC(hipMalloc((void **)&tpl, x.size() * 3 * 2));
                HC(hipMemcpy(tx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
                HC(hipMemcpy(tw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
                launch_gemm(e, tx, K2, tw, ty, N2, n, N2, K2, nullptr, nullptr, 0, ACT_NONE, 1);          // scalar reference
                int S = 0;
                if (cs.planes) {
                    hipLaunchKernelGGL(k_split_planes, dim3(grid1d((size_t)n * K2 / 4)), dim3(256), 0, e->stream, tpl, (size_t)n * K2, (const float *)tx, K2, n, K2);
                    S = launch_rowsgemm(e, tpl, (size_t)n * K2, nullptr, 0, n, tw, N2, K2, tp);
                } else {
                    S = launch_rowsgemm(e, nullptr, 0, tx, K2, n, tw, N2, K2, tp);
                }
                HC(esync(e));
                std::vector<float> part((size_t)S * n * N2);
                HC(hipMemcpy(yref.data(), ty, yref.size() * 4, hipMemcpyDeviceToHost));
                HC(hipMemcpy(part.data(), tp, part.size() * 4, hipMemcpyDeviceToHost));
                double worst = 0;
                for (size_t i = 0; i < yref.size(); i++) {
                    float v = 0.f;
                    for (int z = 0; z < S; z++) v += part[(size_t)z * n * N2 + i];
                    worst = std::max(worst, (double)fabsf(v - yref[i]));
                }
                hipFree(tx); hipFree(ty); hipFree(tp); hipFree(tw); hipFree(tpl);
                if (!(worst < 5e-5)) {
                    fprintf(stderr, "vox_hip: k_rowsgemm self-test FAILED (n %d, N %d, K %d, %s activations: max diff %g)\n", n, N2, K2, cs.planes ? "plane" : "f32", worst);
                    e->use_rowsgemm = false; failed++;
                    break;
                }
            }
        }
        if (vox_disabled("rowsgemm")) e->use_rowsgemm = false;
    }
    hipFree(dx); hipFree(dy0); hipFree(dy1); hipFree(dy2); hipFree(dw); hipFree(dbias);
    if (failed) {
        // A production kernel disagreeing with its plain cross-check is a broken build or device, not a
        // tuning matter: refuse to load unless the operator explicitly accepts the slower plain kernels.
        if (!getenv("VOX_HIP_ALLOW_FALLBACK")) {
            g_err = "vox_hip: start-up self-test failed (see stderr); set VOX_HIP_ALLOW_FALLBACK=1 to run on the plain HIP kernels";
            fprintf(stderr, "%s\n", g_err.c_str());
            return -1;
        }
        fprintf(stderr, "vox_hip: WARNING continuing on the plain HIP variants of the failed kernels (VOX_HIP_ALLOW_FALLBACK); "
                        "vox_hip_active_paths() reports which\n");
    }
    return 0;
}

// Which kernel families the engine runs (bit set = production variant live).  Tests assert the full
// mask on gfx950 so that a silent downgrade cannot pass; bench.py prints it.
extern "C" unsigned vox_hip_active_paths(const vox_hip_engine_t *e) {
    if (!e) return 0;
    const vox_hip_dims_t &d = e->d;
    const bool fast_geom = d.dec_dim == 3072 && e->dec_qd == 4096 && e->dec_kvd == 1024 && d.dec_hidden == 9216;
    unsigned m = 0;
    if (e->use_mfma) m |= VOX_PATH_GEMM_MFMA_F32;
    if (e->use_mfma && e->use_bf16x3) m |= VOX_PATH_GEMM_MFMA_BF16X3;
    if (e->use_attn_mfma) m |= VOX_PATH_ATTN_ENC_MFMA;
    if (e->use_dpp) m |= VOX_PATH_ATTN_DEC_DPP;
    if (e->use_splitk) m |= VOX_PATH_GEMM_SPLITK;
    if (fast_geom && e->use_fast) m |= VOX_PATH_GEMV3;
    if (e->use_fp8) m |= VOX_PATH_FP8_DECODE;
    if (e->use_fp8 && e->use_mfma && !e->fp8_prefill_bf16) m |= VOX_PATH_FP8_MFMA;
    if (fast_geom && e->use_fused && e->use_dpp) m |= VOX_PATH_DEC_FUSED;
    if (e->use_skinny && e->use_mfma) m |= VOX_PATH_SKINNY_ENC;
    if (e->use_planes && e->use_mfma && e->use_bf16x3) m |= VOX_PATH_GEMM_PLANES;
    if (fast_geom && e->use_fused && e->use_dpp && e->use_ffn && !e->use_fp8) m |= VOX_PATH_FFN_FUSED;
    if (e->use_rowsgemm && e->use_mfma) m |= VOX_PATH_ROWSGEMM;
    if (merged_static_ok(e)) m |= VOX_PATH_FFN_ATTN12;
    if (merged_static_ok(e) && e->use_stack && e->d.dec_layers > 1) m |= VOX_PATH_DEC_STACK;
    if (e->enc_stack_ok) m |= VOX_PATH_ENC_STACK;
    return m;
}

// ------------------------------------------------------------------------------------
// Kernel-level API of the reference (voxtral_kernels.h:18-159) on host buffers: the device side
// of host/vox_kernels.c.  Generic shapes, synchronous, one temporary allocation per call: a
// correctness / compatibility surface, not a hot path (vox_kernel_api.h).
// ------------------------------------------------------------------------------------
namespace {
struct Tmp {                      // device temporaries of one call, freed on scope exit
    std::vector<void *> ps;
    ~Tmp() { for (void *p : ps) hipFree(p); }
    template <typename T> T *get(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ps.push_back(p);
        return (T *)p;
    }
    template <typename T> T *up(const T *h, size_t n) {
        T *d = get<T>(n);
        if (d && h && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
};
}  // namespace
#define KNULL(p) do { if (!(p)) { g_err = "vox_hip kernel API: device allocation / upload failed"; return -1; } } while (0)

extern "C" int vox_hip_k_eltwise(vox_hip_engine_t *e, float *a, const float *b, float s, size_t n, int op) {
    if (!e || !a) return -1;
    if (n == 0) return 0;
    HC(hipSetDevice(e->device));
    Tmp t;
    float *da = t.up(a, n); KNULL(da);
    float *db = nullptr;
    if (b) { db = t.up(b, n); KNULL(db); }
    hipLaunchKernelGGL(k_eltwise, dim3(grid1d(n)), dim3(256), 0, e->stream, da, (const float *)db, s, n, op);
    LAUNCH_CHECK("k_eltwise");
    HC(esync(e));
    HC(hipMemcpy(a, da, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// C[M,N] = A[M,K] . B (+bias[n]);  b_is_nk: B is [N,K] (C = A B^T), else [K,N].
extern "C" int vox_hip_k_sgemm(vox_hip_engine_t *e, float *C, const float *A, const float *B, const float *bias,
                               int M, int K, int N, int b_is_nk) {
    if (!e || !C || !A || !B || M <= 0 || N <= 0 || K <= 0) return -1;
    HC(hipSetDevice(e->device));
    Tmp t;
    float *dA = t.up(A, (size_t)M * K), *dB = t.up(B, (size_t)N * K), *dC = t.get<float>((size_t)M * N);
    KNULL(dA); KNULL(dB); KNULL(dC);
    float *db = nullptr;
    if (bias) { db = t.up(bias, (size_t)N); KNULL(db); }
    hipLaunchKernelGGL(k_sgemm, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, e->stream, dC, N, (const float *)dA, K,
                       (const float *)dB, (long)(b_is_nk ? 1 : N), (long)(b_is_nk ? K : 1), M, N, K, (const float *)db,
                       (const float *)nullptr);
    LAUNCH_CHECK("k_sgemm");
    HC(esync(e));
    HC(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return 0;
}

// 1-D convolution on channel-major data: in [C_in, L], weight [C_out, C_in*ks], out [C_out, L_out]
// (vox_conv1d / vox_causal_conv1d, voxtral_kernels.c:255-340): im2col + GEMM, like the reference.
extern "C" int vox_hip_k_conv1d(vox_hip_engine_t *e, float *out, const float *in, const float *weight, const float *bias,
                                int c_in, int c_out, int length, int ks, int stride, int pad_left, int out_len) {
    if (!e || !out || !in || !weight || c_in <= 0 || c_out <= 0 || length <= 0 || ks <= 0 || stride <= 0) return -1;
    if (out_len <= 0) return 0;
    HC(hipSetDevice(e->device));
    Tmp t;
    const int K = c_in * ks;
    float *din = t.up(in, (size_t)c_in * length), *dw = t.up(weight, (size_t)c_out * K);
    float *col = t.get<float>((size_t)K * out_len), *dout = t.get<float>((size_t)c_out * out_len);
    KNULL(din); KNULL(dw); KNULL(col); KNULL(dout);
    float *db = nullptr;
    if (bias) { db = t.up(bias, (size_t)c_out); KNULL(db); }
    hipLaunchKernelGGL(k_conv_im2col, dim3(grid1d((size_t)K * out_len)), dim3(256), 0, e->stream, col, (const float *)din, c_in,
                       length, ks, stride, pad_left, out_len);
    // out[C_out, L_out] = W[C_out, K] . col[K, L_out] + bias[oc] (a per-ROW bias)
    hipLaunchKernelGGL(k_sgemm, dim3((out_len + 63) / 64, (c_out + 63) / 64), dim3(256), 0, e->stream, dout, out_len,
                       (const float *)dw, K, (const float *)col, (long)out_len, 1L, c_out, out_len, K, (const float *)nullptr,
                       (const float *)db);
    LAUNCH_CHECK("conv1d");
    HC(esync(e));
    HC(hipMemcpy(out, dout, (size_t)c_out * out_len * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vox_hip_k_rms_norm(vox_hip_engine_t *e, float *out, const float *x, const float *w, int seq, int hidden, float eps) {
    if (!e || !out || !x || !w || seq <= 0 || hidden <= 0) return -1;
    HC(hipSetDevice(e->device));
    Tmp t;
    float *dx = t.up(x, (size_t)seq * hidden), *dw = t.up(w, (size_t)hidden), *dout = t.get<float>((size_t)seq * hidden);
    KNULL(dx); KNULL(dw); KNULL(dout);
    if (hidden % 4 == 0)
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(seq), dim3(256), 0, e->stream, dout, hidden, (const float *)dx, hidden,
                           (const float *)dw, (const float *)nullptr, hidden, eps);
    else
        hipLaunchKernelGGL(k_rmsnorm_generic, dim3(seq), dim3(256), 0, e->stream, dout, (const float *)dx, (const float *)dw, hidden, eps);
    LAUNCH_CHECK("rms_norm");
    HC(esync(e));
    HC(hipMemcpy(out, dout, (size_t)seq * hidden * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vox_hip_k_softmax(vox_hip_engine_t *e, float *x, int rows, int cols) {
    if (!e || !x || rows <= 0 || cols <= 0) return -1;
    HC(hipSetDevice(e->device));
    Tmp t;
    float *dx = t.up(x, (size_t)rows * cols); KNULL(dx);
    hipLaunchKernelGGL(k_softmax_rows, dim3(rows), dim3(256), 0, e->stream, dx, cols);
    LAUNCH_CHECK("softmax");
    HC(esync(e));
    HC(hipMemcpy(x, dx, (size_t)rows * cols * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vox_hip_k_rope_freqs(vox_hip_engine_t *e, float *freqs, const int *pos, int seq, int dim, float theta) {
    if (!e || !freqs || !pos || seq <= 0 || dim < 2) return -1;
    HC(hipSetDevice(e->device));
    Tmp t;
    std::vector<float> f;
    host_inv_freq(f, dim, theta);
    int *dpos = t.up(pos, (size_t)seq);
    float *df = t.up(f.data(), f.size()), *dout = t.get<float>((size_t)seq * (dim / 2) * 2);
    KNULL(dpos); KNULL(df); KNULL(dout);
    hipLaunchKernelGGL(k_rope_freqs, dim3(grid1d((size_t)seq * (dim / 2))), dim3(256), 0, e->stream, dout, (const int *)dpos, seq,
                       dim / 2, (const float *)df);
    LAUNCH_CHECK("rope_freqs");
    HC(esync(e));
    HC(hipMemcpy(freqs, dout, (size_t)seq * (dim / 2) * 2 * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vox_hip_k_apply_rope(vox_hip_engine_t *e, float *x, const float *freqs, int seq, int heads, int head_dim) {
    if (!e || !x || !freqs || seq <= 0 || heads <= 0 || head_dim < 2) return -1;
    HC(hipSetDevice(e->device));
    Tmp t;
    const int hidden = heads * head_dim;
    float *dx = t.up(x, (size_t)seq * hidden), *df = t.up(freqs, (size_t)seq * (head_dim / 2) * 2);
    KNULL(dx); KNULL(df);
    hipLaunchKernelGGL(k_rope_apply, dim3(grid1d((size_t)seq * hidden / 2)), dim3(256), 0, e->stream, dx, hidden, seq, hidden,
                       head_dim, (const float *)df);
    LAUNCH_CHECK("apply_rope");
    HC(esync(e));
    HC(hipMemcpy(x, dx, (size_t)seq * hidden * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------
// Several GPUs in ONE process (libvoxtral's VOX_DEVICES): stream-ordered peer hand-offs between engines.
// The exact context-parallel encoder of SURVEY 8(e) / DESIGN.md without a host round trip per layer: a
// hand-off is a peer copy (xGMI; plain D2D when both engines sit on one device, which is how a 1-GPU box
// tests this) enqueued on the PRODUCER's stream right behind the kernels that produced the data, an event,
// and a hipStreamWaitEvent on the CONSUMER's stream.  The host only enqueues.
// ------------------------------------------------------------------------------------
static int peer_copy_async(vox_hip_engine *dst, void *dptr, vox_hip_engine *src, const void *sptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (dst->device == src->device) HC(hipMemcpyAsync(dptr, sptr, bytes, hipMemcpyDeviceToDevice, s));
    else HC(hipMemcpyPeerAsync(dptr, dst->device, sptr, src->device, bytes, s));
    return 0;
}
// an event behind everything enqueued so far on the producer's stream (from its ring of 64: a sharded chunk records
// 32 layers + 2 per engine, and every one of them has been waited for before the ring comes round)
static int record_xev(vox_hip_engine *producer, hipEvent_t *out) {
    HC(hipSetDevice(producer->device));
    if (producer->xev.empty()) {
        producer->xev.resize(96);
        for (auto &ev : producer->xev) HC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    hipEvent_t ev = producer->xev[producer->xev_next++ % producer->xev.size()];
    HC(hipEventRecord(ev, producer->stream));
    *out = ev;
    return 0;
}
// consumer's stream waits for everything enqueued so far on the producer's stream
static int chain_streams(vox_hip_engine *producer, vox_hip_engine *consumer) {
    if (producer == consumer) return 0;
    hipEvent_t ev;
    if (record_xev(producer, &ev)) return -1;
    HC(hipStreamWaitEvent(consumer->stream, ev, 0));
    return 0;
}

extern "C" int vox_hip_enable_peer(vox_hip_engine_t *a, vox_hip_engine_t *b) {
    if (!a || !b) return -1;
    if (a->device == b->device) return 0;
    int can = 0;
    HC(hipDeviceCanAccessPeer(&can, a->device, b->device));
    if (vox_disabled("peer")) can = 0;       // (tests: a node whose GPUs cannot map each other - the runtime then stages hipMemcpyPeerAsync through the host)
    if (can) {
        HC(hipSetDevice(a->device));
        hipError_t r = hipDeviceEnablePeerAccess(b->device, 0);
        if (r != hipSuccess && r != hipErrorPeerAccessAlreadyEnabled) { set_err("hipDeviceEnablePeerAccess", r, __FILE__, __LINE__); return -1; }
        (void)hipGetLastError();
    }
    return 0;       // without peer access the copies are staged by the runtime: slower, still correct
}

// Encoder-side weights (conv stem, encoder layers, adapter, final norm, mel tables) of `src` into `dst` (same geometry):
// the extra GPUs of a multi-device model only ever run encoder shards.  Synchronous (load time).
extern "C" int vox_hip_clone_encoder_weights(vox_hip_engine_t *dst, vox_hip_engine_t *src) {
    if (!dst || !src || dst->d.enc_layers != src->d.enc_layers || dst->d.enc_dim != src->d.enc_dim || dst->d.enc_hidden != src->d.enc_hidden ||
        dst->d.dec_dim != src->d.dec_dim || dst->d.mel_bins != src->d.mel_bins) { g_err = "vox_hip_clone_encoder_weights: geometry mismatch"; return -1; }
    const vox_hip_dims_t &d = src->d;
    const size_t ED = d.enc_dim, EQ = src->enc_qd, EH = d.enc_hidden, DD = d.dec_dim;
    HC(hipSetDevice(src->device));
    if (src->up && src->up->flush()) return -1;
    HC(esync(src));
    hipStream_t s = src->stream;
    auto cp = [&](void *dp, const void *sp, size_t bytes) { return peer_copy_async(dst, dp, src, sp, bytes, s); };
    int rc = 0;
    rc |= cp(dst->conv0_w, src->conv0_w, ED * d.mel_bins * 3 * 2); rc |= cp(dst->conv1_w, src->conv1_w, ED * ED * 3 * 2);
    rc |= cp(dst->conv0_b, src->conv0_b, ED * 4); rc |= cp(dst->conv1_b, src->conv1_b, ED * 4);
    rc |= cp(dst->adapter0, src->adapter0, DD * ED * 4 * 2); rc |= cp(dst->adapter1, src->adapter1, DD * DD * 2);
    rc |= cp(dst->enc_final_norm, src->enc_final_norm, ED * 4);
    rc |= cp(dst->hann, src->hann, MEL_NFFT * 4); rc |= cp(dst->cosT, src->cosT, (size_t)MEL_NFFT * MEL_NFREQ * 4);
    rc |= cp(dst->sinT, src->sinT, (size_t)MEL_NFFT * MEL_NFREQ * 4); rc |= cp(dst->filtT, src->filtT, (size_t)MEL_NFREQ * d.mel_bins * 4);
    for (int l = 0; l < d.enc_layers && !rc; l++) {
        EncLayer &S = src->enc[l], &D = dst->enc[l];
        rc |= cp(D.wqkv, S.wqkv, 3 * EQ * ED * 2); rc |= cp(D.wo, S.wo, ED * EQ * 2); rc |= cp(D.w13, S.w13, 2 * EH * ED * 2); rc |= cp(D.w2, S.w2, ED * EH * 2);
        rc |= cp(D.bqkv, S.bqkv, 3 * EQ * 4); rc |= cp(D.bo, S.bo, ED * 4); rc |= cp(D.b2, S.b2, ED * 4); rc |= cp(D.n1, S.n1, ED * 4); rc |= cp(D.n2, S.n2, ED * 4);
    }
    if (rc) return -1;
    HC(esync(src));
    return 0;
}

// Frames [frame0, frame0 + n) of src's device mel queue, appended to dst's queue.  Synchronous (set-up of a sharded chunk).
extern "C" int vox_hip_mel_queue_push(vox_hip_engine_t *src, vox_hip_engine_t *dst, int frame0, int n) {
    if (!src || !dst || n <= 0 || frame0 < 0 || frame0 + n > src->mel_q) { g_err = "vox_hip_mel_queue_push: frames not queued"; return -1; }
    const int MB = src->d.mel_bins;
    HC(hipSetDevice(dst->device));
    if (ensure_keep(dst, dst->conv_in0, (size_t)(2 + dst->mel_q + n) * MB * 4, (size_t)(2 + dst->mel_q) * MB * 4)) return -1;
    HC(hipSetDevice(src->device));
    HC(esync(src));
    if (peer_copy_async(dst, (float *)dst->conv_in0.p + (size_t)(2 + dst->mel_q) * MB, src, (const float *)src->conv_in0.p + (size_t)(2 + frame0) * MB,
                        (size_t)n * MB * 4, src->stream)) return -1;
    HC(esync(src));
    dst->mel_q += n;
    return 0;
}

// Drop the first n queued mel frames (they were handed to other engines).
extern "C" int vox_hip_mel_queue_drop(vox_hip_engine_t *e, int n) {
    if (!e || n < 0 || n > e->mel_q) return -1;
    if (n == 0) return 0;
    HC(hipSetDevice(e->device));
    const int MB = e->d.mel_bins, left = e->mel_q - n;
    float *in0 = (float *)e->conv_in0.p;
    if (left > 0) {
        if (ensure(e, e->stmp_in, (size_t)left * MB * 4)) return -1;
        HC(hipMemcpyAsync(e->stmp_in.p, in0 + (size_t)(2 + n) * MB, (size_t)left * MB * 4, hipMemcpyDeviceToDevice, e->stream));
        HC(hipMemcpyAsync(in0 + (size_t)2 * MB, e->stmp_in.p, (size_t)left * MB * 4, hipMemcpyDeviceToDevice, e->stream));
    }
    e->mel_q = left;
    return 0;
}

// Layer-l K/V rows of positions [pos_first, pos_first + n) from src's ring straight into the same slots of dst's ring,
// behind src's layer-l kernels; dst's stream waits for them before whatever is enqueued next on it.
extern "C" int vox_hip_shard_kv_push(vox_hip_engine_t *src, vox_hip_engine_t *dst, int layer, int pos_first, int n) {
    if (!src || !dst || layer < 0 || layer >= src->d.enc_layers || n <= 0 || src->enc_ring_cap != dst->enc_ring_cap) return -1;
    const int kvd = src->enc_qd, cap = src->enc_ring_cap;
    if (n > cap) { g_err = "vox_hip_shard_kv_push: more rows than the ring holds"; return -1; }
    HC(hipSetDevice(src->device));
    for (int i = 0; i < n;) {
        const int slot = (pos_first + i) % cap;
        const int run = std::min(n - i, cap - slot);
        if (peer_copy_async(dst, dst->enc[layer].kring + (size_t)slot * kvd, src, src->enc[layer].kring + (size_t)slot * kvd, (size_t)run * kvd * 4, src->stream)) return -1;
        if (peer_copy_async(dst, dst->enc[layer].vring + (size_t)slot * kvd, src, src->enc[layer].vring + (size_t)slot * kvd, (size_t)run * kvd * 4, src->stream)) return -1;
        i += run;
    }
    return chain_streams(src, dst);
}

// Make room for n_rows more adapter rows on the owner and account for them (their contents arrive through
// vox_hip_shard_end_push, stream-ordered).  Returns the logical index of the first new row, or -1.
extern "C" int64_t vox_hip_adapter_extend(vox_hip_engine_t *e, int n_rows) {
    if (!e || n_rows <= 0) return -1;
    if (hipSetDevice(e->device) != hipSuccess) return -1;
    if (adapter_reserve(e, n_rows)) return -1;
    const int64_t first = e->adapter_total;
    e->adapter_total += n_rows;
    return first;
}

// Final norm + adapter of the shard in flight on src; the rows land in owner's adapter buffer at logical row first_row.
extern "C" int vox_hip_shard_end_push(vox_hip_engine_t *src, vox_hip_engine_t *owner, int64_t first_row) {
    if (!src || !owner || !src->shard_x) return -1;
    HC(hipSetDevice(src->device));
    const int n = src->shard_n, ED = src->d.enc_dim, DD = src->d.dec_dim;
    if (n % 4) { g_err = "vox_hip_shard_end_push: shard rows must be a multiple of 4"; return -1; }
    const int m = n / 4;
    if (first_row < owner->adapter_row0 || first_row + m > owner->adapter_total) { g_err = "vox_hip_shard_end_push: rows not reserved"; return -1; }
    if (ensure(src, src->stmp_out, (size_t)n * ED * 4)) return -1;
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(n), dim3(256), 0, src->stream, (float *)src->stmp_out.p, ED, src->shard_x, ED,
                       src->enc_final_norm, (const float *)nullptr, ED, src->d.enc_eps);
    float *dst_rows = owner->adapter + (size_t)(first_row - owner->adapter_row0) * DD;
    if (src == owner) {
        if (adapter_dev(src, (const float *)src->stmp_out.p, m, dst_rows)) return -1;
    } else {
        if (ensure(src, src->stmp_in, (size_t)m * DD * 4)) return -1;
        if (adapter_dev(src, (const float *)src->stmp_out.p, m, (float *)src->stmp_in.p)) return -1;
        if (peer_copy_async(owner, dst_rows, src, src->stmp_in.p, (size_t)m * DD * 4, src->stream)) return -1;
        // the owner's stream does NOT wait here: its decoder waits for these rows when it gets to them (row_fences)
        const bool no_overlap = vox_disabled("multi_overlap");                          // the round-3 behaviour (read per call, like host/vox_stream.c)
        if (no_overlap) { if (chain_streams(src, owner)) return -1; }
        else {
            hipEvent_t ev;
            if (record_xev(src, &ev)) return -1;
            owner->row_fences.push_back({first_row, ev});
        }
    }
    src->enc_pos += n;                       // like a streaming chunk: the engine now stands behind its shard
    src->shard_x = nullptr; src->shard_n = 0;
    return m;
}

// Hand the streaming encoder state (KV rings of every layer, conv-stem history rows, 4x-alignment rows, position) of
// src to dst, so that dst continues the stream where src's shard ended.  Stream-ordered.
extern "C" int vox_hip_encoder_state_push(vox_hip_engine_t *src, vox_hip_engine_t *dst) {
    if (!src || !dst || src->enc_ring_cap != dst->enc_ring_cap) return -1;
    if (src == dst) return 0;
    const int MB = src->d.mel_bins, ED = src->d.enc_dim;
    const size_t ring_bytes = (size_t)src->enc_ring_cap * src->enc_qd * 4;
    HC(hipSetDevice(src->device));
    for (int l = 0; l < src->d.enc_layers; l++) {
        if (peer_copy_async(dst, dst->enc[l].kring, src, src->enc[l].kring, ring_bytes, src->stream)) return -1;
        if (peer_copy_async(dst, dst->enc[l].vring, src, src->enc[l].vring, ring_bytes, src->stream)) return -1;
    }
    if (peer_copy_async(dst, dst->conv_in0.p, src, src->conv_in0.p, (size_t)2 * MB * 4, src->stream)) return -1;
    if (peer_copy_async(dst, dst->conv_in1.p, src, src->conv_in1.p, (size_t)2 * ED * 4, src->stream)) return -1;
    if (peer_copy_async(dst, dst->enc_out.p, src, src->enc_out.p, (size_t)3 * ED * 4, src->stream)) return -1;
    dst->enc_pos = src->enc_pos; dst->c0_carry = src->c0_carry; dst->enc_res = src->enc_res;
    const bool no_overlap = vox_disabled("multi_overlap");
    if (no_overlap) return chain_streams(src, dst);
    hipEvent_t ev;
    if (record_xev(src, &ev)) return -1;
    dst->enc_fences.push_back(ev);           // waited for in front of dst's next encoder-side work, not by its decoder
    return 0;
}
