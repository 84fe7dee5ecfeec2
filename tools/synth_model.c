/*
 * tools/synth_model.c — deterministic synthetic Voxtral-Realtime checkpoint writer.
 *
 * There are no real weights on the build or GPU boxes (no network), so parity
 * and benchmarks run on a seeded random-init model of the *exact* architecture:
 * the same tensor names / shapes / BF16 dtype the reference loaders look up
 * (voxtral_encoder.c:50-117, voxtral_decoder.c:49-108, voxtral.c:102-110;
 * names listed in MODEL.md "Tensor Names").  Both the reference oracle and the
 * HIP engine read the same file, so any weights work for parity; the scales
 * (std = 1/sqrt(fan_in)) keep activations O(1) through all layers.
 *
 * Values are counter-based (splitmix64 of tensor-name hash + element index) so
 * the file is bit-identical on every machine and can be generated in parallel.
 *
 * usage: synth_model <out_dir> <preset: full|small|tiny|deep>[-rs] [seed]
 *   writes <out_dir>/consolidated.safetensors and <out_dir>/tekken.json
 *
 * Style "-rs" ("realistic statistics", round 5): the same geometry with the weight statistics a trained
 * checkpoint has and an iid Gaussian one lacks - see the block comment in front of rs_setup().
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct {
    int enc_dim, enc_layers, enc_heads, enc_head_dim, enc_hidden;
    int dec_dim, dec_layers, dec_heads, dec_kv_heads, dec_head_dim, dec_hidden;
    int vocab, ada_dim, mel_bins;
} dims_t;

static const dims_t PRESET_FULL  = {1280, 32, 32, 64, 5120, 3072, 26, 32, 8, 128, 9216, 131072, 32, 128};
static const dims_t PRESET_SMALL = {1280,  2, 32, 64, 5120, 3072,  2, 32, 8, 128, 9216,   4096, 32, 128};
static const dims_t PRESET_TINY  = { 256,  3,  4, 64,  512,  384,  3,  8, 2, 128,  768,   2048, 32, 128};
/* "deep": the full model's depth (32 + 26 layers) at the tiny model's width, with the real attention
 * windows: depth x context effects in seconds of CPU time (oracle/Makefile builds the matching reference). */
static const dims_t PRESET_DEEP  = { 256, 32,  4, 64,  512,  384, 26,  8, 2, 128,  768,   8192, 32, 128};

typedef struct {
    char name[200];
    int64_t shape[3];
    int ndim;
    int kind;        /* 0 = matrix N(0, std), 1 = norm weight 1+N(0,.02), 2 = bias N(0,.02) */
    float std;
    uint64_t offset; /* byte offset in data section */
    uint64_t numel;
    /* "-rs" style only */
    int stack;       /* 0 = encoder width, 1 = decoder width: which outlier-channel set applies */
    int col_out;     /* input columns of the stack's outlier channels carry the factor chan_f[] (wq, wk, w1, w3) */
    int row_out;     /* output rows of the stack's outlier channels are multiplied by RS_ROW_GAIN (wo, w2: "massive activations") */
    int norm_spike;  /* norm weight: the outlier channels carry chan_s[] instead of ~1 */
} tensor_t;

static tensor_t *g_t = NULL;
static int g_nt = 0, g_cap = 0;

static void add(const char *name, int kind, float std, int ndim, int64_t a, int64_t b, int64_t c) {
    if (g_nt == g_cap) { g_cap = g_cap ? g_cap * 2 : 1024; g_t = realloc(g_t, (size_t)g_cap * sizeof(tensor_t)); }
    tensor_t *t = &g_t[g_nt++];
    memset(t, 0, sizeof(*t));
    snprintf(t->name, sizeof(t->name), "%s", name);
    t->ndim = ndim; t->shape[0] = a; t->shape[1] = b; t->shape[2] = c;
    t->kind = kind; t->std = std;
    t->numel = (uint64_t)a * (ndim > 1 ? b : 1) * (ndim > 2 ? c : 1);
}

static uint64_t fnv1a(const char *s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; s++) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
    return h;
}
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* Irwin-Hall(4) approximate standard normal from one 64-bit hash. */
static inline float approx_normal(uint64_t h) {
    float s = (float)(h & 0xFFFF) + (float)((h >> 16) & 0xFFFF) +
              (float)((h >> 32) & 0xFFFF) + (float)((h >> 48) & 0xFFFF);
    s = s * (1.0f / 65536.0f) - 2.0f;   /* mean 0, var 1/3 */
    return s * 1.7320508f;
}
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7FFFu + lsb;
    return (uint16_t)(u >> 16);
}

/* ---------------------------------------------------------------------------------------------------
 * Style "-rs": realistic statistics.  The plain presets are iid Gaussian with a damped residual stream, which
 * cannot fail a kernel on the things trained checkpoints are known for.  This style adds, at the same geometry:
 *  (i)   heavy tails: every matrix entry is Student-t(3) (unit variance, clamped at 24 sigma) instead of Gaussian:
 *        kurtosis, single weights tens of sigma out - what block / row scales of an fp8 copy have to live with;
 *  (ii)  outlier channels: a fixed ~0.5 % of the channels of each residual stream (the same set in every layer, as
 *        in trained transformers) carry x30 - x100 columns in wq / wk / w1 / w3 and x3 - x6 spikes in the two norm
 *        weights in front of them, so a handful of terms dominates those dot products (cancellation,
 *        summation-order sensitivity, large score magnitudes); the matrices are renormalised so that the rms
 *        of q, k, gate and up stays O(1);
 *  (iii) massive activations: the rows of wo / w2 (and the adapter's output layer) that write the outlier channels
 *        are x RS_ROW_GAIN, so the residual stream itself has a few channels well above the rest and RMSNorm
 *        works on a skewed vector;
 *  (iv)  residual gains at the limit the reference itself reproduces instead of a nearly linear stream: chosen
 *        with tools/amplification.py (the reference run twice, the audio perturbed by 1e-6 relative).
 * The sweep at the full geometry (30 s night1968 clip, 386 steps; distinct greedy ids / worst-step amplification / smallest top-2
 * margin; profiles/r05_rs_checkpoint_sweep.txt):
 *      row x6, gains (0.05, 0.15), qk 1   13 ids /  74            row x6 rows drown the stream in a few channels
 *      row x6, gains (0.15, 0.30)          1 id  /  31            MORE gain does not make a random deep stack livelier: with
 *      row x6, gains (0.30, 0.50)          2 ids /  16            near-uniform attention every layer adds the same mean-V vector to
 *                                                                  every position; the encoder's output is then 0.96-correlated across
 *                                                                  time and the decoder amplifies that to 0.996 (reference taps)
 *      row x1, gains (0.05, 0.15)         48 ids / 328 / 1.1e-3
 *      row x2, gains (0.05, 0.15)         81 ids / 358 / 1.2e-3
 *      row x3, gains (0.05, 0.15)         71 ids / 209 / 3.2e-3
 *      row x2, gains (0.10, 0.25), qk 1.5 52 ids / 690 / 8.1e-4
 *      row x2, gains (0.07, 0.20), qk 1.5 98 ids / 421 / 1.7e-3    <- chosen: the liveliest, a factor 2.4 below the 1e3 at which
 *                                                                  "logits within 1e-3" stops being a property of the arithmetic
 *      (plain preset, gains (0.05, 0.15): 110 ids; without the column outliers 65 ids / 128; Gaussian instead of Student-t 38 / 113)
 * ------------------------------------------------------------------------------------------------- */
static float RS_ROW_GAIN = 2.0f;   /* SYNTH_RS_ROW */
static float RS_COL_LO = 30.0f, RS_COL_HI = 100.0f, RS_SPIKE_LO = 3.0f, RS_SPIKE_HI = 6.0f;   /* SYNTH_RS_COL / SYNTH_RS_SPIKE scale both ends */
static float RS_FNORM = 0.1f;      /* SYNTH_RS_FNORM: the two final norms damp the outlier channels (as trained final norms do) */
/* residual gains of the deep stacks in this style (tools/amplification.py; the plain presets use 0.05 / 0.15 / 1) */
#define RS_DEEP_WO 0.07f
#define RS_DEEP_W2 0.2f
#define RS_DEEP_QK 1.5f
static int g_rs = 0;
static float *g_chan_f[2], *g_chan_s[2];     /* per stack: column factor (1 = ordinary channel), norm spike (0 = none) */
static int g_chan_n[2];

static void rs_setup(const dims_t *d, uint64_t seed) {
    const int width[2] = {d->enc_dim, d->dec_dim};
    for (int st = 0; st < 2; st++) {
        int n = width[st], n_out = 0;
        g_chan_n[st] = n;
        g_chan_f[st] = malloc(sizeof(float) * n);
        g_chan_s[st] = malloc(sizeof(float) * n);
        for (int c = 0; c < n; c++) {
            uint64_t h = splitmix64(seed ^ (0xC0FFEEull + 977ull * st) ^ ((uint64_t)c << 20));
            int out = (h % 200) == 0;
            float u = (float)((h >> 16) & 0xFFFF) / 65536.0f, u2 = (float)((h >> 32) & 0xFFFF) / 65536.0f;
            g_chan_f[st][c] = out ? RS_COL_LO * powf(RS_COL_HI / RS_COL_LO, u) : 1.0f;
            g_chan_s[st][c] = out ? RS_SPIKE_LO + (RS_SPIKE_HI - RS_SPIKE_LO) * u2 : 0.0f;
            n_out += out;
        }
        if (n_out == 0) {                     /* narrow stacks: at least one outlier channel */
            int c = (int)(splitmix64(seed ^ st) % n);
            g_chan_f[st][c] = 55.0f; g_chan_s[st][c] = 4.5f;
        }
    }
}
/* variance of a dot product of a col_out matrix row with a normalised input, relative to the plain case
 * (inputs ~unit on ordinary channels; RS_ROW_GAIN x spike on the outlier channels) */
static float rs_col_boost(int st) {
    double acc = 0;
    for (int c = 0; c < g_chan_n[st]; c++) {
        double f = g_chan_f[st][c], sp = g_chan_s[st][c] > 0 ? g_chan_s[st][c] * RS_ROW_GAIN : 1.0;
        acc += f * f * sp * sp;
    }
    return (float)(acc / g_chan_n[st]);
}

typedef struct { uint16_t *dst; const tensor_t *t; uint64_t seed; uint64_t lo, hi; } job_t;

/* Student-t(3) with unit variance from two hashes: n0 / sqrt(chi2_3 / 3) / sqrt(3). */
static inline float approx_t3(uint64_t h) {
    float n0 = approx_normal(h);
    uint64_t g = splitmix64(h ^ 0x5851F42D4C957F2Dull);
    float c2 = 0;
    for (int k = 0; k < 3; k++) {
        uint32_t b = (uint32_t)(g >> (21 * k)) & 0x1FFFFF;
        float s = (float)(b & 0x7F) + (float)((b >> 7) & 0x7F) + (float)((b >> 14) & 0x7F);   /* Irwin-Hall(3) of 7-bit pieces */
        s = (s + 1.5f) * (1.0f / 128.0f) - 1.5f;                                             /* mean 0, var 1/4 */
        s *= 2.0f;
        c2 += s * s;
    }
    float t = n0 / sqrtf(c2 * (1.0f / 3.0f) + 1e-4f) * 0.57735027f;
    return t > 24.0f ? 24.0f : t < -24.0f ? -24.0f : t;
}

static void *fill_job(void *arg) {
    job_t *j = (job_t *)arg;
    const tensor_t *t = j->t;
    const int rs_mat = g_rs && t->kind == 0 && !(getenv("SYNTH_RS_GAUSS_EMB") && strstr(t->name, "tok_embeddings")) && !getenv("SYNTH_RS_GAUSS");
    const uint64_t cols = t->ndim > 1 ? (uint64_t)t->shape[1] * (t->ndim > 2 ? t->shape[2] : 1) : 1;
    const float *cf = g_rs ? g_chan_f[t->stack] : NULL, *cs = g_rs ? g_chan_s[t->stack] : NULL;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        uint64_t h = splitmix64(j->seed + i);
        float n = rs_mat ? approx_t3(h) : approx_normal(h);
        float v = (t->kind == 1) ? 1.0f + 0.02f * n : (t->kind == 3) ? -3.0f + t->std * n : t->std * n;
        if (g_rs) {
            if (t->col_out) v *= cf[i % cols];
            if (t->row_out && cs[i / cols] > 0) v *= RS_ROW_GAIN;
            if (t->norm_spike == 1 && cs[i] > 0) v *= cs[i];
            if (t->norm_spike == 2 && cs[i] > 0) v *= RS_FNORM;
        }
        j->dst[i] = f32_to_bf16_rne(v);
    }
    return NULL;
}

static void b64(const unsigned char *in, int n, char *out) {
    static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    int o = 0;
    for (int i = 0; i < n; i += 3) {
        unsigned v = in[i] << 16 | (i + 1 < n ? in[i + 1] << 8 : 0) | (i + 2 < n ? in[i + 2] : 0);
        out[o++] = T[(v >> 18) & 63]; out[o++] = T[(v >> 12) & 63];
        out[o++] = (i + 1 < n) ? T[(v >> 6) & 63] : '=';
        out[o++] = (i + 2 < n) ? T[v & 63] : '=';
    }
    out[o] = 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <out_dir> <full|small|tiny|deep>[-rs] [seed]\n", argv[0]); return 2; }
    const char *out_dir = argv[1];
    dims_t d;
    char geom[32];
    snprintf(geom, sizeof geom, "%s", argv[2]);
    char *dash = strchr(geom, '-');
    if (dash) {
        if (strcmp(dash, "-rs")) { fprintf(stderr, "unknown style %s\n", dash); return 2; }
        *dash = 0; g_rs = 1;
    }
    if (!strcmp(geom, "full")) d = PRESET_FULL;
    else if (!strcmp(geom, "small")) d = PRESET_SMALL;
    else if (!strcmp(geom, "tiny")) d = PRESET_TINY;
    else if (!strcmp(geom, "deep")) d = PRESET_DEEP;
    else { fprintf(stderr, "unknown preset %s\n", argv[2]); return 2; }
    uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 1234;
    if (getenv("SYNTH_RS_ROW")) RS_ROW_GAIN = (float)atof(getenv("SYNTH_RS_ROW"));
    if (getenv("SYNTH_RS_FNORM")) RS_FNORM = (float)atof(getenv("SYNTH_RS_FNORM"));
    if (getenv("SYNTH_RS_COL")) { float k = (float)atof(getenv("SYNTH_RS_COL")); RS_COL_LO = 0.3f * k; RS_COL_HI = k; }
    if (getenv("SYNTH_RS_SPIKE")) { float k = (float)atof(getenv("SYNTH_RS_SPIKE")); RS_SPIKE_LO = 0.5f * k; RS_SPIKE_HI = k; }
    if (g_rs) rs_setup(&d, seed);

    char nm[256];
    const char *EP = "mm_streams_embeddings.embedding_module.whisper_encoder";
    int eq = d.enc_heads * d.enc_head_dim;
    int dq = d.dec_heads * d.dec_head_dim, dkv = d.dec_kv_heads * d.dec_head_dim;
#define STD(fan) (1.0f / sqrtf((float)(fan)))
    /* Residual-branch gains and attention sharpness.  Two failure modes bracket the choice:
     *  - unit gains: a 32 + 26 layer random-init stack collapses - near-uniform attention adds the same
     *    mean-V vector to every position in every layer, the per-position signal drowns and the greedy
     *    stream degenerates (round-1 checkpoint: 2 distinct ids in 61 steps), so id parity is vacuous;
     *  - sharp attention + unit FFN gain (wo x0.3, wq x4): 356 distinct ids in 386 steps, but the stack is
     *    chaotic - a 1e-6 relative perturbation of the audio moves the reference's OWN logits by 2e-2 at
     *    the full depth, so no two summation orders can agree to 1e-3 (measured with oracle/_ref, "deep" geometry, 30 s night1968 clip).
     * In between: shallow stacks (<= 4 layers: the tiny / small presets) take wo x0.3, wq x2 (peaky
     * attention that exercises the window logic, amplification ~4e2 at worst); deep stacks take a nearly
     * linear residual stream (wo x0.05, w2 x0.15, wq x1): the argmax follows the audio through conv stem,
     * encoder and adapter (64 distinct ids in 386 steps at the deep geometry), the worst-step
     * amplification of a 1e-6 input perturbation is 9e2 (median 3e1), top-2 margins stay >= 2e-3.
     * The distinct-id count and the margin histogram of every golden are stored in the fixture
     * (tools/make_golden.py).  Overridable through the environment for tuning only. */
    const int enc_deep = d.enc_layers > 4, dec_deep = d.dec_layers > 4;
    #define GAIN_ENV(name, dflt) (getenv(name) ? (float)atof(getenv(name)) : (dflt))
    /* "-rs" style: gains at the reproducibility limit, see RS_GAINS below */
    const float enc_gain = GAIN_ENV("SYNTH_ENC_GAIN", enc_deep ? (g_rs ? RS_DEEP_W2 : 0.15f) : 1.0f);
    const float dec_gain = GAIN_ENV("SYNTH_DEC_GAIN", dec_deep ? (g_rs ? RS_DEEP_W2 : 0.15f) : 1.0f);
    const float enc_wo = GAIN_ENV("SYNTH_ENC_WO", enc_deep ? (g_rs ? RS_DEEP_WO : 0.05f) : 0.3f);
    const float dec_wo = GAIN_ENV("SYNTH_DEC_WO", dec_deep ? (g_rs ? RS_DEEP_WO : 0.05f) : 0.3f);
    const float enc_qk = GAIN_ENV("SYNTH_ENC_QK", enc_deep ? (g_rs ? RS_DEEP_QK : 1.0f) : 2.0f);
    const float dec_qk = GAIN_ENV("SYNTH_DEC_QK", dec_deep ? (g_rs ? RS_DEEP_QK : 1.0f) : 2.0f);
    /* column-outlier matrices are renormalised so that q / k / gate / up keep the plain presets' rms */
    const float ecol = g_rs ? 1.0f / sqrtf(rs_col_boost(0)) : 1.0f, dcol = g_rs ? 1.0f / sqrtf(rs_col_boost(1)) : 1.0f;
#define LAST (g_t[g_nt - 1])
#define COL(st) do { LAST.stack = (st); LAST.col_out = 1; } while (0)
#define ROW(st) do { LAST.stack = (st); LAST.row_out = 1; } while (0)
#define SPIKE(st) do { LAST.stack = (st); LAST.norm_spike = 1; } while (0)
#define DAMP(st) do { LAST.stack = (st); LAST.norm_spike = 2; } while (0)
    /* tok_embeddings: logits std ~3 (realistic range); adapter output is scaled to a
     * comparable norm below so that the previous-token feedback visibly steers the
     * greedy sequence (a constant-token sequence would make id parity vacuous). */
    add("mm_streams_embeddings.embedding_module.tok_embeddings.weight", 0, STD(d.dec_dim), 2, d.vocab, d.dec_dim, 0);
    snprintf(nm, sizeof nm, "%s.conv_layers.0.conv.weight", EP); add(nm, 0, 3.0f * STD(d.mel_bins * 3), 3, d.enc_dim, d.mel_bins, 3);
    snprintf(nm, sizeof nm, "%s.conv_layers.0.conv.bias", EP);   add(nm, 3, 0.02f, 1, d.enc_dim, 0, 0);
    snprintf(nm, sizeof nm, "%s.conv_layers.1.conv.weight", EP); add(nm, 0, 8.0f * STD(d.enc_dim * 3), 3, d.enc_dim, d.enc_dim, 3); ROW(0);
    snprintf(nm, sizeof nm, "%s.conv_layers.1.conv.bias", EP);   add(nm, 3, 0.02f, 1, d.enc_dim, 0, 0);
    for (int i = 0; i < d.enc_layers; i++) {
#define EN(sfx) snprintf(nm, sizeof nm, "%s.transformer.layers.%d." sfx, EP, i)
        EN("attention.wq.weight"); add(nm, 0, ecol * enc_qk * STD(d.enc_dim), 2, eq, d.enc_dim, 0); COL(0);
        EN("attention.wq.bias");   add(nm, 2, 0.02f, 1, eq, 0, 0);
        EN("attention.wk.weight"); add(nm, 0, ecol * STD(d.enc_dim), 2, eq, d.enc_dim, 0); COL(0);
        EN("attention.wv.weight"); add(nm, 0, STD(d.enc_dim), 2, eq, d.enc_dim, 0);
        EN("attention.wv.bias");   add(nm, 2, 0.02f, 1, eq, 0, 0);
        EN("attention.wo.weight"); add(nm, 0, enc_wo * STD(eq), 2, d.enc_dim, eq, 0); ROW(0);
        EN("attention.wo.bias");   add(nm, 2, 0.02f, 1, d.enc_dim, 0, 0);
        EN("attention_norm.weight"); add(nm, 1, 0, 1, d.enc_dim, 0, 0); SPIKE(0);
        EN("feed_forward.w1.weight"); add(nm, 0, ecol * STD(d.enc_dim), 2, d.enc_hidden, d.enc_dim, 0); COL(0);
        EN("feed_forward.w2.weight"); add(nm, 0, enc_gain * STD(d.enc_hidden), 2, d.enc_dim, d.enc_hidden, 0); ROW(0);
        EN("feed_forward.w2.bias");   add(nm, 2, 0.02f, 1, d.enc_dim, 0, 0);
        EN("feed_forward.w3.weight"); add(nm, 0, ecol * STD(d.enc_dim), 2, d.enc_hidden, d.enc_dim, 0); COL(0);
        EN("ffn_norm.weight");        add(nm, 1, 0, 1, d.enc_dim, 0, 0); SPIKE(0);
    }
    snprintf(nm, sizeof nm, "%s.transformer.norm.weight", EP); add(nm, 1, 0, 1, d.enc_dim, 0, 0); DAMP(0);
    add("mm_streams_embeddings.embedding_module.audio_language_projection.0.weight", 0, STD(d.enc_dim * 4), 2, d.dec_dim, d.enc_dim * 4, 0);
    add("mm_streams_embeddings.embedding_module.audio_language_projection.2.weight", 0, STD(d.dec_dim), 2, d.dec_dim, d.dec_dim, 0); ROW(1);
    for (int i = 0; i < d.dec_layers; i++) {
#define DN(sfx) snprintf(nm, sizeof nm, "layers.%d." sfx, i)
        DN("ada_rms_norm_t_cond.0.weight"); add(nm, 0, STD(d.dec_dim), 2, d.ada_dim, d.dec_dim, 0);
        DN("ada_rms_norm_t_cond.2.weight"); add(nm, 0, 0.1f * STD(d.ada_dim), 2, d.dec_dim, d.ada_dim, 0);
        DN("attention.wq.weight"); add(nm, 0, dcol * dec_qk * STD(d.dec_dim), 2, dq, d.dec_dim, 0); COL(1);
        DN("attention.wk.weight"); add(nm, 0, dcol * STD(d.dec_dim), 2, dkv, d.dec_dim, 0); COL(1);
        DN("attention.wv.weight"); add(nm, 0, STD(d.dec_dim), 2, dkv, d.dec_dim, 0);
        DN("attention.wo.weight"); add(nm, 0, dec_wo * STD(dq), 2, d.dec_dim, dq, 0); ROW(1);
        DN("attention_norm.weight"); add(nm, 1, 0, 1, d.dec_dim, 0, 0); SPIKE(1);
        DN("feed_forward.w1.weight"); add(nm, 0, dcol * STD(d.dec_dim), 2, d.dec_hidden, d.dec_dim, 0); COL(1);
        DN("feed_forward.w2.weight"); add(nm, 0, dec_gain * STD(d.dec_hidden), 2, d.dec_dim, d.dec_hidden, 0); ROW(1);
        DN("feed_forward.w3.weight"); add(nm, 0, dcol * STD(d.dec_dim), 2, d.dec_hidden, d.dec_dim, 0); COL(1);
        DN("ffn_norm.weight"); add(nm, 1, 0, 1, d.dec_dim, 0, 0); SPIKE(1);
    }
    add("norm.weight", 1, 0, 1, d.dec_dim, 0, 0); DAMP(1);

    /* header */
    uint64_t off = 0;
    for (int i = 0; i < g_nt; i++) { g_t[i].offset = off; off += g_t[i].numel * 2; }
    size_t hcap = (size_t)g_nt * 400 + 64, hl = 0;
    char *hdr = malloc(hcap);
    hl += snprintf(hdr + hl, hcap - hl, "{");
    for (int i = 0; i < g_nt; i++) {
        tensor_t *t = &g_t[i];
        hl += snprintf(hdr + hl, hcap - hl, "%s\"%s\":{\"dtype\":\"BF16\",\"shape\":[", i ? "," : "", t->name);
        for (int k = 0; k < t->ndim; k++) hl += snprintf(hdr + hl, hcap - hl, "%s%lld", k ? "," : "", (long long)t->shape[k]);
        hl += snprintf(hdr + hl, hcap - hl, "],\"data_offsets\":[%llu,%llu]}", (unsigned long long)t->offset,
                       (unsigned long long)(t->offset + t->numel * 2));
    }
    hl += snprintf(hdr + hl, hcap - hl, "}");
    while (hl % 8) hdr[hl++] = ' ';

    mkdir(out_dir, 0755);
    char path[1024];
    snprintf(path, sizeof path, "%s/consolidated.safetensors", out_dir);
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    uint64_t hl64 = hl;
    fwrite(&hl64, 8, 1, f);
    fwrite(hdr, 1, hl, f);

    int nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads > 32) nthreads = 32;
    if (nthreads < 1) nthreads = 1;
    for (int i = 0; i < g_nt; i++) {
        tensor_t *t = &g_t[i];
        uint16_t *buf = malloc(t->numel * 2);
        if (!buf) { fprintf(stderr, "oom\n"); return 1; }
        uint64_t tseed = splitmix64(fnv1a(t->name) ^ seed) ;
        int nt = t->numel < 65536 ? 1 : nthreads;
        pthread_t th[32]; job_t jobs[32];
        for (int k = 0; k < nt; k++) {
            jobs[k] = (job_t){buf, t, tseed, t->numel * k / nt, t->numel * (k + 1) / nt};
            if (nt == 1) fill_job(&jobs[k]); else pthread_create(&th[k], NULL, fill_job, &jobs[k]);
        }
        if (nt > 1) for (int k = 0; k < nt; k++) pthread_join(th[k], NULL);
        if (fwrite(buf, 2, t->numel, f) != t->numel) { perror("write"); return 1; }
        free(buf);
    }
    fclose(f);

    /* tekken.json (format parsed by voxtral_tokenizer.c:186-331): every vocab
     * string encodes its own id so string parity implies id parity.  Rank 0
     * (token id 1000) is the NUL byte like the real Tekken vocab (voxtral.c:487). */
    snprintf(path, sizeof path, "%s/tekken.json", out_dir);
    f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    int n_vocab = d.vocab - 1000;
    fprintf(f, "{\"config\":{\"default_vocab_size\":%d,\"default_num_special_tokens\":1000},\"vocab\":[", d.vocab);
    for (int r = 0; r < n_vocab; r++) {
        unsigned char raw[32]; char enc[64]; int n;
        if (r == 0) { raw[0] = 0; n = 1; }
        else n = snprintf((char *)raw, sizeof raw, " t%d", 1000 + r);
        b64(raw, n, enc);
        fprintf(f, "%s{\"rank\":%d,\"token_bytes\":\"%s\",\"token_str\":null}", r ? "," : "", r, enc);
    }
    fprintf(f, "],\"special_tokens\":[");
    for (int r = 0; r < 1000; r++) {
        const char *s = r == 0 ? "<unk>" : r == 1 ? "<s>" : r == 2 ? "</s>" : r == 32 ? "[STREAMING_PAD]" : NULL;
        char tmp[32];
        if (!s) { snprintf(tmp, sizeof tmp, "<SPECIAL_%d>", r); s = tmp; }
        fprintf(f, "%s{\"rank\":%d,\"token_str\":\"%s\",\"is_control\":true}", r ? "," : "", r, s);
    }
    fprintf(f, "]}\n");
    fclose(f);
    fprintf(stderr, "synth_model: %d tensors, %.1f MB -> %s\n", g_nt, (double)off / 1e6, out_dir);
    return 0;
}
