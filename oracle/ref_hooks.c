/*
 * oracle/ref_hooks.c — TEST INFRASTRUCTURE ONLY.
 *
 * Linked into oracle/_ref/libvoxref*.so next to the *unmodified* reference
 * objects with  -Wl,--wrap=vox_decoder_forward  so that every decoder step the
 * reference stream orchestrator performs (call sites voxtral.c:1012 and
 * voxtral.c:1063) is recorded: greedy token id + (optionally) the full logits
 * row.  The reference API never exposes token ids (strings only), so this is
 * the only way to obtain id/logit goldens from the real reference.
 *
 * Nothing here implements model arithmetic; it only observes.
 */
#include <stdlib.h>
#include <string.h>

struct vox_ctx_opaque;
int __real_vox_decoder_forward(void *ctx, const float *input_embeds, float *logits);

static int   *g_tokens = NULL;
static int    g_n = 0, g_cap = 0;
static float *g_logits = NULL;      /* [g_logit_rows_cap, g_vocab] */
static int    g_vocab = 0;          /* 0 = do not record logits */
static int    g_logit_rows = 0, g_logit_rows_cap = 0;

void voxref_hook_reset(int vocab_to_record, int max_logit_rows) {
    g_n = 0;
    g_logit_rows = 0;
    g_vocab = vocab_to_record;
    if (g_vocab > 0 && max_logit_rows > 0) {
        free(g_logits);
        g_logits = (float *)malloc((size_t)max_logit_rows * g_vocab * sizeof(float));
        g_logit_rows_cap = g_logits ? max_logit_rows : 0;
    } else {
        g_logit_rows_cap = 0;
    }
}

int voxref_hook_count(void) { return g_n; }
const int *voxref_hook_tokens(void) { return g_tokens; }
int voxref_hook_logit_rows(void) { return g_logit_rows; }
const float *voxref_hook_logits(void) { return g_logits; }

int __wrap_vox_decoder_forward(void *ctx, const float *input_embeds, float *logits) {
    int tok = __real_vox_decoder_forward(ctx, input_embeds, logits);
    if (g_n == g_cap) {
        int nc = g_cap ? g_cap * 2 : 1024;
        int *t = (int *)realloc(g_tokens, (size_t)nc * sizeof(int));
        if (t) { g_tokens = t; g_cap = nc; }
    }
    if (g_n < g_cap) g_tokens[g_n++] = tok;
    if (g_vocab > 0 && g_logit_rows < g_logit_rows_cap && logits) {
        memcpy(g_logits + (size_t)g_logit_rows * g_vocab, logits,
               (size_t)g_vocab * sizeof(float));
        g_logit_rows++;
    }
    return tok;
}

/* ---- residual-stream taps ------------------------------------------------------------------------------------------
 * -Wl,--wrap=vox_rms_norm: every RMSNorm the decoder applies to ONE row (vox_decoder_forward: attention_norm and ffn_norm
 * of each layer, then the final norm - voxtral_decoder.c:653-694) sees the residual stream as its input.  For a few chosen
 * decoder steps the inputs are recorded in call order: [2 L + 1][hidden] per step = x at the start of every layer, x after
 * every attention block, x after the last layer.  Goldens at this level pin the depth of the stack, not only its logits. */
void __real_vox_rms_norm(float *out, const float *x, const float *weight, int seq_len, int hidden, float eps);

#define TAP_MAX_STEPS 16
static int    g_tap_steps[TAP_MAX_STEPS], g_tap_n = 0, g_tap_hidden = 0, g_tap_per_step = 0;
static int    g_tap_count[TAP_MAX_STEPS];
static float *g_taps = NULL;        /* [g_tap_n][g_tap_per_step][g_tap_hidden] */

void voxref_tap_config(const int *steps, int n, int hidden, int vectors_per_step) {
    free(g_taps); g_taps = NULL; g_tap_n = 0;
    if (n <= 0 || n > TAP_MAX_STEPS || hidden <= 0 || vectors_per_step <= 0) return;
    g_taps = (float *)calloc((size_t)n * vectors_per_step * hidden, sizeof(float));
    if (!g_taps) return;
    memcpy(g_tap_steps, steps, (size_t)n * sizeof(int));
    memset(g_tap_count, 0, sizeof g_tap_count);
    g_tap_n = n; g_tap_hidden = hidden; g_tap_per_step = vectors_per_step;
}
const float *voxref_taps(void) { return g_taps; }
const int *voxref_tap_counts(void) { return g_tap_count; }

void __wrap_vox_rms_norm(float *out, const float *x, const float *weight, int seq_len, int hidden, float eps) {
    if (g_taps && seq_len == 1 && hidden == g_tap_hidden)
        for (int i = 0; i < g_tap_n; i++)
            if (g_tap_steps[i] == g_n && g_tap_count[i] < g_tap_per_step) {
                memcpy(g_taps + ((size_t)i * g_tap_per_step + g_tap_count[i]) * hidden, x, (size_t)hidden * sizeof(float));
                g_tap_count[i]++;
            }
    __real_vox_rms_norm(out, x, weight, seq_len, hidden, eps);
}
