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
