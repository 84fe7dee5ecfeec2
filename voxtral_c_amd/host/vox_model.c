/* vox_model.c — model lifetime: checkpoint -> HBM, time conditioning, stage-level API.
 *
 * Replaces vox_load / vox_free / vox_set_delay (reference voxtral.c:116-349,1629-1635),
 * the weight binders vox_encoder_load / vox_decoder_load (voxtral_encoder.c:50-117,
 * voxtral_decoder.c:49-108) and the exported forwards (voxtral.h:309-328).  Geometry is
 * read from the tensor shapes, so the reduced oracle models load with the same binary.
 */
#include "vox_internal.h"
#include "vox_safetensors.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int vox_verbose = 0;
int vox_monitor = 0;

#define ENC_PFX "mm_streams_embeddings.embedding_module.whisper_encoder"
#define EMB_PFX "mm_streams_embeddings.embedding_module"

static const vox_st_tensor_t *need(const vox_st_file_t *sf, const char *name) {
    const vox_st_tensor_t *t = vox_st_find(sf, name);
    if (!t) fprintf(stderr, "vox_load: weight not found: %s\n", name);
    return t;
}

/* Upload a BF16 tensor byte for byte; *view (optional) = the tensor inside the mmap, as the reference's
 * load_bf16_direct leaves it (voxtral_encoder.c:42-48). */
static int up_bf16(vox_ctx_t *ctx, int slot, int layer, const char *name, uint16_t **view) {
    const vox_st_tensor_t *t = need((const vox_st_file_t *)ctx->safetensors, name);
    if (!t) return -1;
    if (t->dtype != VOX_ST_BF16) { fprintf(stderr, "vox_load: %s is not BF16\n", name); return -1; }
    if (view) *view = (uint16_t *)t->data;
    return vox_hip_upload_bf16((vox_hip_engine_t *)ctx->engine, slot, layer, (const uint16_t *)t->data,
                               (size_t)vox_st_numel(t));
}

/* the ctx owns the f32 conversions it exposes through the reference's weight views */
static int own_f32(vox_ctx_t *ctx, float *v) {
    if (ctx->n_owned_f32 == ctx->cap_owned_f32) {
        const int nc = ctx->cap_owned_f32 ? ctx->cap_owned_f32 * 2 : 256;
        float **t = (float **)realloc(ctx->owned_f32, (size_t)nc * sizeof(float *));
        if (!t) return -1;
        ctx->owned_f32 = t; ctx->cap_owned_f32 = nc;
    }
    ctx->owned_f32[ctx->n_owned_f32++] = v;
    return 0;
}

/* Convert to f32 as the reference's load_f32 does (voxtral_encoder.c:32-40), upload, and keep the conversion as the
 * host view *view (owned by the ctx) - or drop it when no view is asked for. */
static int up_f32(vox_ctx_t *ctx, int slot, int layer, const char *name, float **view) {
    const vox_st_tensor_t *t = need((const vox_st_file_t *)ctx->safetensors, name);
    if (!t) return -1;
    float *v = vox_st_to_f32(t);
    if (!v) return -1;
    int rc = vox_hip_upload_f32((vox_hip_engine_t *)ctx->engine, slot, layer, v, (size_t)vox_st_numel(t));
    if (!view) free(v);
    else if (own_f32(ctx, v) == 0) *view = v;
    else { free(v); fprintf(stderr, "vox_load: out of memory keeping the f32 view of %s\n", name); rc = -1; }   /* a NULL view in a "loaded" ctx would be a trap */
    return rc;
}

/* f32 view of a tensor the engine takes as bf16 (the conv weights: the reference converts them at load,
 * voxtral_encoder.c:56-63; identical values). */
static int view_f32(vox_ctx_t *ctx, const char *name, float **view) {
    const vox_st_tensor_t *t = vox_st_find((const vox_st_file_t *)ctx->safetensors, name);
    float *v = t ? vox_st_to_f32(t) : NULL;
    if (v && own_f32(ctx, v) == 0) { *view = v; return 0; }
    free(v);
    fprintf(stderr, "vox_load: cannot build the f32 view of %s\n", name);
    return -1;              /* the header promises these views filled: fail the load rather than leave a hole */
}

/* ---- time conditioning (reference voxtral.c:31-80) --------------------------------
 * t_cond = [cos(t f_i), sin(t f_i)], f_i = exp(-ln(1e4) i / (D/2)); per layer
 * ada_scale = W_up . gelu(W_down . t_cond).  A few kFLOP once per vox_set_delay: host. */
static float gelu_tanh_host(float v) {
    const float c = v * v * v;
    return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * c)));
}

static int update_time_conditioning(vox_ctx_t *ctx) {
    const int D = ctx->dims.dec_dim, A = ctx->dims.ada_dim, L = ctx->dims.dec_layers, half = D / 2;
    const float log_theta = logf(10000.0f);
    const float t = (float)ctx->delay_tokens;
    for (int i = 0; i < half; i++) {
        const float e = t * expf(-log_theta * (float)i / (float)half);
        ctx->t_cond[i] = cosf(e);
        ctx->t_cond[i + half] = sinf(e);
    }
    float *hid = (float *)malloc((size_t)A * sizeof(float));
    if (!hid) { fprintf(stderr, "vox: out of memory in time conditioning\n"); return -1; }
    for (int l = 0; l < L; l++) {
        for (int i = 0; i < A; i++) {
            const float *row = ctx->ada_down[l] + (size_t)i * D;
            float s = 0.0f;
            for (int j = 0; j < D; j++) s += row[j] * ctx->t_cond[j];
            hid[i] = gelu_tanh_host(s);
        }
        float *scale = ctx->ada_scale + (size_t)l * D;
        for (int i = 0; i < D; i++) {
            const float *row = ctx->ada_up[l] + (size_t)i * A;
            float s = 0.0f;
            for (int j = 0; j < A; j++) s += row[j] * hid[j];
            scale[i] = s;
        }
        if (vox_hip_upload_f32((vox_hip_engine_t *)ctx->engine, VOXT_DEC_ADA_SCALE, l, scale, (size_t)D) != 0) {
            free(hid);
            return -1;
        }
    }
    free(hid);
    return 0;
}

static int count_layers(const vox_st_file_t *sf, const char *fmt) {
    char name[384];
    int n = 0;
    for (;; n++) {
        snprintf(name, sizeof name, fmt, n);
        if (!vox_st_find(sf, name)) break;
    }
    return n;
}

static int discover_dims(const vox_st_file_t *sf, vox_model_dims_t *d) {
    char nm[384];
    const vox_st_tensor_t *t;
    memset(d, 0, sizeof *d);
    d->enc_head_dim = VOX_ENC_HEAD_DIM;
    d->dec_head_dim = VOX_DEC_HEAD_DIM;
    if (!(t = need(sf, ENC_PFX ".conv_layers.0.conv.weight")) || t->ndim != 3) return -1;
    d->enc_dim = (int)t->shape[0]; d->mel_bins = (int)t->shape[1];
    d->enc_layers = count_layers(sf, ENC_PFX ".transformer.layers.%d.attention.wq.weight");
    snprintf(nm, sizeof nm, ENC_PFX ".transformer.layers.0.attention.wq.weight");
    if (!(t = need(sf, nm))) return -1;
    d->enc_heads = (int)t->shape[0] / d->enc_head_dim;
    snprintf(nm, sizeof nm, ENC_PFX ".transformer.layers.0.feed_forward.w1.weight");
    if (!(t = need(sf, nm))) return -1;
    d->enc_hidden = (int)t->shape[0];
    if (!(t = need(sf, EMB_PFX ".tok_embeddings.weight")) || t->ndim != 2) return -1;
    d->vocab = (int)t->shape[0]; d->dec_dim = (int)t->shape[1];
    d->dec_layers = count_layers(sf, "layers.%d.attention.wq.weight");
    if (!(t = need(sf, "layers.0.attention.wq.weight"))) return -1;
    d->dec_heads = (int)t->shape[0] / d->dec_head_dim;
    if (!(t = need(sf, "layers.0.attention.wk.weight"))) return -1;
    d->dec_kv_heads = (int)t->shape[0] / d->dec_head_dim;
    if (!(t = need(sf, "layers.0.feed_forward.w1.weight"))) return -1;
    d->dec_hidden = (int)t->shape[0];
    if (!(t = need(sf, "layers.0.ada_rms_norm_t_cond.0.weight"))) return -1;
    d->ada_dim = (int)t->shape[0];
    d->enc_window = VOX_ENC_WINDOW;
    d->dec_window = VOX_DEC_WINDOW;
    if (d->mel_bins != VOX_MEL_BINS || d->enc_layers <= 0 || d->dec_layers <= 0) {
        fprintf(stderr, "vox_load: unexpected checkpoint geometry\n");
        return -1;
    }
    /* the header is user data: refuse geometries the engine cannot mean (a hostile or truncated
     * file must fail here, not in an allocation of 2^60 bytes) */
    const int lim[][3] = {
        {d->enc_dim, 64, 1 << 16}, {d->enc_heads, 1, 1024}, {d->enc_hidden, 64, 1 << 18}, {d->enc_layers, 1, 1024},
        {d->dec_dim, 64, 1 << 16}, {d->dec_heads, 1, 1024}, {d->dec_kv_heads, 1, 1024}, {d->dec_hidden, 64, 1 << 18},
        {d->dec_layers, 1, 1024}, {d->vocab, 1000 + 2, 1 << 22}, {d->ada_dim, 1, 1 << 16},
    };
    for (size_t i = 0; i < sizeof lim / sizeof lim[0]; i++)
        if (lim[i][0] < lim[i][1] || lim[i][0] > lim[i][2]) {
            fprintf(stderr, "vox_load: checkpoint geometry out of range (field %d = %d)\n", (int)i, lim[i][0]);
            return -1;
        }
    if (d->enc_layers > VOX_ENC_LAYERS || d->dec_layers > VOX_DEC_LAYERS || d->dec_dim > VOX_DEC_DIM) {
        fprintf(stderr, "vox_load: checkpoint deeper / wider than the vox_ctx_t weight views (%d/%d layers, dec_dim %d)\n",
                d->enc_layers, d->dec_layers, d->dec_dim);
        return -1;
    }
    if (d->dec_heads % d->dec_kv_heads != 0 || d->enc_dim % 8 != 0 || d->dec_dim % 8 != 0 || d->enc_hidden % 8 != 0 ||
        d->dec_hidden % 8 != 0) {
        fprintf(stderr, "vox_load: checkpoint geometry not supported (head grouping / alignment)\n");
        return -1;
    }
    return 0;
}

vox_ctx_t *vox_load(const char *model_dir) { return vox_load_ex(model_dir, NULL); }

vox_ctx_t *vox_load_ex(const char *model_dir, const vox_load_opts_t *opts) {
    if (!model_dir) return NULL;
    vox_ctx_t *ctx = (vox_ctx_t *)calloc(1, sizeof *ctx);
    if (!ctx) return NULL;
    strncpy(ctx->model_dir, model_dir, sizeof(ctx->model_dir) - 1);
    ctx->delay_tokens = 6;
    ctx->use_bf16 = 1;

    char path[1024];
    snprintf(path, sizeof path, "%s/consolidated.safetensors", model_dir);
    if (vox_verbose >= 2) fprintf(stderr, "Loading model from %s\n", path);
    vox_st_file_t *sf = vox_st_open(path);
    if (!sf) { fprintf(stderr, "vox_load: cannot open %s\n", path); free(ctx); return NULL; }
    ctx->safetensors = sf;
    if (discover_dims(sf, &ctx->dims) != 0) { vox_free(ctx); return NULL; }

    vox_model_dims_t *d = &ctx->dims;
    const char *ev;
    ctx->device = opts ? opts->device : 0;
    if (opts && opts->enc_window > 0) d->enc_window = opts->enc_window;
    if (opts && opts->dec_window > 0) d->dec_window = opts->dec_window;
    if ((ev = getenv("VOX_DEVICE"))) ctx->device = atoi(ev);
    if ((ev = getenv("VOX_ENC_WINDOW")) && atoi(ev) > 0) d->enc_window = atoi(ev);
    if ((ev = getenv("VOX_DEC_WINDOW")) && atoi(ev) > 0) d->dec_window = atoi(ev);

    vox_hip_dims_t hd;
    memset(&hd, 0, sizeof hd);
    hd.mel_bins = d->mel_bins;
    hd.enc_dim = d->enc_dim; hd.enc_layers = d->enc_layers; hd.enc_heads = d->enc_heads;
    hd.enc_head_dim = d->enc_head_dim; hd.enc_hidden = d->enc_hidden; hd.enc_window = d->enc_window;
    hd.dec_dim = d->dec_dim; hd.dec_layers = d->dec_layers; hd.dec_heads = d->dec_heads;
    hd.dec_kv_heads = d->dec_kv_heads; hd.dec_head_dim = d->dec_head_dim; hd.dec_hidden = d->dec_hidden;
    hd.dec_window = d->dec_window; hd.vocab = d->vocab; hd.ada_dim = d->ada_dim;
    hd.enc_eps = VOX_ENC_NORM_EPS; hd.dec_eps = VOX_DEC_NORM_EPS; hd.rope_theta = VOX_ROPE_THETA;
    /* devices of a multi-GPU model: opts, overridden by VOX_DEVICES=0,1,... (first entry = the stream's device) */
    int devs[VOX_MAX_DEVICES], n_dev = 0;
    if (opts && opts->n_devices > 1)
        for (int i = 0; i < opts->n_devices && i < VOX_MAX_DEVICES; i++) devs[n_dev++] = opts->devices[i];
    if ((ev = getenv("VOX_DEVICES")) && *ev) {
        n_dev = 0;
        for (const char *p = ev; *p && n_dev < VOX_MAX_DEVICES;) {
            devs[n_dev++] = (int)strtol(p, (char **)&p, 10);
            while (*p == ',' || *p == ' ') p++;
        }
    }
    if (n_dev > 1) ctx->device = devs[0];
    ctx->engine = vox_hip_engine_create(ctx->device, &hd);
    if (!ctx->engine) {
        fprintf(stderr, "vox_load: cannot create the HIP engine (%s)\n", vox_hip_last_error());
        vox_free(ctx);
        return NULL;
    }
    if (vox_verbose >= 1) fprintf(stderr, "Loading weights...\n");

    int rc = 0;
    char nm[384];
    rc |= up_bf16(ctx, VOXT_TOK_EMB, 0, EMB_PFX ".tok_embeddings.weight", &ctx->decoder.tok_embeddings_bf16);
    rc |= up_bf16(ctx, VOXT_CONV0_W, 0, ENC_PFX ".conv_layers.0.conv.weight", NULL);
    rc |= up_f32(ctx, VOXT_CONV0_B, 0, ENC_PFX ".conv_layers.0.conv.bias", &ctx->encoder.conv0_bias);
    rc |= up_bf16(ctx, VOXT_CONV1_W, 0, ENC_PFX ".conv_layers.1.conv.weight", NULL);
    rc |= up_f32(ctx, VOXT_CONV1_B, 0, ENC_PFX ".conv_layers.1.conv.bias", &ctx->encoder.conv1_bias);
    rc |= view_f32(ctx, ENC_PFX ".conv_layers.0.conv.weight", &ctx->encoder.conv0_weight);
    rc |= view_f32(ctx, ENC_PFX ".conv_layers.1.conv.weight", &ctx->encoder.conv1_weight);
    for (int i = 0; i < d->enc_layers && !rc; i++) {
        vox_enc_layer_t *Lv = &ctx->encoder.layers[i];
#define EL(sfx) (snprintf(nm, sizeof nm, ENC_PFX ".transformer.layers.%d." sfx, i), nm)
        rc |= up_bf16(ctx, VOXT_ENC_WQ, i, EL("attention.wq.weight"), &Lv->wq_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_WK, i, EL("attention.wk.weight"), &Lv->wk_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_WV, i, EL("attention.wv.weight"), &Lv->wv_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_WO, i, EL("attention.wo.weight"), &Lv->wo_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_W1, i, EL("feed_forward.w1.weight"), &Lv->w1_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_W2, i, EL("feed_forward.w2.weight"), &Lv->w2_weight_bf16);
        rc |= up_bf16(ctx, VOXT_ENC_W3, i, EL("feed_forward.w3.weight"), &Lv->w3_weight_bf16);
        rc |= up_f32(ctx, VOXT_ENC_BQ, i, EL("attention.wq.bias"), &Lv->wq_bias);
        rc |= up_f32(ctx, VOXT_ENC_BV, i, EL("attention.wv.bias"), &Lv->wv_bias);
        rc |= up_f32(ctx, VOXT_ENC_BO, i, EL("attention.wo.bias"), &Lv->wo_bias);
        rc |= up_f32(ctx, VOXT_ENC_B2, i, EL("feed_forward.w2.bias"), &Lv->w2_bias);
        rc |= up_f32(ctx, VOXT_ENC_ATTN_NORM, i, EL("attention_norm.weight"), &Lv->attention_norm);
        rc |= up_f32(ctx, VOXT_ENC_FFN_NORM, i, EL("ffn_norm.weight"), &Lv->ffn_norm);
        if (vox_verbose >= 2) fprintf(stderr, "  Encoder layer %d/%d loaded\n", i + 1, d->enc_layers);
    }
    rc |= up_f32(ctx, VOXT_ENC_FINAL_NORM, 0, ENC_PFX ".transformer.norm.weight", &ctx->encoder.norm);
    rc |= up_bf16(ctx, VOXT_ADAPTER0, 0, EMB_PFX ".audio_language_projection.0.weight", &ctx->adapter.linear0_weight_bf16);
    rc |= up_bf16(ctx, VOXT_ADAPTER1, 0, EMB_PFX ".audio_language_projection.2.weight", &ctx->adapter.linear1_weight_bf16);

    ctx->ada_down = (float **)calloc((size_t)d->dec_layers, sizeof(float *));
    ctx->ada_up = (float **)calloc((size_t)d->dec_layers, sizeof(float *));
    ctx->ada_scale = (float *)calloc((size_t)d->dec_layers * d->dec_dim, sizeof(float));
    if (!ctx->ada_down || !ctx->ada_up || !ctx->ada_scale) rc = -1;
    for (int i = 0; i < d->dec_layers && !rc; i++) {
        vox_dec_layer_t *Lv = &ctx->decoder.layers[i];
#define DL(sfx) (snprintf(nm, sizeof nm, "layers.%d." sfx, i), nm)
        rc |= up_bf16(ctx, VOXT_DEC_WQ, i, DL("attention.wq.weight"), &Lv->wq_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_WK, i, DL("attention.wk.weight"), &Lv->wk_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_WV, i, DL("attention.wv.weight"), &Lv->wv_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_WO, i, DL("attention.wo.weight"), &Lv->wo_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_W1, i, DL("feed_forward.w1.weight"), &Lv->w1_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_W2, i, DL("feed_forward.w2.weight"), &Lv->w2_weight_bf16);
        rc |= up_bf16(ctx, VOXT_DEC_W3, i, DL("feed_forward.w3.weight"), &Lv->w3_weight_bf16);
        rc |= up_f32(ctx, VOXT_DEC_ATTN_NORM, i, DL("attention_norm.weight"), &Lv->attention_norm);
        rc |= up_f32(ctx, VOXT_DEC_FFN_NORM, i, DL("ffn_norm.weight"), &Lv->ffn_norm);
        const vox_st_tensor_t *t0 = need(sf, DL("ada_rms_norm_t_cond.0.weight"));
        const vox_st_tensor_t *t2 = need(sf, DL("ada_rms_norm_t_cond.2.weight"));
        if (!t0 || !t2) { rc = -1; break; }
        Lv->ada_norm_down = ctx->ada_down[i] = vox_st_to_f32(t0);
        Lv->ada_norm_up = ctx->ada_up[i] = vox_st_to_f32(t2);
        if (!ctx->ada_down[i] || !ctx->ada_up[i]) { rc = -1; break; }
        if (vox_verbose >= 2) fprintf(stderr, "  Decoder layer %d/%d loaded\n", i + 1, d->dec_layers);
    }
    rc |= up_f32(ctx, VOXT_DEC_FINAL_NORM, 0, "norm.weight", &ctx->decoder.norm);
    if (!rc) {
        const vox_mel_tables_t *mt = vox_mel_tables();
        rc |= vox_hip_upload_mel_tables((vox_hip_engine_t *)ctx->engine, mt->filters, mt->hann, mt->dft_cos, mt->dft_sin);
    }
    if (!rc) rc |= update_time_conditioning(ctx);
    if (!rc) rc |= vox_hip_upload_done((vox_hip_engine_t *)ctx->engine);      /* the staged weight copies have reached HBM */
    if (rc) { fprintf(stderr, "vox_load: failed to load weights\n"); vox_free(ctx); return NULL; }

    ctx->shard_engines[0] = ctx->engine;
    ctx->n_shard_engines = 1;
    for (int i = 1; i < n_dev; i++) {
        /* encoder-only peers: same encoder / adapter geometry, no decoder stack, weights cloned GPU to GPU */
        vox_hip_dims_t he = hd;
        he.dec_layers = 0; he.vocab = 32;
        vox_hip_engine_t *pe = vox_hip_engine_create(devs[i], &he);
        if (!pe || vox_hip_enable_peer((vox_hip_engine_t *)ctx->engine, pe) || vox_hip_enable_peer(pe, (vox_hip_engine_t *)ctx->engine) ||
            vox_hip_clone_encoder_weights(pe, (vox_hip_engine_t *)ctx->engine)) {
            fprintf(stderr, "vox_load: cannot set up device %d for the sharded encoder (%s)\n", devs[i], vox_hip_last_error());
            if (pe) vox_hip_engine_destroy(pe);
            vox_free(ctx);
            return NULL;
        }
        for (int k = 1; k < ctx->n_shard_engines; k++) {
            vox_hip_enable_peer((vox_hip_engine_t *)ctx->shard_engines[k], pe);
            vox_hip_enable_peer(pe, (vox_hip_engine_t *)ctx->shard_engines[k]);
        }
        ctx->shard_engines[ctx->n_shard_engines++] = pe;
        if (vox_verbose >= 1) fprintf(stderr, "HIP engine: device %d joins the encoder (%.1f MB resident)\n", devs[i],
                                      (double)vox_hip_memory_used(pe) / (1024.0 * 1024.0));
    }

    ctx->kv_cache_max = 0;
    if (vox_verbose >= 1) {
        fprintf(stderr, "HIP engine: device %d, %.1f MB resident\n", ctx->device,
                (double)vox_hip_memory_used((vox_hip_engine_t *)ctx->engine) / (1024.0 * 1024.0));
        fprintf(stderr, "Model loaded.\n");
    }
    {
        int fmt = opts ? opts->weight_format : 0;
        const char *wf = getenv("VOX_WEIGHTS");
        if (wf && !strcmp(wf, "fp8")) fmt = 1;
        if (wf && !strcmp(wf, "bf16")) fmt = 0;
        if (fmt == 1) {
            if (vox_hip_quantize_decoder_fp8((vox_hip_engine_t *)ctx->engine) != 0) {
                fprintf(stderr, "vox_load: fp8 decode weights unavailable: %s\n", vox_hip_last_error());
                vox_free(ctx);
                return NULL;
            }
            if (vox_verbose >= 1) fprintf(stderr, "Decoder GEMV weights: fp8 e4m3 (per-row scale)\n");
        }
    }
    return ctx;
}

void vox_free(vox_ctx_t *ctx) {
    if (!ctx) return;
    /* the stream engine may still hold waits on events of the shard engines (a sharded chunk's adapter rows / handed-over encoder
     * state that nothing has touched since): resolve them while those engines - and their events - still exist */
    if (ctx->engine) vox_hip_sync((vox_hip_engine_t *)ctx->engine);
    for (int i = 1; i < ctx->n_shard_engines; i++)
        if (ctx->shard_engines[i]) vox_hip_sync((vox_hip_engine_t *)ctx->shard_engines[i]);
    for (int i = 1; i < ctx->n_shard_engines; i++)
        if (ctx->shard_engines[i]) vox_hip_engine_destroy((vox_hip_engine_t *)ctx->shard_engines[i]);
    if (ctx->engine) vox_hip_engine_destroy((vox_hip_engine_t *)ctx->engine);
    if (ctx->ada_down) for (int i = 0; i < ctx->dims.dec_layers; i++) free(ctx->ada_down[i]);
    if (ctx->ada_up) for (int i = 0; i < ctx->dims.dec_layers; i++) free(ctx->ada_up[i]);
    free(ctx->ada_down); free(ctx->ada_up); free(ctx->ada_scale);
    for (int i = 0; i < ctx->n_owned_f32; i++) free(ctx->owned_f32[i]);
    free(ctx->owned_f32);
    if (ctx->safetensors) vox_st_close((vox_st_file_t *)ctx->safetensors);
    if (ctx->tokenizer) vox_tokenizer_free((vox_tokenizer_t *)ctx->tokenizer);
    free(ctx);
}

void vox_set_delay(vox_ctx_t *ctx, int delay_ms) {
    if (!ctx) return;
    if (delay_ms < 80) delay_ms = 80;
    if (delay_ms > 2400) delay_ms = 2400;
    ctx->delay_tokens = delay_ms / 80;
    update_time_conditioning(ctx);
}

/* ---- engine for the engine-less public mel API -------------------------------------- */
static vox_hip_engine_t *g_mel_engine = NULL;
vox_hip_engine_t *vox_default_mel_engine(void) {
    if (g_mel_engine) return g_mel_engine;
    vox_hip_dims_t hd;
    memset(&hd, 0, sizeof hd);
    hd.mel_bins = VOX_MEL_BINS;
    hd.enc_dim = 32; hd.enc_layers = 0; hd.enc_heads = 1; hd.enc_head_dim = 64; hd.enc_hidden = 32; hd.enc_window = 64;
    hd.dec_dim = 32; hd.dec_layers = 0; hd.dec_heads = 4; hd.dec_kv_heads = 1; hd.dec_head_dim = 128; hd.dec_hidden = 32;
    hd.dec_window = 64; hd.vocab = 32; hd.ada_dim = 32;
    hd.enc_eps = hd.dec_eps = 1e-5f; hd.rope_theta = VOX_ROPE_THETA;
    const char *ev = getenv("VOX_DEVICE");
    g_mel_engine = vox_hip_engine_create(ev ? atoi(ev) : 0, &hd);
    if (g_mel_engine) {
        const vox_mel_tables_t *mt = vox_mel_tables();
        vox_hip_upload_mel_tables(g_mel_engine, mt->filters, mt->hann, mt->dft_cos, mt->dft_sin);
    }
    return g_mel_engine;
}

/* ---- stage-level API: host buffers in/out, GPU compute ------------------------------- */

/* Physical-length bookkeeping of the decoder cache, as the reference does it
 * (kv_cache_init / kv_cache_compact / kv_cache_grow, voxtral_decoder.c:171-347,615-623). */
static void kv_mirror_prefill(vox_ctx_t *ctx, int seq_len) {
    const int W = ctx->dims.dec_window;
    if (ctx->kv_cache_max == 0) ctx->kv_cache_max = W + seq_len + 1024;
    else if (ctx->kv_cache_len + seq_len > ctx->kv_cache_max) {
        const int need_rows = ctx->kv_cache_len + seq_len + 1024;
        while (ctx->kv_cache_max < need_rows) ctx->kv_cache_max *= 2;
    }
    ctx->kv_cache_len += seq_len;
}
void vox_kv_mirror_step(vox_ctx_t *ctx) {
    const int W = ctx->dims.dec_window;
    if (ctx->kv_cache_max == 0) ctx->kv_cache_max = W + 1 + 1024;
    if (ctx->kv_cache_len >= ctx->kv_cache_max) {
        if (ctx->kv_cache_len > W) {
            ctx->kv_pos_offset += ctx->kv_cache_len - W;
            ctx->kv_cache_len = W;
        }
        if (ctx->kv_cache_len >= ctx->kv_cache_max) {
            const int need_rows = ctx->kv_cache_len + 1024;
            while (ctx->kv_cache_max < need_rows) ctx->kv_cache_max *= 2;
        }
    }
    ctx->kv_cache_len += 1;
}
void vox_kv_mirror_prefill(vox_ctx_t *ctx, int seq_len) { kv_mirror_prefill(ctx, seq_len); }

void vox_enc_mirror_chunk(vox_ctx_t *ctx, int new_len) {
    const int W = ctx->dims.enc_window;
    if (ctx->enc_kv_cache_len + new_len > W && ctx->enc_kv_cache_len > W) {
        ctx->enc_kv_pos_offset += ctx->enc_kv_cache_len - W;
        ctx->enc_kv_cache_len = W;
    }
    ctx->enc_kv_cache_len += new_len;
}

float *vox_encoder_forward_incremental(vox_ctx_t *ctx, const float *x_new, int new_len, int *out_len) {
    if (out_len) *out_len = 0;
    if (!ctx || new_len <= 0) return NULL;
    /* The device positions follow the mirrored counters, so a caller that resets them
     * (as vox_stream_init does, voxtral.c:1227-1228) restarts the sequence. */
    float *out = (float *)malloc((size_t)new_len * ctx->dims.enc_dim * sizeof(float));
    if (!out) return NULL;
    if (ctx->enc_kv_cache_len == 0 && ctx->enc_kv_pos_offset == 0) vox_hip_reset_encoder((vox_hip_engine_t *)ctx->engine);
    if (vox_hip_encoder_chunk((vox_hip_engine_t *)ctx->engine, x_new, new_len, out) != 0) { free(out); return NULL; }
    vox_enc_mirror_chunk(ctx, new_len);
    if (out_len) *out_len = new_len;
    return out;
}

float *vox_adapter_forward(vox_ctx_t *ctx, const float *enc_out, int enc_seq_len, int *out_seq_len) {
    if (out_seq_len) *out_seq_len = 0;
    if (!ctx) return NULL;
    const int m = enc_seq_len / VOX_DOWNSAMPLE;
    float *out = (float *)malloc((size_t)(m > 0 ? m : 1) * ctx->dims.dec_dim * sizeof(float));
    if (!out) return NULL;
    if (m > 0 && vox_hip_adapter((vox_hip_engine_t *)ctx->engine, enc_out, enc_seq_len, out) < 0) { free(out); return NULL; }
    if (out_seq_len) *out_seq_len = m;
    return out;
}

/* Batch encoder incl. its own conv stem (reference vox_encoder_forward, voxtral_encoder.c:135).
 * Like the reference's batch conv (vox_causal_conv1d, voxtral_kernels.c:293-340) an odd number of
 * frames is right-padded with one zero frame: ceil(mel_frames / 2) rows come out. */
float *vox_encoder_forward(vox_ctx_t *ctx, const float *mel, int mel_frames, int *out_seq_len) {
    if (out_seq_len) *out_seq_len = 0;
    if (!ctx || mel_frames <= 0) return NULL;
    vox_hip_engine_t *e = (vox_hip_engine_t *)ctx->engine;
    vox_hip_reset_encoder(e);
    ctx->enc_kv_cache_len = 0; ctx->enc_kv_pos_offset = 0;
    const int rows_cap = mel_frames / 2 + 1;
    float *x = (float *)malloc((size_t)rows_cap * ctx->dims.enc_dim * sizeof(float));
    if (!x) return NULL;
    int rows = vox_hip_conv_stem(e, mel, mel_frames, x, rows_cap);
    if (rows < 0) { free(x); return NULL; }
    if (mel_frames & 1) {
        const int extra = vox_hip_conv_stem_pad_odd(e, x + (size_t)rows * ctx->dims.enc_dim);
        if (extra < 0) { free(x); return NULL; }
        rows += extra;
    }
    if (rows <= 0) { free(x); return NULL; }
    int n = 0;
    float *out = vox_encoder_forward_incremental(ctx, x, rows, &n);
    free(x);
    vox_hip_reset_encoder(e);
    ctx->enc_kv_cache_len = 0; ctx->enc_kv_pos_offset = 0;
    if (out_seq_len) *out_seq_len = n;
    return out;
}

void vox_decoder_prefill(vox_ctx_t *ctx, const float *input_embeds, int seq_len) {
    if (!ctx || seq_len <= 0) return;
    vox_hip_engine_t *e = (vox_hip_engine_t *)ctx->engine;
    if (ctx->kv_cache_len == 0 && ctx->kv_pos_offset == 0 && vox_hip_decoder_kv_len(e) != 0) {
        /* caller reset the counters (voxtral.c:1001-1002): restart the device window too,
         * keeping the adapter rows */
        extern void vox_hip_reset_decoder_kv(vox_hip_engine_t *);
        vox_hip_reset_decoder_kv(e);
    }
    if (vox_hip_decoder_prefill(e, input_embeds, seq_len) != 0) return;
    kv_mirror_prefill(ctx, seq_len);
}

int vox_decoder_forward(vox_ctx_t *ctx, const float *input_embeds, float *logits) {
    if (!ctx) return 2;
    vox_hip_engine_t *e = (vox_hip_engine_t *)ctx->engine;
    if (ctx->kv_cache_len == 0 && ctx->kv_pos_offset == 0 && vox_hip_decoder_kv_len(e) != 0) {
        extern void vox_hip_reset_decoder_kv(vox_hip_engine_t *);
        vox_hip_reset_decoder_kv(e);
    }
    const int tok = vox_hip_decoder_step(e, input_embeds, logits);
    if (tok < 0) return 2;   /* the reference reports failures as EOS (voxtral_decoder.c:621,649) */
    vox_kv_mirror_step(ctx);
    return tok;
}

/* Both windows are allocated when the engine is created. */
int vox_decoder_kv_cache_preallocate(vox_ctx_t *ctx, int max_seq) { (void)max_seq; return ctx ? 0 : -1; }
int vox_encoder_kv_cache_preallocate(vox_ctx_t *ctx, int max_pos) { (void)max_pos; return ctx ? 0 : -1; }
