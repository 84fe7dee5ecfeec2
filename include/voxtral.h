/*
 * voxtral.h — public C API of the MI355X-native Voxtral-Realtime engine.
 *
 * Drop-in for the reference library header (antirez/voxtral.c voxtral.h:217-328):
 * every function a client such as the reference CLI (main.c) links against is
 * declared here with the same name, argument meaning, ownership and error
 * convention; the reference file:line each one replaces is cited.  What differs
 * is everything behind it: weights, activations, both KV windows and the adapter
 * rows live in GPU HBM and all arithmetic runs in hand-written HIP kernels
 * (include/vox_hip.h is the device boundary).  There is no CPU compute path:
 * vox_load() fails if no gfx950 device is available.
 *
 * vox_ctx_t starts with the reference's own fields (same names, types, order: voxtral.h:154-204) and the
 * weight-view types it is built from (vox_enc_layer_t ... vox_adapter_t, voxtral.h:56-148) are declared and
 * filled, so code that names those types or reads those fields compiles and runs; the engine's own state is appended.
 */
#ifndef VOXTRAL_H
#define VOXTRAL_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Voxtral-Realtime-4B constants (reference voxtral.h:19-50) ------------------
 * The engine reads the actual geometry from the checkpoint (so reduced test models
 * load too); these are the values of the released 4B model. */
#define VOX_SAMPLE_RATE      16000
#define VOX_MEL_BINS         128
#define VOX_HOP_LENGTH       160
#define VOX_WINDOW_SIZE      400
#define VOX_FRAME_RATE       12.5f
#define VOX_LOG_MEL_MAX      1.5f
#define VOX_ENC_DIM          1280
#define VOX_ENC_LAYERS       32
#define VOX_ENC_HEADS        32
#define VOX_ENC_KV_HEADS     32
#define VOX_ENC_HEAD_DIM     64
#define VOX_ENC_HIDDEN       5120
#define VOX_ENC_WINDOW       750
#define VOX_ENC_NORM_EPS     1e-5f
#define VOX_DOWNSAMPLE       4
#define VOX_DEC_DIM          3072
#define VOX_DEC_LAYERS       26
#define VOX_DEC_HEADS        32
#define VOX_DEC_KV_HEADS     8
#define VOX_DEC_HEAD_DIM     128
#define VOX_DEC_HIDDEN       9216
#define VOX_DEC_WINDOW       8192
#define VOX_DEC_NORM_EPS     1e-5f
#define VOX_VOCAB_SIZE       131072
#define VOX_ADA_NORM_DIM     32
#define VOX_ROPE_THETA       1000000.0f
#define VOX_MAX_ALT          4

/* ---- weight views (reference voxtral.h:56-148: vox_enc_layer_t, vox_encoder_t, vox_dec_layer_t, vox_decoder_t,
 * vox_adapter_t) -----------------------------------------------------------------------------------------------------
 * Same type names, field names, field types and field order as the reference, so that code written against its
 * header - naming the types, walking ctx->decoder.layers[i], reading a bias - compiles and finds what it expects.
 * Here they are HOST VIEWS of the checkpoint, filled by vox_load: the *_bf16 pointers point into the mmap'd
 * consolidated.safetensors exactly as the reference's load_bf16_direct leaves them (voxtral_encoder.c:42-48,
 * voxtral_decoder.c:41-47), the f32 pointers are the load_f32 conversions (conv weights, biases, norms, ada MLP);
 * the f32 variants of the big matrices are NULL ("NULL if bf16").  The engine computes from its own copy in HBM:
 * writing through these pointers does not change what the GPU runs.  The arrays are sized by the 4B constants above;
 * a reduced test checkpoint fills the first dims.enc_layers / dims.dec_layers entries. */
typedef struct {
    float *wq_weight;        uint16_t *wq_weight_bf16;   /* [heads*64, enc_dim] */
    float *wq_bias;
    float *wk_weight;        uint16_t *wk_weight_bf16;   /* no bias */
    float *wv_weight;        uint16_t *wv_weight_bf16;
    float *wv_bias;
    float *wo_weight;        uint16_t *wo_weight_bf16;   /* [enc_dim, heads*64] */
    float *wo_bias;
    float *attention_norm;
    float *w1_weight;        uint16_t *w1_weight_bf16;   /* [enc_hidden, enc_dim] gate */
    float *w2_weight;        uint16_t *w2_weight_bf16;   /* [enc_dim, enc_hidden] down */
    float *w2_bias;
    float *w3_weight;        uint16_t *w3_weight_bf16;   /* [enc_hidden, enc_dim] up */
    float *ffn_norm;
} vox_enc_layer_t;

typedef struct {
    float *conv0_weight;     /* [enc_dim, mel_bins, 3] */
    float *conv0_bias;
    float *conv1_weight;     /* [enc_dim, enc_dim, 3] */
    float *conv1_bias;
    vox_enc_layer_t layers[VOX_ENC_LAYERS];
    float *norm;
} vox_encoder_t;

typedef struct {
    float *ada_norm_down;    /* [ada_dim, dec_dim] */
    float *ada_norm_up;      /* [dec_dim, ada_dim] */
    float *wq_weight;        uint16_t *wq_weight_bf16;
    float *wk_weight;        uint16_t *wk_weight_bf16;
    float *wv_weight;        uint16_t *wv_weight_bf16;
    float *wo_weight;        uint16_t *wo_weight_bf16;
    float *attention_norm;
    float *w1_weight;        uint16_t *w1_weight_bf16;
    float *w2_weight;        uint16_t *w2_weight_bf16;
    float *w3_weight;        uint16_t *w3_weight_bf16;
    float *ffn_norm;
} vox_dec_layer_t;

typedef struct {
    float *tok_embeddings;   uint16_t *tok_embeddings_bf16;   /* [vocab, dec_dim], tied LM head */
    vox_dec_layer_t layers[VOX_DEC_LAYERS];
    float *norm;
} vox_decoder_t;

typedef struct {
    float *linear0_weight;   uint16_t *linear0_weight_bf16;   /* [dec_dim, 4*enc_dim] */
    float *linear1_weight;   uint16_t *linear1_weight_bf16;   /* [dec_dim, dec_dim] */
} vox_adapter_t;

/* Geometry discovered from consolidated.safetensors at load time. */
#define VOX_MAX_DEVICES 8

typedef struct vox_model_dims {
    int mel_bins;
    int enc_dim, enc_layers, enc_heads, enc_head_dim, enc_hidden, enc_window;
    int dec_dim, dec_layers, dec_heads, dec_kv_heads, dec_head_dim, dec_hidden, dec_window;
    int vocab, ada_dim;
} vox_model_dims_t;

/* vox_ctx_t (reference voxtral.h:154-204).  The reference's fields come first, same names, types and order, so the
 * struct is layout-compatible with code compiled against the reference header; what the MI355X engine adds is
 * APPENDED.  Fields of the reference that describe host buffers this engine does not have - both KV caches and the
 * per-call scratch live in HBM - are present and NULL / 0; the cache COUNTERS are live mirrors of the reference's
 * physical-length arithmetic (its restart watchdogs key on them, voxtral.c:1147). */
typedef struct vox_ctx {
    vox_encoder_t encoder;      /* host views, see above */
    vox_adapter_t adapter;
    vox_decoder_t decoder;

    void *safetensors;          /* mmap'd checkpoint (kept open for the ctx lifetime) */
    char model_dir[512];

    float *kv_cache_k, *kv_cache_v;             /* NULL: the decoder KV window is a ring in HBM */
    uint16_t *kv_cache_k_f16, *kv_cache_v_f16;  /* NULL */
    int kv_cache_fp16;                          /* 0 */
    int kv_cache_len, kv_cache_max, kv_pos_offset;   /* mirrors (voxtral_decoder.c:171-347,615-623) */

    int delay_tokens;           /* transcription delay in 80 ms tokens, default 6 */
    float t_cond[VOX_DEC_DIM];  /* time embedding of delay_tokens (first dims.dec_dim entries) */
    float *ada_scale;           /* [dec_layers * dec_dim], also resident on the device */
    int use_bf16;               /* always 1: weights stay bf16 in HBM */

    float *enc_kv_cache_k, *enc_kv_cache_v;     /* NULL: the encoder KV window is a ring in HBM */
    int enc_kv_cache_len, enc_kv_cache_max, enc_kv_cache_is_shared, enc_kv_pos_offset;   /* len / offset: mirrors */

    int enc_inc_cap;            /* 0; the scratch pointers below are NULL (device scratch) */
    float *enc_inc_x_norm, *enc_inc_q, *enc_inc_k, *enc_inc_v;
    float *enc_inc_attn_out, *enc_inc_proj_out;
    float *enc_inc_gate, *enc_inc_up, *enc_inc_ffn_out;
    int *enc_inc_positions;
    float *enc_inc_rope_freqs;
    float *dec_x, *dec_x_norm, *dec_q, *dec_k, *dec_v;
    float *dec_attn_out, *dec_proj_out;
    float *dec_gate, *dec_up, *dec_ffn_out;
    float *dec_rope_freqs;

    /* ---- appended by this engine ------------------------------------------------------------------------------ */
    vox_model_dims_t dims;
    int device;                 /* HIP device ordinal */
    void *engine;               /* vox_hip_engine_t*  (include/vox_hip.h) */
    float **ada_down, **ada_up; /* per-layer f32 copies of the ada MLP weights (= decoder.layers[i].ada_norm_*) */
    void *tokenizer;            /* vox_tokenizer_t shared by the streams of this model (parsed once) */
    /* Extra GPUs of a multi-device model (vox_load_opts_t.devices / VOX_DEVICES=0,1,...): encoder-only engines that take
     * contiguous position ranges of a large chunk (exact context parallelism, host/vox_multi.c).  engine above is
     * shard_engines[0]'s peer on devices[0] and runs everything else (streaming chunks, prefill, decode). */
    void *shard_engines[VOX_MAX_DEVICES];
    int n_shard_engines;        /* engines taking part in a sharded chunk, including `engine` (1 = single GPU) */
    float **owned_f32; int n_owned_f32, cap_owned_f32;   /* the f32 views this ctx allocated (freed by vox_free) */
    int n_sharded_chunks;       /* chunks encoded by all shard engines together so far (statistics, tests) */
    int shard_disabled;         /* set (atomically) when a sharded chunk failed: later streams of this model run on one GPU; the
                                 * shard engines stay alive until vox_free (another stream may be using them) */
} vox_ctx_t;

/* Optional load parameters (vox_load uses the defaults; the environment variables
 * VOX_DEVICE, VOX_ENC_WINDOW, VOX_DEC_WINDOW, VOX_WEIGHTS override them). */
typedef struct vox_load_opts {
    int device;        /* HIP device ordinal, default 0 */
    int enc_window;    /* encoder sliding window, default 750 */
    int dec_window;    /* decoder sliding window, default 8192 */
    int weight_format; /* 0 = bf16 as stored (default); 1 = fp8 e4m3 copies of the decoder matrices for
                          the decode GEMVs (BASELINE config 5; env VOX_WEIGHTS=fp8) */
    int n_devices;     /* > 1: BASELINE config 4 - devices[0] runs the stream (= device above when n_devices <= 1), the others
                          join it for the encoder of a large first chunk (env VOX_DEVICES=0,1,2,...).  Entries may repeat
                          (several engines on one GPU: how a 1-GPU box tests the path). */
    int devices[VOX_MAX_DEVICES];
} vox_load_opts_t;

/* ---- model lifetime (reference voxtral.h:217-223) ------------------------------- */
vox_ctx_t *vox_load(const char *model_dir);                       /* NULL on error */
vox_ctx_t *vox_load_ex(const char *model_dir, const vox_load_opts_t *opts);
void vox_free(vox_ctx_t *ctx);
void vox_set_delay(vox_ctx_t *ctx, int delay_ms);                 /* clamps to 80..2400 */

/* ---- streaming API (reference voxtral.h:239-289) --------------------------------- */
typedef struct vox_stream vox_stream_t;

vox_stream_t *vox_stream_init(vox_ctx_t *ctx);
int  vox_stream_feed(vox_stream_t *s, const float *samples, int n_samples);   /* 0 / -1 */
int  vox_stream_finish(vox_stream_t *s);                                      /* 0 / -1 */
int  vox_stream_get(vox_stream_t *s, const char **out_tokens, int max);
void vox_stream_set_alt(vox_stream_t *s, int n_alt, float cutoff);
int  vox_stream_get_alt(vox_stream_t *s, const char **out_tokens, int max_tokens, int n_alt);
void vox_set_processing_interval(vox_stream_t *s, float seconds);
void vox_stream_set_continuous(vox_stream_t *s, int enable);
int  vox_stream_flush(vox_stream_t *s);
void vox_stream_free(vox_stream_t *s);

/* ---- convenience (reference voxtral.h:296-302); returned strings are malloc'd ---- */
char *vox_transcribe(vox_ctx_t *ctx, const char *wav_path);
char *vox_transcribe_audio(vox_ctx_t *ctx, const float *samples, int n_samples);
char *vox_transcribe_stdin(vox_ctx_t *ctx);

/* ---- stage-level functions (reference voxtral.h:309-328, "internal" but exported).
 * Host buffers in / out exactly like the reference; the work runs on the GPU. ------ */
float *vox_encoder_forward(vox_ctx_t *ctx, const float *mel, int mel_frames, int *out_seq_len);
float *vox_encoder_forward_incremental(vox_ctx_t *ctx, const float *x_new, int new_len, int *out_len);
float *vox_adapter_forward(vox_ctx_t *ctx, const float *enc_out, int enc_seq_len, int *out_seq_len);
int   vox_decoder_forward(vox_ctx_t *ctx, const float *input_embeds, float *logits);
void  vox_decoder_prefill(vox_ctx_t *ctx, const float *input_embeds, int seq_len);
int   vox_decoder_kv_cache_preallocate(vox_ctx_t *ctx, int max_seq);
int   vox_encoder_kv_cache_preallocate(vox_ctx_t *ctx, int max_pos);

/* ---- extensions (not in the reference) -------------------------------------------- */
/* Opt-in history of the engine's greedy id of every decoder step since init (tests, tooling; off by
 * default so that a long-running stream holds no growing buffer).  vox_stream_token_ids returns the
 * number of ids copied (<= max), or the number recorded when out_ids is NULL. */
void vox_stream_record_ids(vox_stream_t *s, int enable);
int vox_stream_token_ids(vox_stream_t *s, int *out_ids, int max);
/* Teacher forcing (parity tests): decoder step i (counted from vox_stream_init) carries ids[i]
 * forward - as the previous token of step i+1 and for the stream's control flow - instead of the
 * engine's own argmax, which is still what vox_stream_token_ids / the recorded logits report.
 * `ids` must stay valid while the stream is used.  Steps beyond n run freely. */
void vox_stream_force_tokens(vox_stream_t *s, const int *ids, int n);
/* Record the full logits row of each decoder step (up to max_rows) for parity tests. */
void vox_stream_record_logits(vox_stream_t *s, int max_rows);
int  vox_stream_recorded_logits(vox_stream_t *s, const float **rows_out);

#ifdef __cplusplus
}
#endif
#endif /* VOXTRAL_H */
