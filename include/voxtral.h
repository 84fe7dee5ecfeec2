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
 * vox_ctx_t is an engine-owned struct.  Clients only hold pointers to it (main.c
 * never dereferences it); the handful of counters the reference exposes in its
 * own struct (voxtral.h:163-190) are mirrored with the same names.
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

/* Geometry discovered from consolidated.safetensors at load time. */
#define VOX_MAX_DEVICES 8

typedef struct vox_model_dims {
    int mel_bins;
    int enc_dim, enc_layers, enc_heads, enc_head_dim, enc_hidden, enc_window;
    int dec_dim, dec_layers, dec_heads, dec_kv_heads, dec_head_dim, dec_hidden, dec_window;
    int vocab, ada_dim;
} vox_model_dims_t;

typedef struct vox_ctx {
    char model_dir[512];
    vox_model_dims_t dims;
    int device;                 /* HIP device ordinal */
    void *safetensors;          /* mmap'd checkpoint (kept open for the ctx lifetime) */
    void *engine;               /* vox_hip_engine_t*  (include/vox_hip.h) */

    int delay_tokens;           /* transcription delay in 80 ms tokens, default 6 */
    float *t_cond;              /* [dec_dim] time embedding of delay_tokens */
    float *ada_scale;           /* [dec_layers * dec_dim], also resident on the device */
    float **ada_down, **ada_up; /* per-layer f32 copies of the ada MLP weights */

    /* Mirrors of the reference's cache counters (same names, voxtral.h:169-171,186-189).
     * The device keeps both KV windows as position-indexed rings; these reproduce the
     * reference's physical-length arithmetic, which its restart watchdogs key on. */
    int kv_cache_len, kv_cache_max, kv_pos_offset;
    int enc_kv_cache_len, enc_kv_pos_offset;
    int use_bf16;               /* always 1: weights stay bf16 in HBM */
    void *tokenizer;            /* vox_tokenizer_t shared by the streams of this model (parsed once) */
    /* Extra GPUs of a multi-device model (vox_load_opts_t.devices / VOX_DEVICES=0,1,...): encoder-only engines that take
     * contiguous position ranges of a large first chunk (exact context parallelism, host/vox_multi.c).  engine above is
     * shard_engines[0]'s peer on devices[0] and runs everything else (streaming chunks, prefill, decode). */
    void *shard_engines[VOX_MAX_DEVICES];
    int n_shard_engines;        /* engines taking part in a sharded chunk, including `engine` (1 = single GPU) */
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
