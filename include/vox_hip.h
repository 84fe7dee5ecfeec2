/*
 * vox_hip.h — thin C-ABI boundary between the plain-C host library (libvoxtral.so)
 * and the hand-written HIP/CDNA4 device engine (libvoxhip.so, gfx950 only).
 *
 * Plain pointers and sizes only; no HIP, C++ or torch types cross this line and the
 * host .c files never include a HIP header.  The seam mirrors the only places where
 * the reference ever leaves its CPU kernels for a GPU backend (the USE_METAL hooks):
 *
 *   reference seam (file:line)                         ->  entry point here
 *   ---------------------------------------------------------------------------------
 *   vox_metal_init / shutdown          main.c:178,404      vox_hip_engine_create/destroy
 *   weight warm-up in vox_load         voxtral.c:163-235   vox_hip_upload_bf16 / _f32
 *   KV pre-allocation                  voxtral.c:237-245   (inside engine_create)
 *   vox_mel_feed / mel_compute_avail.  voxtral_audio.c:454,560   vox_hip_mel_frames
 *   stream_conv_stem                   voxtral.c:537       vox_hip_conv_stem
 *   vox_metal_encoder_full_step        voxtral_encoder.c:508-517 vox_hip_encoder_chunk
 *   vox_adapter_forward                voxtral_encoder.c:642     vox_hip_adapter
 *   4x alignment + adapter_buf append  voxtral.c:824-890   vox_hip_stream_encode
 *   vox_metal_decoder_prefill_step     voxtral_decoder.c:448-456 vox_hip_decoder_prefill
 *   vox_metal_decoder_full_step        voxtral_decoder.c:632-645 vox_hip_decoder_step
 *   prompt / step embedding build      voxtral.c:993-999,1057-1061 vox_hip_decoder_prefill_stream /
 *                                                           vox_hip_decoder_run
 *   stream_reset_*                     voxtral.c:734-780   vox_hip_reset_encoder/_decoder
 *
 * Conventions: int returns are 0 on success, -1 on error (message on stderr), like
 * the reference (voxtral.h:246-251).  All calls are synchronous on return unless the
 * name ends in _async.  One engine = one GPU = one active stream (voxtral.c:1227).
 */
#ifndef VOX_HIP_H
#define VOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vox_hip_engine vox_hip_engine_t;

/* Model geometry (voxtral.h:19-50 are the Voxtral-Realtime-4B values). */
typedef struct vox_hip_dims {
    int mel_bins;                                  /* 128 */
    int enc_dim, enc_layers, enc_heads, enc_head_dim, enc_hidden, enc_window;
    int dec_dim, dec_layers, dec_heads, dec_kv_heads, dec_head_dim, dec_hidden, dec_window;
    int vocab, ada_dim;
    float enc_eps, dec_eps, rope_theta;
} vox_hip_dims_t;

/* Tensor slots.  "layer" is ignored for the global ones. */
enum vox_hip_tensor {
    /* bf16 matrices, uploaded byte-for-byte from the mmap'd safetensors */
    VOXT_TOK_EMB = 0,      /* [vocab, dec_dim]                                  */
    VOXT_CONV0_W,          /* [enc_dim, mel_bins*3]                             */
    VOXT_CONV1_W,          /* [enc_dim, enc_dim*3]                              */
    VOXT_ENC_WQ, VOXT_ENC_WK, VOXT_ENC_WV,   /* [heads*hd, enc_dim] (merged QKV in HBM) */
    VOXT_ENC_WO,           /* [enc_dim, heads*hd]                               */
    VOXT_ENC_W1, VOXT_ENC_W3,                /* [enc_hidden, enc_dim] (merged W1;W3)    */
    VOXT_ENC_W2,           /* [enc_dim, enc_hidden]                             */
    VOXT_ADAPTER0,         /* [dec_dim, enc_dim*4]                              */
    VOXT_ADAPTER1,         /* [dec_dim, dec_dim]                                */
    VOXT_DEC_WQ, VOXT_DEC_WK, VOXT_DEC_WV,   /* merged QKV [ (h+2kvh)*hd, dec_dim ]     */
    VOXT_DEC_WO,           /* [dec_dim, heads*hd]                               */
    VOXT_DEC_W1, VOXT_DEC_W3,                /* [dec_hidden, dec_dim]                   */
    VOXT_DEC_W2,           /* [dec_dim, dec_hidden]                             */
    /* f32 vectors (the reference converts these to f32 at load: load_f32,
     * voxtral_encoder.c:32, voxtral_decoder.c:31) */
    VOXT_CONV0_B, VOXT_CONV1_B,
    VOXT_ENC_BQ, VOXT_ENC_BV, VOXT_ENC_BO, VOXT_ENC_B2,
    VOXT_ENC_ATTN_NORM, VOXT_ENC_FFN_NORM, VOXT_ENC_FINAL_NORM,
    VOXT_DEC_ATTN_NORM, VOXT_DEC_FFN_NORM, VOXT_DEC_FINAL_NORM,
    VOXT_DEC_ADA_SCALE,    /* [dec_dim] per layer: ada_up(gelu(ada_down(t_cond))), voxtral.c:47-80 */
    VOXT_COUNT
};

/* ---- lifetime ------------------------------------------------------------------ */
int  vox_hip_device_count(void);
vox_hip_engine_t *vox_hip_engine_create(int device, const vox_hip_dims_t *dims);
void vox_hip_engine_destroy(vox_hip_engine_t *e);
const char *vox_hip_last_error(void);
/* Bytes of HBM currently held by the engine. */
size_t vox_hip_memory_used(const vox_hip_engine_t *e);

/* ---- weights: host (mmap) -> HBM ----------------------------------------------- */
int vox_hip_upload_bf16(vox_hip_engine_t *e, int tensor, int layer,
                        const uint16_t *host_bf16, size_t n_elems);
int vox_hip_upload_f32(vox_hip_engine_t *e, int tensor, int layer,
                       const float *host_f32, size_t n_elems);
/* Weights >= 4 MB go through a staged, multi-threaded pinned pipeline (page cache -> pinned slots -> hipMemcpyAsync); the engine's
 * compute stream is ordered behind them.  vox_hip_upload_done waits for the last copy and releases the staging threads and buffers
 * (reference side: the mmap loader voxtral_safetensors.c:204-429 + the per-tensor warm-up voxtral.c:163-251). */
int vox_hip_upload_done(vox_hip_engine_t *e);
/* (mandatory after the last upload for direct users of this header: the staging threads and 32 MB of pinned memory live
 * until this call or vox_hip_engine_destroy.  If staging cannot be set up the engine says so once on stderr and uses
 * plain copies for the rest of its life.) */
/* Mel tables built by the host exactly as voxtral_audio.c:248-285,531-542 does:
 * filters [mel_bins,201], hann[400], dft_cos/sin [201,400]. */
int vox_hip_upload_mel_tables(vox_hip_engine_t *e, const float *filters, const float *hann,
                              const float *dft_cos, const float *dft_sin);

/* ---- stage-level entry points (host buffers in, host buffers out) --------------
 * Same argument meaning as the reference functions they replace; used by the
 * voxtral.h "internal" API and by the parity tests. */

/* Log-mel frames: frame t uses samples[t*160 .. t*160+399] (voxtral_audio.c:454-513).
 * samples must hold (n_frames-1)*160+400 floats. out_mel: [n_frames, mel_bins] or NULL
 * (frames are also appended to the engine's device mel queue when to_queue != 0). */
int vox_hip_mel_frames(vox_hip_engine_t *e, const float *samples, int n_frames,
                       float *out_mel, int to_queue);

/* vox_encoder_forward_incremental (voxtral_encoder.c:452): x_new [new_len, enc_dim]
 * post-conv-stem rows -> out [new_len, enc_dim]; advances the encoder KV window. */
int vox_hip_encoder_chunk(vox_hip_engine_t *e, const float *x_new, int new_len, float *out);

/* vox_adapter_forward (voxtral_encoder.c:642): enc_out [enc_len, enc_dim] ->
 * out [enc_len/4, dec_dim]. Returns number of adapter rows or -1. */
int vox_hip_adapter(vox_hip_engine_t *e, const float *enc_out, int enc_len, float *out);

/* stream_conv_stem (voxtral.c:537) on host mel rows [n_mel, mel_bins]; keeps the
 * boundary state on the device. out may be NULL. Returns rows produced ([rows, enc_dim]). */
int vox_hip_conv_stem(vox_hip_engine_t *e, const float *mel_new, int n_mel, float *out, int out_cap_rows);

/* Batch-conv tail (vox_causal_conv1d right padding, voxtral_kernels.c:293-340 as used by
 * vox_encoder_forward, voxtral_encoder.c:162-176): if an odd conv0 frame is waiting for its stride-2
 * partner, pair it with a zero frame and emit the last row. out_row [enc_dim]. Returns 1 / 0 / -1. */
int vox_hip_conv_stem_pad_odd(vox_hip_engine_t *e, float *out_row);

/* vox_decoder_prefill (voxtral_decoder.c:410): embeds [seq_len, dec_dim]. */
int vox_hip_decoder_prefill(vox_hip_engine_t *e, const float *embeds, int seq_len);

/* vox_decoder_forward (voxtral_decoder.c:586): embed [dec_dim] -> greedy token id
 * (lowest index wins ties); logits [vocab] may be NULL. Returns token or -1. */
int vox_hip_decoder_step(vox_hip_engine_t *e, const float *embed, float *logits);

/* ---- fused streaming path (activations never leave HBM) ------------------------
 * vox_hip_stream_encode consumes n_mel frames from the device mel queue and runs
 * conv stem -> encoder -> 4x alignment -> adapter, appending adapter rows to the
 * device adapter buffer (stream_run_encoder, voxtral.c:783-907). Outputs the counters
 * the host state machine needs. Returns number of new adapter rows or -1. */
int vox_hip_stream_encode(vox_hip_engine_t *e, int n_mel, int *conv_rows, int *enc_residual);

/* Prompt prefill straight from the adapter buffer (voxtral.c:990-1012):
 * embeds[i] = adapter[first_row+i] + tok_emb[i==0 ? bos : pad], i < n_prompt;
 * prefill n_prompt-1 rows, then one step. Returns the first generated token. */
int vox_hip_decoder_prefill_stream(vox_hip_engine_t *e, int64_t first_row, int n_prompt,
                                   int bos_token, int pad_token, float *logits);

/* Greedy loop (voxtral.c:1056-1093): for i < n_steps: embed = adapter[first_row+i] +
 * tok_emb[prev]; prev = step(embed). The previous token lives on the device, so all
 * n_steps are enqueued back-to-back with one synchronisation at the end.
 * tokens_out[n_steps]; stops early after eos_token (< 0: never). logits_out is
 * [n_steps, vocab] or NULL. Returns number of steps actually taken (<= n_steps). */
int vox_hip_decoder_run(vox_hip_engine_t *e, int64_t first_row, int n_steps, int prev_token,
                        int eos_token, int *tokens_out, float *logits_out);

/* Copy rows out of the device adapter buffer (tests / multi-GPU gather). */
int vox_hip_adapter_read(vox_hip_engine_t *e, int64_t first_row, int n_rows, float *out);
/* Append externally produced adapter rows (multi-GPU: rows gathered over xGMI). */
int vox_hip_adapter_append(vox_hip_engine_t *e, const float *rows, int n_rows);
int64_t vox_hip_adapter_rows(const vox_hip_engine_t *e);
/* Device pointer of the adapter ring (for RCCL/torch interop) and its row capacity. */
void *vox_hip_adapter_devptr(vox_hip_engine_t *e, int64_t *cap_rows);

/* ---- multi-GPU encoder sharding (exact context parallelism) ------
 * A rank owns encoder positions [pos0, pos0+n). Per layer it imports the layer's K/V of the
 * window-1 positions before pos0 (received from its left neighbour over xGMI), runs the
 * layer, and exports the K/V tail its right neighbour needs. Pointers are device pointers. */
int vox_hip_shard_begin(vox_hip_engine_t *e, int n_mel, int discard_rows, int pos0);  /* conv stem on the queue; returns rows */
int vox_hip_shard_layer(vox_hip_engine_t *e, int layer);
int vox_hip_shard_kv_export(vox_hip_engine_t *e, int layer, int pos_first, int n, void *dst_dev);       /* [2][n][kv] */
int vox_hip_shard_kv_import(vox_hip_engine_t *e, int layer, int pos_first, int n, const void *src_dev);
int vox_hip_shard_end(vox_hip_engine_t *e, void *adapter_rows_dev);                  /* returns adapter rows written */
/* Stream-ordered variants: nothing below waits on the host.  A transport that is ordered behind the engine stream (RCCL
 * issued on vox_hip_stream_handle through torch.cuda.ExternalStream, or an event) needs no host round trip per layer;
 * vox_hip_host_syncs counts the host-side waits on the engine stream so that tests can assert there were none between
 * vox_hip_shard_begin and vox_hip_shard_end_async.  The plain names above = the async call + one synchronisation. */
int vox_hip_shard_kv_export_async(vox_hip_engine_t *e, int layer, int pos_first, int n, void *dst_dev);
int vox_hip_shard_end_async(vox_hip_engine_t *e, void *adapter_rows_dev);
int vox_hip_adapter_append_dev_async(vox_hip_engine_t *e, const void *rows_dev, int n_rows);
void vox_hip_reset_encoder_async(vox_hip_engine_t *e);
void *vox_hip_stream_handle(vox_hip_engine_t *e);                   /* the engine's hipStream_t */
unsigned long long vox_hip_host_syncs(const vox_hip_engine_t *e);
int vox_hip_adapter_append_dev(vox_hip_engine_t *e, const void *rows_dev, int n_rows);
void *vox_hip_device_alloc(vox_hip_engine_t *e, size_t bytes);
void vox_hip_device_free(vox_hip_engine_t *e, void *p);
int vox_hip_memcpy(vox_hip_engine_t *e, void *dst, const void *src, size_t bytes, int kind); /* 0 H2D, 1 D2H, 2 D2D */

/* ---- several GPUs in one process (libvoxtral VOX_DEVICES): stream-ordered peer hand-offs ----------
 * A hand-off = peer copy on the producer's stream + event + hipStreamWaitEvent on the consumer's stream: the
 * host only enqueues, there is no host synchronisation per layer.  Both engines may sit on one device (tests). */
int vox_hip_enable_peer(vox_hip_engine_t *a, vox_hip_engine_t *b);
int vox_hip_clone_encoder_weights(vox_hip_engine_t *dst, vox_hip_engine_t *src);           /* load time, synchronous */
int vox_hip_mel_queue_push(vox_hip_engine_t *src, vox_hip_engine_t *dst, int frame0, int n); /* synchronous (chunk set-up) */
int vox_hip_mel_queue_drop(vox_hip_engine_t *e, int n);
int vox_hip_shard_kv_push(vox_hip_engine_t *src, vox_hip_engine_t *dst, int layer, int pos_first, int n);
int64_t vox_hip_adapter_extend(vox_hip_engine_t *e, int n_rows);                           /* first new logical row or -1 */
int vox_hip_shard_end_push(vox_hip_engine_t *src, vox_hip_engine_t *owner, int64_t first_row);
int vox_hip_encoder_state_push(vox_hip_engine_t *src, vox_hip_engine_t *dst);
/* Round 4.  vox_hip_shard_end_push and vox_hip_encoder_state_push no longer make the owner's stream wait for the other engine
 * at once: the owner's DECODER waits (on its stream) for a shard's adapter rows right in front of the first step that reads
 * them, its ENCODER side waits for the handed-over state before it touches encoder state again - so decoding starts on the first
 * shard's rows while later shards still encode.  VOX_HIP_DISABLE=multi_overlap restores the immediate waits.
 * vox_hip_encoder_aligned: 1 if the stream's encoder state sits on a token boundary (a sharded chunk may start from it);
 * vox_hip_encoder_pos: encoder positions done so far; vox_hip_pending_fences: waits not yet placed (tests). */
int vox_hip_encoder_aligned(const vox_hip_engine_t *e);
int vox_hip_encoder_pos(const vox_hip_engine_t *e);
int vox_hip_pending_fences(const vox_hip_engine_t *e);

/* ---- state ---------------------------------------------------------------------- */
void vox_hip_reset_encoder(vox_hip_engine_t *e);   /* mel queue, conv tails, encoder KV, 4x residual */
void vox_hip_reset_decoder(vox_hip_engine_t *e);   /* decoder KV + adapter buffer */
void vox_hip_reset_decoder_kv(vox_hip_engine_t *e);/* decoder KV only (prefill restart, voxtral.c:1001-1002) */
int  vox_hip_decoder_kv_len(const vox_hip_engine_t *e);   /* logical positions stored so far */
int  vox_hip_mel_queue_len(const vox_hip_engine_t *e);
void vox_hip_sync(vox_hip_engine_t *e);

/* ---- kernel-level test/bench surface (host buffers; mirrors voxtral_kernels.h) -- */
/* y[M,N] = x[M,K] @ W_bf16[N,K]^T (+bias) — vox_linear_bf16 (voxtral_kernels.c:216).
 * impl: 0 auto (M==1 -> GEMV, else MFMA GEMM), 1 force GEMV rows, 2 force MFMA GEMM,
 * 3 scalar reference kernel (plain fp32 FMA loop, for cross-checks). */
int vox_hip_linear_bf16(vox_hip_engine_t *e, float *y, const float *x, const uint16_t *w,
                        const float *bias, int M, int K, int N, int impl);
/* vox_causal_attention (voxtral_kernels.c:412) for head_dim 64 (encoder) / 128 (decoder). */
int vox_hip_causal_attention(vox_hip_engine_t *e, float *out, const float *q, const float *k,
                             const float *v, int seq_q, int seq_k, int n_heads, int n_kv_heads,
                             int head_dim, float scale, int window, int q_offset);
/* Micro-benchmark of the decode GEMV on resident weights: streams `bytes` of the
 * decoder weights `iters` times; returns average seconds per pass measured with HIP
 * events on the engine stream (used by bench.py's roofline leg). */
double vox_hip_time_decoder_step(vox_hip_engine_t *e, int iters, int kv_len);
/* The same for the encoder stack on an n_rows-row chunk (vox_encoder_forward_incremental, voxtral_encoder.c:452-636) that
 * follows ctx_rows already-encoded positions (the K/V window is full from 750 on): HIP events on the engine stream around
 * `iters` passes of all layers on resident weights; seconds per pass.  Resets the encoder stream state (call it between
 * streams only).  bench.py's roofline leg of the streaming configuration. */
double vox_hip_time_encoder_rows(vox_hip_engine_t *e, int n_rows, int ctx_rows, int iters);

/* Per-kernel breakdown of one decode step: avg_us[9] / launches[9] indexed by
 * {0 step_begin, 1 qkv gemv, 2 attention, 3 split-K combine, 4 wo gemv, 5 swiglu gemv,
 *  6 w2 gemv, 7 logits gemv, 8 argmax}; HIP events on the engine stream. Returns s/step. */
double vox_hip_profile_decode(vox_hip_engine_t *e, int iters, int kv_len, double *avg_us, int *launches);

/* Seconds per launch of n back-to-back launches of an (almost) empty kernel with `grid` blocks of 256
 * threads on the engine stream: the kernel-boundary floor of the launch model on this stack. */
double vox_hip_time_empty_launches(vox_hip_engine_t *e, int n, int grid);
/* The same chain captured into a hipGraph and replayed (no host launch cost in the timed region). */
double vox_hip_time_empty_launches_graph(vox_hip_engine_t *e, int n, int grid);

/* In-situ cost of one decode kernel kind (1 qkv, 2 attention, 4 wo, 5 w1;w3, 6 w2): seconds per step
 * with and without its launches; (full - skipped) / layers = what one launch adds to the chain. */
int vox_hip_time_decoder_step_without(vox_hip_engine_t *e, int iters, int kv_len, int kind,
                                      double *full_s, double *skipped_s);
/* Launches of k_ffn_attn12 in a decode step at this KV length (0 = the step uses k_dec_attn_fused + k_ffn_fused per layer).  With a non-zero
 * answer, kind 6 of vox_hip_time_decoder_step_without leaves out exactly these launches. */
int vox_hip_merged_launches_per_step(const vox_hip_engine_t *e, int kv_len);
/* Round 5: layers whose blocks run inside the ONE k_dec_stack launch of a decode step at this KV length (FFN block of layer 0, attention
 * and FFN blocks of layers 1 .. L-1; 0 = the stack kernel is not used there: beyond 1024 keys, fp8 mode, VOX_HIP_DISABLE=stack).  With a
 * non-zero answer, kind 6 of vox_hip_time_decoder_step_without leaves out exactly that launch. */
int vox_hip_stack_layers(const vox_hip_engine_t *e, int kv_len);
/* 1 if `name` is listed in VOX_HIP_DISABLE (comma-separated): the one switch of the fallback ladder - every name turns one production
 * kernel family off so that the next older HIP path runs instead (never a CPU path).  The names are listed at vox_disabled() in
 * csrc/vox_hip_engine.hip. */
int vox_hip_switch_disabled(const char *name);

/* BASELINE config 5: quantise the decoder matrices and the tied embedding to fp8 e4m3 (one f32 scale
 * per output row) for the decode GEMVs; prefill and the encoder keep bf16.  Call after the uploads.
 * vox_hip_weight_format: 0 = bf16, 1 = fp8 decode weights. */
int vox_hip_quantize_decoder_fp8(vox_hip_engine_t *e);
int vox_hip_weight_format(vox_hip_engine_t *e);
/* Agreement study hook (tools/fp8_agreement.py): the decode step streams bf16 copies of the decoder matrices (lm_head != 0:
 * and of the LM head) holding dequant(quant_e4m3(w)) with one power-of-two scale per `block` weights of a row (0 = per row) -
 * bit for bit what a block-scaled fp8 GEMV with such scales computes.  block < 0 = off.  4B geometry, bf16 mode only. */
int vox_hip_simulate_block_fp8(vox_hip_engine_t *e, int block, int lm_head);

/* Which kernel families are live (bit set = the production variant).  The start-up self-tests compare
 * each MFMA / DPP kernel with a plain HIP cross-check; a mismatch makes vox_hip_engine_create (and so
 * vox_load) FAIL unless VOX_HIP_ALLOW_FALLBACK=1, in which case the bit is cleared.  The A/B switches
 * names in VOX_HIP_DISABLE clear bits too.  VOX_PATH_GEMV3 is reported only for the 4B decoder shapes the kernel
 * is specialised for (other geometries run the generic k_gemv). */
enum vox_hip_path {
    VOX_PATH_GEMM_MFMA_BF16X3 = 1u << 0,   /* large-M GEMM: 3-term bf16 split on v_mfma_f32_32x32x16_bf16 */
    VOX_PATH_GEMM_MFMA_F32    = 1u << 1,   /* large-M GEMM: v_mfma_f32_32x32x2_f32 (K % 64 != 0 shapes)    */
    VOX_PATH_GEMM_SPLITK      = 1u << 2,   /* split-K + fixed-order reduce for skinny tile counts          */
    VOX_PATH_ATTN_ENC_MFMA    = 1u << 3,   /* encoder attention on the f32 MFMA                            */
    VOX_PATH_ATTN_DEC_DPP     = 1u << 4,   /* decoder attention with DPP row reductions                    */
    VOX_PATH_GEMV3            = 1u << 5,   /* decode GEMVs: k_gemv3 (LDS-DMA activations, ordered queue)   */
    VOX_PATH_FP8_DECODE       = 1u << 6,   /* fp8 decode weights in use (config 5 only)                    */
    VOX_PATH_SKINNY_ENC       = 1u << 7,   /* streaming encoder chunks (<= 32 rows) on the weight-streaming kernels */
    VOX_PATH_DEC_FUSED        = 1u << 8,   /* decode step: qkv + attention + wo as one launch (k_dec_attn_fused), 3 launches per layer */
    VOX_PATH_GEMM_PLANES      = 1u << 9,   /* large-M GEMMs on producer-split bf16 planes, LDS-DMA pipeline (k_gemm_planes) */
    VOX_PATH_FFN_FUSED        = 1u << 10,  /* decode step: the FFN block as one launch (k_ffn_fused), 2 launches per layer */
    VOX_PATH_ROWSGEMM         = 1u << 11,  /* 33 .. 128-row passes (decoder prefill, encoder flush) on the weight-streaming MFMA kernel k_rowsgemm */
    VOX_PATH_FFN_ATTN12       = 1u << 12,  /* decode step: FFN block of layer l + attention block of layer l + 1 as ONE launch (k_ffn_attn12, every context length since round 5; fp8: k_w2x_attn12 up to 1024 keys) */
    VOX_PATH_DEC_STACK        = 1u << 13,  /* decode step: FFN(0) and the attention + FFN blocks of layers 1 .. L-1 as ONE launch (k_dec_stack): 4 launches per token */
    VOX_PATH_ENC_STACK        = 1u << 15,  /* streaming encoder chunks (<= 32 rows): all layers of a chunk as ONE persistent launch (k_enc_stack, round 6); VOX_PATH_SKINNY_ENC's launches are its fallback */
    VOX_PATH_FP8_MFMA         = 1u << 14,  /* fp8 mode (config 5 only): the decoder prefill multiplies on the fp8 MFMA (k_rowsgemm_f8: e4m3 weights, activations as two e4m3 terms) */
};
#define VOX_PATH_ALL_BF16 (VOX_PATH_GEMM_MFMA_BF16X3 | VOX_PATH_GEMM_MFMA_F32 | VOX_PATH_GEMM_SPLITK | \
                           VOX_PATH_ATTN_ENC_MFMA | VOX_PATH_ATTN_DEC_DPP | VOX_PATH_GEMV3 | VOX_PATH_DEC_FUSED | VOX_PATH_SKINNY_ENC | \
                           VOX_PATH_GEMM_PLANES | VOX_PATH_FFN_FUSED | VOX_PATH_ROWSGEMM | VOX_PATH_FFN_ATTN12 | VOX_PATH_DEC_STACK | VOX_PATH_ENC_STACK)
unsigned vox_hip_active_paths(const vox_hip_engine_t *e);

/* The fused decode kernel (VOX_PATH_DEC_FUSED) needs its 256 workgroups co-resident; a hand-off that times out (another
 * process or stream held CUs) makes the engine repeat the batch on the launch-per-GEMV chain and SUSPEND the fused kernel
 * for 256 clean decode steps (doubling with every further time-out, capped at 16384), after which it is re-armed.
 * failures = time-outs so far, armed = 1 if the fused kernel is live now, rearm_in = steps left of a suspension.
 * Returns 0 if the engine has the fused kernel at all, 1 if not, -1 on error. */
int vox_hip_fuse_stats(const vox_hip_engine_t *e, int *failures, int *armed, long *rearm_in);
/* Diagnosis: the bounded spins of the decode launches budget ACTIVE waiting time (gaps between consecutive polls, each capped at
 * 40 us), so that a dispatch that was switched out for milliseconds (more HSA queues in the process than the hardware scheduler
 * maps) does not read as a time-out.  Holes (> 1 ms between two polls) are counted: *count = how many so far, *longest_us = the
 * longest.  Returns 0, -1 on error / no fused kernels. */
int vox_hip_spin_holes(vox_hip_engine_t *e, unsigned long long *count, double *longest_us);
/* fp8 mode (VOX_PATH_FP8_MFMA): k_rowsgemm_f8 splits the f32 activations into two e4m3 terms behind a fixed prescale, which covers
 * |x| <= 112.  Activations beyond that are COUNTED; a prefill that counted any is repeated on the bf16 matrices and the fp8 MFMA prefill
 * stays off for the engine (the bit disappears from vox_hip_active_paths).  *fallbacks = such repeats so far, *clamped = the device
 * counter now (read and cleared; vox_hip_linear_bf16 impl 6 counts into it too). */
int vox_hip_fp8_prefill_stats(vox_hip_engine_t *e, int *fallbacks, unsigned *clamped);
/* The encoder stack kernel (VOX_PATH_ENC_STACK, k_enc_stack) needs its 256 workgroups co-resident too; a hand-off that times out makes
 * the engine repeat the chunk on the launch-per-GEMM path (VOX_PATH_SKINNY_ENC) and suspend the stack kernel for 64 chunks (doubling).
 * launches / failures so far, armed = 1 if it is live now.  vox_hip_debug_inject_enc_stack_timeout: test hook, the next chunk's check
 * behaves as if a hand-off had timed out. */
int vox_hip_enc_stack_stats(const vox_hip_engine_t *e, long *launches, int *failures, int *armed);
int vox_hip_debug_inject_enc_stack_timeout(vox_hip_engine_t *e);
/* Test hook: the next check after a synchronisation behaves as if a hand-off had timed out. */
int vox_hip_debug_inject_fuse_timeout(vox_hip_engine_t *e);
/* Test hook: set the hand-off epoch counter (the tags of the {epoch, value} granules) - e.g. just below the point where
 * the engine zeroes the granule buffers and restarts it (0xFFF00000).  Returns the previous value via *old (may be NULL). */
int vox_hip_debug_set_handoff_epoch(vox_hip_engine_t *e, unsigned epoch, unsigned *old);
/* Test hook: residual-stream taps of the decode steps that run at the listed KV positions (n <= 16): x at the start of every
 * layer, x after every attention block, x after the last layer = [2 L + 1][dec_dim] per position (the inputs of the reference's
 * vox_rms_norm calls inside vox_decoder_forward, voxtral_decoder.c:653-694).  Copied in stream order; the kernels are unchanged.
 * vox_hip_debug_tap_read writes [n][2 L + 1][dec_dim] and ends the tapping. */
int vox_hip_debug_tap_config(vox_hip_engine_t *e, const int *positions, int n);
int vox_hip_debug_tap_read(vox_hip_engine_t *e, float *out);

/* Experiment: seconds per pass over ONE decoder layer's five kernels run back to back (weights
 * stay in the 256 MB Infinity Cache), for comparison with the streamed per-layer time. */
double vox_hip_time_layer_repeat(vox_hip_engine_t *e, int iters, int kv_len, double *avg_us, int *launches);

/* Timing of the last fused calls (HIP events on the engine stream), milliseconds. */
typedef struct vox_hip_timing {
    double encode_ms, prefill_ms, decode_ms;
    int    decode_steps;
} vox_hip_timing_t;
void vox_hip_get_timing(const vox_hip_engine_t *e, vox_hip_timing_t *t);
void vox_hip_reset_timing(vox_hip_engine_t *e);
void vox_hip_add_encode_ms(vox_hip_engine_t *e, double ms);   /* encoder time measured by the host (multi-engine chunk) */

#ifdef __cplusplus
}
#endif
#endif /* VOX_HIP_H */
