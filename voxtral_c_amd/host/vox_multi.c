/* vox_multi.c — several GPUs behind voxtral.h: exact context-parallel encoder of a large chunk (a stream's first one or any later one).
 *
 * BASELINE config 4 / SURVEY 8(e) option E1, inside the C library (main.c uses it through VOX_DEVICES=0,1,...; no Python,
 * no torch).  The reference has nothing distributed; what is reproduced exactly is its single-GPU arithmetic:
 * every encoder position sees the same K/V window as in one long incremental call (voxtral_encoder.c:452-636).
 *
 *   - the chunk's encoder positions are split into N contiguous token-aligned ranges; engine r gets its mel frames
 *     (plus a 4-frame halo) from the stream engine's device mel queue by a peer copy and runs its own conv stem;
 *   - the 32 layers run as a wavefront: right behind its layer-l kernels engine r pushes the layer-l K/V of its last
 *     window-1 positions into engine r+1's ring (one peer copy per neighbour pair and layer, xGMI point to point) and
 *     engine r+1's stream waits for exactly that event before its own layer l; the host only enqueues;
 *   - adapter rows are written by each engine straight into the stream engine's adapter buffer; the last engine's
 *     streaming state (KV rings, conv history, position) moves to the stream engine, which carries on alone:
 *     incremental chunks, prefill and the strictly sequential decoder (replicas only).
 */
#include "vox_internal.h"
#include <stdio.h>

#define MIN_TOKENS_PER_ENGINE 16

/* Encode the first `frames_avail` mel frames queued on the stream engine across all engines.  Round 4: not only a stream's first
 * chunk - any chunk, as long as the stream engine's encoder state sits on a token boundary (no conv0 frame waiting for its
 * stride-2 partner, no encoder rows waiting for 4x alignment: always true while every chunk so far was a multiple of 8 frames,
 * which is what this function consumes).  Engine 0 continues from its own state (conv history, K/V rings: base position B);
 * engines 1 .. N-1 start from a zero conv history 4 frames early and get their K/V halo from their left neighbour layer by layer.
 * Uses a multiple of 8 frames (whole tokens); returns the number of frames consumed (0 = chunk too small or state not aligned:
 * the caller runs the single-engine path) or -1.  *new_tokens = adapter rows appended (their contents arrive stream-ordered: the
 * decoder waits for each shard's rows when it reaches them, vox_hip_shard_end_push). */
int vox_multi_encode_chunk(vox_ctx_t *ctx, int frames_avail, int *new_tokens) {
    const int N = ctx->n_shard_engines;
    *new_tokens = 0;
    if (N < 2) return 0;
    const int T = frames_avail / 8;
    if (T < MIN_TOKENS_PER_ENGINE * N) return 0;
    vox_hip_engine_t **E = (vox_hip_engine_t **)ctx->shard_engines;
    if (!vox_hip_encoder_aligned(E[0])) return 0;
    const int F = T * 8, W = ctx->dims.enc_window, L = ctx->dims.enc_layers;
    const int B = vox_hip_encoder_pos(E[0]);                 /* encoder positions already done by this stream */
    int pos0[VOX_MAX_DEVICES], pos1[VOX_MAX_DEVICES];
    for (int r = 0, t = 0; r < N; r++) {
        const int cnt = T / N + (r < T % N ? 1 : 0);
        pos0[r] = B + 4 * t; pos1[r] = B + 4 * (t + cnt); t += cnt;
    }
    for (int r = 1; r < N; r++) vox_hip_sync(E[r]);
    const int64_t first_row = vox_hip_adapter_extend(E[0], T);
    if (first_row < 0) return -1;
    /* rows p in [pos0, pos1) need conv0 frames 2p-1 .. 2p+1 and those need mel frames 2p-3 .. 2p+1: feeding an engine
     * from frame 2 pos0 - 4 through a zero-history conv stem contaminates exactly its first two rows.  Frame indices below are
     * relative to the stream engine's queue, whose first frame is global frame 2 B. */
    for (int r = 1; r < N; r++) {
        vox_hip_reset_encoder(E[r]);
        if (vox_hip_mel_queue_push(E[0], E[r], 2 * (pos0[r] - B) - 4, 2 * (pos1[r] - pos0[r]) + 4)) return -1;
    }
    for (int r = 0; r < N; r++) {
        const int n_mel = r == 0 ? 2 * (pos1[0] - B) : 2 * (pos1[r] - pos0[r]) + 4;
        const int rows = vox_hip_shard_begin(E[r], n_mel, r == 0 ? 0 : 2, pos0[r]);
        if (rows != pos1[r] - pos0[r]) { fprintf(stderr, "vox_multi: shard %d got %d rows, expected %d (%s)\n", r, rows, pos1[r] - pos0[r], vox_hip_last_error()); return -1; }
    }
    if (vox_hip_mel_queue_drop(E[0], F - 2 * (pos1[0] - B))) return -1;     /* frames that went to the other engines */
    for (int l = 0; l < L; l++)
        for (int r = 0; r < N; r++) {
            if (vox_hip_shard_layer(E[r], l)) return -1;
            if (r + 1 < N) {
                const int tail = pos1[r] < W - 1 ? pos1[r] : W - 1;            /* what engine r+1's first rows look back on */
                if (vox_hip_shard_kv_push(E[r], E[r + 1], l, pos1[r] - tail, tail)) return -1;
            }
        }
    for (int r = 0; r < N; r++)
        if (vox_hip_shard_end_push(E[r], E[0], first_row + (pos0[r] - B) / 4) != (pos1[r] - pos0[r]) / 4) return -1;
    if (vox_hip_encoder_state_push(E[N - 1], E[0])) return -1;
    *new_tokens = T;
    return F;
}
