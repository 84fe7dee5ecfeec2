/* vox_internal.h — private glue between the host-side translation units. */
#ifndef VOX_INTERNAL_H
#define VOX_INTERNAL_H
#include "../../include/voxtral.h"
#include "../../include/voxtral_audio.h"
#include "../../include/voxtral_tokenizer.h"
#include "../../include/vox_hip.h"

#define VOX_MEL_NFFT  400
#define VOX_MEL_NFREQ 201

/* Host-built mel tables (filters [128,201], hann[400], dft cos/sin [201,400]). */
typedef struct {
    float *filters, *hann, *dft_cos, *dft_sin;
} vox_mel_tables_t;
const vox_mel_tables_t *vox_mel_tables(void);

/* Mel context bound to a device engine.  queue_mode != 0: frames are appended to the
 * engine's device mel queue and never copied back (stream path); otherwise they are
 * downloaded into a host buffer (public vox_mel_* API). */
vox_mel_ctx_t *vox_mel_ctx_init_engine(vox_hip_engine_t *engine, int left_pad_samples, int queue_mode);
/* Total frames produced so far (global index of the next frame). */
int vox_mel_total_frames(vox_mel_ctx_t *ctx);
/* Engine used by the engine-less public mel API (created on first use). */
vox_hip_engine_t *vox_default_mel_engine(void);

/* host/vox_multi.c: sharded encoder of a large chunk over ctx->shard_engines; frames consumed / 0 / -1 */
int vox_multi_encode_chunk(vox_ctx_t *ctx, int frames_avail, int *new_tokens);

#endif
