/*
 * voxtral_audio.h — WAV ingestion and log-mel front-end.
 * Same surface as the reference header (voxtral_audio.h:18-69).  WAV parsing is host
 * code; the spectrogram itself (windowed 400-point DFT, Slaney filterbank, log clamp:
 * reference voxtral_audio.c:454-513) runs on the GPU (k_mel_frames).
 */
#ifndef VOXTRAL_AUDIO_H
#define VOXTRAL_AUDIO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

extern int vox_verbose_audio;

/* 16-bit PCM WAV (any rate / channel count) -> mono f32 @16 kHz in [-1,1]; caller frees.
 * reference: vox_load_wav voxtral_audio.c:143, vox_parse_wav_buffer :49, vox_read_pcm_stdin :168 */
float *vox_load_wav(const char *path, int *out_n_samples);
float *vox_parse_wav_buffer(const uint8_t *data, size_t size, int *out_n_samples);
float *vox_read_pcm_stdin(int *out_n_samples);

/* Batch log-mel with reflect padding (reference vox_mel_spectrogram voxtral_audio.c:294).
 * Returns [n_frames, 128] (caller frees). */
float *vox_mel_spectrogram(const float *samples, int n_samples, int *out_frames);

/* Incremental log-mel (reference voxtral_audio.c:515-662). */
typedef struct vox_mel_ctx vox_mel_ctx_t;
vox_mel_ctx_t *vox_mel_ctx_init(int left_pad_samples);
int    vox_mel_feed(vox_mel_ctx_t *ctx, const float *samples, int n_samples);
int    vox_mel_finish(vox_mel_ctx_t *ctx, int right_pad_samples);
float *vox_mel_data(vox_mel_ctx_t *ctx, int *out_n_frames);
int    vox_mel_frame_offset(vox_mel_ctx_t *ctx);
void   vox_mel_discard_before(vox_mel_ctx_t *ctx, int keep_from_frame);
void   vox_mel_free(vox_mel_ctx_t *ctx);

#ifdef __cplusplus
}
#endif
#endif
