/* voxtral_mic.h — microphone capture surface (reference voxtral_mic.h:13-23).
 * Device capture is macOS-only in the reference; on Linux it ships stubs
 * (voxtral_mic_macos.c:126-144) and so does this library, so that the reference CLI
 * links unchanged. */
#ifndef VOXTRAL_MIC_H
#define VOXTRAL_MIC_H
#ifdef __cplusplus
extern "C" {
#endif
int vox_mic_start(void);
int vox_mic_read(float *out, int max_samples);
int vox_mic_read_available(void);
void vox_mic_stop(void);
#ifdef __cplusplus
}
#endif
#endif
