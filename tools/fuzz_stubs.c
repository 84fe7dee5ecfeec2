/* Link-time stand-ins for the two device entry points vox_audio.c references; the fuzzer only
 * exercises the WAV / safetensors / tokenizer parsers, which never reach them. */
#include <stdlib.h>
struct vox_hip_engine;
struct vox_hip_engine *vox_default_mel_engine(void) { abort(); }
int vox_hip_mel_frames(struct vox_hip_engine *e, const float *s, int n, float *out, int q) { (void)e; (void)s; (void)n; (void)out; (void)q; abort(); }
int vox_verbose = 0;
