/* vox_audio.c — WAV ingestion (host) and the sample-side bookkeeping of the log-mel
 * front-end.  The spectrogram arithmetic itself runs on the GPU (k_mel_frames in
 * csrc/vox_misc.h); this file decides *which* 400-sample windows exist, exactly as the
 * reference's incremental mel does (voxtral_audio.c:432-662):
 *   - the stream starts with 200 + left_pad zeros (center padding over silence + the 32
 *     left-pad tokens),
 *   - frame t covers padded samples [160 t, 160 t + 400),
 *   - finish() appends right_pad zeros and a 200-sample reflection of the tail, computes
 *     what fits and drops the last frame (vLLM: stft[..., :-1]).
 */
#include "vox_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int vox_verbose_audio = 0;

#define HOP 160
#define WIN 400
#define SR  16000

/* ---------------------------------------------------------------------------------
 * WAV
 * --------------------------------------------------------------------------------- */
static unsigned le16(const uint8_t *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

float *vox_parse_wav_buffer(const uint8_t *data, size_t size, int *out_n_samples) {
    if (size < 44 || memcmp(data, "RIFF", 4) || memcmp(data + 8, "WAVE", 4)) {
        fprintf(stderr, "parse_wav_buffer: not a valid WAV file\n");
        return NULL;
    }
    int fmt = 0, channels = 0, rate = 0, bits = 0, pcm_bytes = 0;
    const uint8_t *pcm = NULL, *end = data + size;
    for (const uint8_t *p = data + 12; p + 8 <= end;) {
        const uint32_t csz = le32(p + 4);
        if (!memcmp(p, "fmt ", 4) && csz >= 16 && p + 8 + csz <= end) {
            fmt = (int)le16(p + 8); channels = (int)le16(p + 10);
            rate = (int)le32(p + 12); bits = (int)le16(p + 22);
        } else if (!memcmp(p, "data", 4)) {
            pcm = p + 8;
            pcm_bytes = (int)csz;
            /* streaming writers leave 0xFFFFFFFF here: take the rest of the buffer */
            if (pcm_bytes <= 0 || pcm + pcm_bytes > end) pcm_bytes = (int)(end - pcm);
            break;
        }
        if (p + 8 + csz > end) break;
        p += 8 + csz + (csz & 1);
    }
    if (fmt != 1 || bits != 16 || !pcm || channels < 1) {
        fprintf(stderr, "parse_wav_buffer: unsupported format (need 16-bit PCM, got fmt=%d bits=%d)\n", fmt, bits);
        return NULL;
    }
    /* a zero / absurd sample rate divides by zero in the resampler below (the reference does, too:
     * voxtral_audio.c:112-116); found by tools/fuzz_host.c */
    if (rate < 100 || rate > 3072000) {
        fprintf(stderr, "parse_wav_buffer: unsupported sample rate %d\n", rate);
        return NULL;
    }
    int n = pcm_bytes / (channels * 2);
    float *mono = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!mono) return NULL;
    for (int i = 0; i < n; i++) {
        if (channels == 1) {
            int16_t v; memcpy(&v, pcm + (size_t)i * 2, 2);
            mono[i] = v / 32768.0f;
        } else {
            float acc = 0;
            for (int c = 0; c < channels; c++) { int16_t v; memcpy(&v, pcm + ((size_t)i * channels + c) * 2, 2); acc += v; }
            mono[i] = (acc / channels) / 32768.0f;
        }
    }
    if (rate != SR) {   /* linear-interpolation resampler, as the reference (voxtral_audio.c:114-137) */
        const int m = (int)((long long)n * SR / rate);
        float *rs = (float *)malloc((size_t)(m > 0 ? m : 1) * sizeof(float));
        if (!rs) { free(mono); return NULL; }
        /* The reference is built with -ffast-math (Makefile:5): `(float)i * rate / 16000` becomes a
         * multiplication by the hoisted constant rate * (1/16000), `1 - (pos - k)` becomes `(1 - pos) + k`
         * and the blend is one fused multiply-add.  At source positions ~1e4 an ulp of `pos` is 1e-3 of
         * interpolation weight, so the literal formula differs from the reference binary by up to 2e-4
         * per sample; this is its exact evaluation order (checked against oracle/_ref bit for bit,
         * tests/test_host_cpu.py::test_wav_parser_matches_reference). */
        const float step = (float)rate * 6.25e-05f;
        for (int i = 0; i < m; i++) {
            const float pos = (float)i * step;
            const int k = (int)pos;
            const float fk = (float)k;
            if (k + 1 < n) {
                const float w0 = (1.0f - pos) + fk;
                const float hi = (pos - fk) * mono[k + 1];
                rs[i] = fmaf(w0, mono[k], hi);
            } else rs[i] = (k < n) ? mono[k] : 0.0f;
        }
        free(mono);
        mono = rs; n = m;
        if (vox_verbose_audio) fprintf(stderr, "  Resampled %d -> %d Hz (%d samples)\n", rate, SR, n);
    }
    *out_n_samples = n;
    return mono;
}

float *vox_load_wav(const char *path, int *out_n_samples) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "vox_load_wav: cannot open %s\n", path); return NULL; }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return NULL; }
    uint8_t *buf = (uint8_t *)malloc((size_t)sz);
    if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    float *s = vox_parse_wav_buffer(buf, (size_t)sz, out_n_samples);
    free(buf);
    return s;
}

float *vox_read_pcm_stdin(int *out_n_samples) {
    size_t cap = 1 << 20, size = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return NULL;
    for (;;) {
        if (size == cap) { cap *= 2; uint8_t *t = (uint8_t *)realloc(buf, cap); if (!t) { free(buf); return NULL; } buf = t; }
        const size_t n = fread(buf + size, 1, cap - size, stdin);
        if (!n) break;
        size += n;
    }
    if (size < 4) { fprintf(stderr, "vox_read_pcm_stdin: no data on stdin\n"); free(buf); return NULL; }
    fprintf(stderr, "Read %zu bytes from stdin\n", size);
    float *out;
    if (!memcmp(buf, "RIFF", 4)) {
        fprintf(stderr, "Detected WAV format on stdin\n");
        out = vox_parse_wav_buffer(buf, size, out_n_samples);
    } else {
        fprintf(stderr, "Treating stdin as raw s16le 16kHz mono\n");
        const int n = (int)(size / 2);
        out = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
        if (out) {
            for (int i = 0; i < n; i++) { int16_t v; memcpy(&v, buf + (size_t)i * 2, 2); out[i] = v / 32768.0f; }
            *out_n_samples = n;
        }
    }
    free(buf);
    return out;
}

/* ---------------------------------------------------------------------------------
 * Mel tables — built on the host with the reference's own formulas and operation order
 * (Slaney filterbank: voxtral_audio.c:223-285; periodic Hann and DFT tables: :531-542),
 * then uploaded once per engine.
 * --------------------------------------------------------------------------------- */
static float hz_to_mel(float f) {
    const float logstep = 27.0f / logf(6.4f);
    float m = 3.0f * f / 200.0f;
    if (f >= 1000.0f) m = 15.0f + logf(f / 1000.0f) * logstep;
    return m;
}
static float mel_to_hz(float m) {
    const float logstep = logf(6.4f) / 27.0f;
    float f = 200.0f * m / 3.0f;
    if (m >= 15.0f) f = 1000.0f * expf(logstep * (m - 15.0f));
    return f;
}

const vox_mel_tables_t *vox_mel_tables(void) {
    static vox_mel_tables_t T;
    if (T.filters) return &T;
    const int NM = VOX_MEL_BINS, NF = VOX_MEL_NFREQ, N = VOX_MEL_NFFT;
    T.filters = (float *)calloc((size_t)NM * NF, sizeof(float));
    T.hann = (float *)malloc(N * sizeof(float));
    T.dft_cos = (float *)malloc((size_t)NF * N * sizeof(float));
    T.dft_sin = (float *)malloc((size_t)NF * N * sizeof(float));
    float bin_hz[VOX_MEL_NFREQ], edge[VOX_MEL_BINS + 2], gap[VOX_MEL_BINS + 1];
    for (int i = 0; i < NF; i++) bin_hz[i] = (float)i * ((float)SR / 2.0f) / (float)(NF - 1);
    const float m0 = hz_to_mel(0.0f), m1 = hz_to_mel((float)SR / 2.0f);
    for (int i = 0; i < NM + 2; i++) edge[i] = mel_to_hz(m0 + (m1 - m0) * (float)i / (float)(NM + 1));
    for (int i = 0; i < NM + 1; i++) { gap[i] = edge[i + 1] - edge[i]; if (gap[i] == 0.0f) gap[i] = 1e-6f; }
    for (int m = 0; m < NM; m++) {
        const float norm = 2.0f / (edge[m + 2] - edge[m]);
        for (int k = 0; k < NF; k++) {
            const float rise = (bin_hz[k] - edge[m]) / gap[m];
            const float fall = (edge[m + 2] - bin_hz[k]) / gap[m + 1];
            float v = fminf(rise, fall);
            if (v < 0.0f) v = 0.0f;
            T.filters[(size_t)m * NF + k] = v * norm;
        }
    }
    for (int i = 0; i < N; i++) T.hann[i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)N));
    /* angle = ((2*pi/400)*k)*n with 2*pi/400 folded to one f32 constant: this is what the
     * reference's -ffast-math build evaluates (checked bit-for-bit against the tables in
     * oracle/_ref).  The association order matters: k*n reaches 8e4, so an ulp of the angle
     * is up to 2.4e-4 in the table entry, far above every other error source of the mel. */
    const float step = 2.0f * (float)M_PI / (float)N;
    for (int k = 0; k < NF; k++)
        for (int n = 0; n < N; n++) {
            const float ang = step * (float)k * (float)n;
            T.dft_cos[(size_t)k * N + n] = cosf(ang);
            T.dft_sin[(size_t)k * N + n] = sinf(ang);
        }
    return &T;
}

/* ---------------------------------------------------------------------------------
 * Incremental mel context
 * --------------------------------------------------------------------------------- */
struct vox_mel_ctx {
    vox_hip_engine_t *engine;
    int queue_mode;
    float *buf;            /* padded samples not yet retired */
    int64_t base;          /* global padded-sample index of buf[0] */
    int n, cap;
    int frames_done;       /* global index of the next frame to compute */
    int finished;
    /* host mode only */
    float *mel;
    int mel_n, mel_cap, mel_off;
};

static int grow_samples(vox_mel_ctx_t *c, int extra) {
    if (c->n + extra <= c->cap) return 0;
    int nc = c->cap ? c->cap : 65536;
    while (nc < c->n + extra) nc *= 2;
    float *t = (float *)realloc(c->buf, (size_t)nc * sizeof(float));
    if (!t) return -1;
    c->buf = t; c->cap = nc;
    return 0;
}

/* Frames whose 400-sample window lies inside the samples seen so far. */
static int frames_available(const vox_mel_ctx_t *c) {
    const int64_t total = c->base + c->n;
    if (total < WIN) return 0;
    return (int)((total - WIN) / HOP) + 1;
}

/* Compute frames [frames_done, upto) on the device; retire consumed samples. */
static int compute_frames(vox_mel_ctx_t *c, int upto) {
    const int t0 = c->frames_done, cnt = upto - t0;
    if (cnt <= 0) return 0;
    const int64_t first = (int64_t)t0 * HOP - c->base;
    if (first < 0) return 0;
    float *dst = NULL;
    if (!c->queue_mode) {
        if (c->mel_n + cnt > c->mel_cap) {
            int nc = c->mel_cap ? c->mel_cap : 1024;
            while (nc < c->mel_n + cnt) nc *= 2;
            float *t = (float *)realloc(c->mel, (size_t)nc * VOX_MEL_BINS * sizeof(float));
            if (!t) return 0;
            c->mel = t; c->mel_cap = nc;
        }
        dst = c->mel + (size_t)c->mel_n * VOX_MEL_BINS;
    }
    if (vox_hip_mel_frames(c->engine, c->buf + first, cnt, dst, c->queue_mode) != 0) return 0;
    if (!c->queue_mode) c->mel_n += cnt;
    c->frames_done = upto;
    /* samples before the next frame's window are never needed again */
    const int64_t drop = (int64_t)upto * HOP - c->base;
    if (drop > 0 && drop <= c->n) {
        memmove(c->buf, c->buf + drop, (size_t)(c->n - drop) * sizeof(float));
        c->n -= (int)drop;
        c->base += drop;
    }
    return cnt;
}

vox_mel_ctx_t *vox_mel_ctx_init_engine(vox_hip_engine_t *engine, int left_pad_samples, int queue_mode) {
    if (!engine) return NULL;
    vox_mel_ctx_t *c = (vox_mel_ctx_t *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->engine = engine;
    c->queue_mode = queue_mode;
    const int pad = 200 + left_pad_samples;
    if (grow_samples(c, pad + SR) != 0) { free(c); return NULL; }
    memset(c->buf, 0, (size_t)pad * sizeof(float));
    c->n = pad;
    return c;
}

vox_mel_ctx_t *vox_mel_ctx_init(int left_pad_samples) {
    return vox_mel_ctx_init_engine(vox_default_mel_engine(), left_pad_samples, 0);
}

int vox_mel_feed(vox_mel_ctx_t *c, const float *samples, int n_samples) {
    if (!c || n_samples <= 0) return 0;
    if (grow_samples(c, n_samples) != 0) return 0;
    memcpy(c->buf + c->n, samples, (size_t)n_samples * sizeof(float));
    c->n += n_samples;
    return compute_frames(c, frames_available(c));
}

int vox_mel_finish(vox_mel_ctx_t *c, int right_pad_samples) {
    if (!c) return 0;
    if (c->finished) return vox_mel_total_frames(c) - c->mel_off;
    if (right_pad_samples > 0) {
        if (grow_samples(c, right_pad_samples) != 0) return vox_mel_total_frames(c) - c->mel_off;
        memset(c->buf + c->n, 0, (size_t)right_pad_samples * sizeof(float));
        c->n += right_pad_samples;
    }
    if (grow_samples(c, 200) != 0) return vox_mel_total_frames(c) - c->mel_off;
    const int real_end = c->n - right_pad_samples;
    for (int i = 0; i < 200; i++) {
        const int src = real_end - 2 - i;
        c->buf[c->n + i] = (src >= 0) ? c->buf[src] : 0.0f;
    }
    c->n += 200;
    /* everything that now fits, minus the last frame */
    const int avail = frames_available(c);
    if (avail - 1 > c->frames_done) compute_frames(c, avail - 1);
    else if (avail <= c->frames_done && !c->queue_mode && c->mel_n > 0) { c->mel_n--; c->frames_done--; }
    c->finished = 1;
    return vox_mel_total_frames(c) - c->mel_off;
}

int vox_mel_total_frames(vox_mel_ctx_t *c) { return c ? c->frames_done : 0; }

float *vox_mel_data(vox_mel_ctx_t *c, int *out_n_frames) {
    if (!c || c->queue_mode) { if (out_n_frames) *out_n_frames = 0; return NULL; }
    if (out_n_frames) *out_n_frames = c->mel_n;
    return c->mel;
}

int vox_mel_frame_offset(vox_mel_ctx_t *c) { return c ? c->mel_off : 0; }

void vox_mel_discard_before(vox_mel_ctx_t *c, int keep_from_frame) {
    if (!c || c->queue_mode || keep_from_frame <= c->mel_off) return;
    int drop = keep_from_frame - c->mel_off;
    if (drop > c->mel_n) drop = c->mel_n;
    if (drop <= 0) return;
    memmove(c->mel, c->mel + (size_t)drop * VOX_MEL_BINS, (size_t)(c->mel_n - drop) * VOX_MEL_BINS * sizeof(float));
    c->mel_n -= drop;
    c->mel_off += drop;
}

void vox_mel_free(vox_mel_ctx_t *c) {
    if (!c) return;
    free(c->buf);
    free(c->mel);
    free(c);
}

/* Batch spectrogram with reflect padding of the real audio (reference :294-399). */
float *vox_mel_spectrogram(const float *samples, int n_samples, int *out_frames) {
    vox_hip_engine_t *eng = vox_default_mel_engine();
    if (!eng) return NULL;
    const int pad = WIN / 2, padded = n_samples + 2 * pad;
    const int frames = (padded - WIN) / HOP + 1 - 1;    /* last STFT frame dropped */
    if (frames <= 0) { fprintf(stderr, "vox_mel_spectrogram: audio too short (%d samples)\n", n_samples); return NULL; }
    float *p = (float *)malloc((size_t)padded * sizeof(float));
    if (!p) return NULL;
    for (int i = 0; i < pad; i++) { const int s = pad - i; p[i] = s < n_samples ? samples[s] : 0.0f; }
    memcpy(p + pad, samples, (size_t)n_samples * sizeof(float));
    for (int i = 0; i < pad; i++) { const int s = n_samples - 2 - i; p[pad + n_samples + i] = s >= 0 ? samples[s] : 0.0f; }
    float *mel = (float *)malloc((size_t)frames * VOX_MEL_BINS * sizeof(float));
    if (mel && vox_hip_mel_frames(eng, p, frames, mel, 0) != 0) { free(mel); mel = NULL; }
    free(p);
    if (mel) *out_frames = frames;
    return mel;
}

/* Microphone stubs (Linux), as voxtral_mic_macos.c:126-144. */
int vox_mic_start(void) { fprintf(stderr, "Microphone capture is not supported on this platform\n"); return -1; }
int vox_mic_read(float *out, int max_samples) { (void)out; (void)max_samples; return 0; }
int vox_mic_read_available(void) { return 0; }
void vox_mic_stop(void) {}
