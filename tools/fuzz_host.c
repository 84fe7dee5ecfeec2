/* Mutation fuzzer for the host-side parsers (WAV, safetensors header, tekken.json) under
 * AddressSanitizer + UBSan: they read files handed in by users.  Built and run by
 * tests/test_host_fuzz.py:
 *     gcc -O1 -g -fsanitize=address,undefined -Iinclude -Ivoxtral_c_amd/host tools/fuzz_host.c \
 *         voxtral_c_amd/host/vox_safetensors.c voxtral_c_amd/host/vox_tokenizer.c voxtral_c_amd/host/vox_audio_wav.c ...
 * usage: fuzz_host <seed file wav> <seed file safetensors> <seed file tekken.json> <iterations> <rng seed>
 * Any sanitizer report aborts with a non-zero exit code. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "voxtral_audio.h"
#include "voxtral_tokenizer.h"
#include "vox_safetensors.h"

static uint64_t rng_state;
static uint64_t rnd(void) {
    rng_state += 0x9e3779b97f4a7c15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static uint8_t *slurp(const char *path, size_t *n, size_t cap) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t *b = malloc(cap);
    *n = fread(b, 1, cap, f);
    fclose(f);
    return b;
}

static size_t mutate(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    size_t m = n;
    memcpy(dst, src, n);
    const int kind = (int)(rnd() % 6);
    if (kind == 0 && m > 8) m = rnd() % m;                                  /* truncate */
    const int flips = 1 + (int)(rnd() % 8);
    for (int i = 0; i < flips && m > 0; i++) {
        const size_t p = (kind == 1) ? rnd() % (m < 64 ? m : 64) : rnd() % m;   /* kind 1: header area */
        switch (rnd() % 4) {
            case 0: dst[p] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 1: dst[p] = (uint8_t)rnd(); break;
            case 2: dst[p] = 0xff; break;
            default: dst[p] = 0; break;
        }
    }
    if (kind == 2 && m + 16 < cap) { memset(dst + m, (int)(rnd() & 0xff), 16); m += 16; }   /* junk tail */
    if (kind == 4 && m > 32) {                                               /* structural characters (JSON) */
        static const char sc[] = "\"\\{}[]:,u0\n";
        const int k = 1 + (int)(rnd() % 6);
        for (int i = 0; i < k; i++) dst[rnd() % m] = (uint8_t)sc[rnd() % (sizeof sc - 1)];
    }
    if (kind == 5 && m > 64) {                                               /* delete a span */
        const size_t p = rnd() % (m - 32), len = 1 + rnd() % 31;
        memmove(dst + p, dst + p + len, m - p - len);
        m -= len;
    }
    if (kind == 3 && m > 16) {                                               /* huge little-endian length */
        const size_t p = rnd() % (m - 8);
        const uint64_t big = (rnd() & 1) ? 0xffffffffffffffffull : (0x7fffffffull + rnd() % 4096);
        memcpy(dst + p, &big, (rnd() & 1) ? 8 : 4);
    }
    return m;
}

static void write_tmp(const char *path, const uint8_t *b, size_t n) {
    FILE *f = fopen(path, "wb");
    fwrite(b, 1, n, f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s wav safetensors tekken.json iterations seed\n", argv[0]); return 2; }
    const int iters = atoi(argv[4]);
    rng_state = strtoull(argv[5], NULL, 10);
    const size_t CAP = 1 << 20;
    size_t nw, ns, nt;
    uint8_t *wav = slurp(argv[1], &nw, CAP), *st = slurp(argv[2], &ns, CAP), *tk = slurp(argv[3], &nt, CAP);
    uint8_t *buf = malloc(CAP + 64);
    char p_st[64], p_tk[64], p_wav[64];
    snprintf(p_st, sizeof p_st, "/tmp/fuzz_%d.safetensors", (int)getpid());
    snprintf(p_tk, sizeof p_tk, "/tmp/fuzz_%d.json", (int)getpid());
    snprintf(p_wav, sizeof p_wav, "/tmp/fuzz_%d.wav", (int)getpid());
    long ok_wav = 0, ok_st = 0, ok_tk = 0;
    for (int i = 0; i < iters; i++) {
        size_t m = mutate(wav, nw, buf, CAP);
        int n = 0;
        float *s = vox_parse_wav_buffer(buf, m, &n);
        if (s) { ok_wav++; volatile float acc = 0; for (int j = 0; j < n; j += 97) acc += s[j]; (void)acc; free(s); }
        if (i % 8 == 0) { write_tmp(p_wav, buf, m); s = vox_load_wav(p_wav, &n); if (s) free(s); }

        m = mutate(st, ns, buf, CAP);
        write_tmp(p_st, buf, m);
        vox_st_file_t *f = vox_st_open(p_st);
        if (f) {
            ok_st++;
            const vox_st_tensor_t *t = vox_st_find(f, "mm_streams_embeddings.embedding_module.tok_embeddings.weight");
            if (t) (void)vox_st_numel(t);
            static const char *names[] = {"norm.weight", "layers.0.attention.wq.weight",
                                          "mm_streams_embeddings.embedding_module.tok_embeddings.weight"};
            for (int k = 0; k < 3; k++) {
                const vox_st_tensor_t *u = vox_st_find(f, names[k]);
                if (u) { float *v = vox_st_to_f32(u); if (v) { volatile float a = v[0]; (void)a; free(v); } }
            }
            vox_st_close(f);
        }

        m = mutate(tk, nt, buf, CAP);
        write_tmp(p_tk, buf, m);
        vox_tokenizer_t *T = vox_tokenizer_load(p_tk);
        if (T) {
            ok_tk++;
            for (int id = 0; id < 4096; id += 61) { const char *p = vox_tokenizer_decode(T, id); if (p) { volatile size_t l = strlen(p); (void)l; } }
            vox_tokenizer_free(T);
        }
    }
    unlink(p_st); unlink(p_tk); unlink(p_wav);
    free(wav); free(st); free(tk); free(buf);
    printf("fuzz_host: %d iterations, accepted wav %ld / safetensors %ld / tokenizer %ld, no sanitizer report\n", iters, ok_wav, ok_st, ok_tk);
    return 0;
}
