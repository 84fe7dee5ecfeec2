/* vox_tokenizer.c — Tekken id -> text table (decode only).
 * Behaviour follows the reference voxtral_tokenizer.c:186-392: ids 0..999 are the
 * special tokens (by rank), id >= 1000 is vocab[id-1000].token_bytes (base64), pieces are
 * C strings (a NUL byte ends them, which is what makes id 1000 "empty", voxtral.c:487). */
#include "../../include/voxtral_tokenizer.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int vox_verbose;

#define TEKKEN_SPECIAL 1000
#define TEKKEN_MAX_VOCAB 130072   /* reference MAX_VOCAB (voxtral_tokenizer.c:23): 131072 - 1000 */
#define PIECE_MAX 256

struct vox_tokenizer {
    char **vocab;   int n_vocab, cap_vocab;
    char **special; int n_special;
};

typedef struct { const char *p; } jc_t;

static void jws(jc_t *c) { while (*c->p == ' ' || *c->p == '\n' || *c->p == '\t' || *c->p == '\r') c->p++; }

static int hexv(char ch) {
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
    if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
    return 0;
}

/* Parse a JSON string into buf (size cap, always NUL-terminated, silently truncated). */
static int jstr(jc_t *c, char *buf, int cap) {
    jws(c);
    if (*c->p != '"') return -1;
    c->p++;
    int o = 0;
    while (*c->p && *c->p != '"') {
        unsigned char out[4]; int n = 1;
        if (*c->p == '\\') {
            c->p++;
            switch (*c->p) {
                case 'n': out[0] = '\n'; break;
                case 't': out[0] = '\t'; break;
                case 'r': out[0] = '\r'; break;
                case 'u': {
                    unsigned cp = 0;
                    for (int i = 1; i <= 4 && c->p[i]; i++) cp = (cp << 4) | (unsigned)hexv(c->p[i]);
                    int adv = 0; while (adv < 4 && c->p[adv + 1]) adv++;
                    c->p += adv;
                    if (cp < 0x80) { out[0] = (unsigned char)cp; n = 1; }
                    else if (cp < 0x800) { out[0] = 0xC0 | (cp >> 6); out[1] = 0x80 | (cp & 0x3F); n = 2; }
                    else { out[0] = 0xE0 | (cp >> 12); out[1] = 0x80 | ((cp >> 6) & 0x3F); out[2] = 0x80 | (cp & 0x3F); n = 3; }
                    break;
                }
                default: out[0] = (unsigned char)*c->p; break;
            }
        } else out[0] = (unsigned char)*c->p;
        if (*c->p) c->p++;
        if (o + n < cap) { memcpy(buf + o, out, (size_t)n); o += n; }
    }
    buf[o] = 0;
    if (*c->p == '"') c->p++;
    return 0;
}

static void jskip(jc_t *c) {
    jws(c);
    if (*c->p == '"') { char tmp[8]; jstr(c, tmp, sizeof tmp); return; }
    if (*c->p == '{' || *c->p == '[') {
        int depth = 0;
        while (*c->p) {
            if (*c->p == '"') { char tmp[8]; jstr(c, tmp, sizeof tmp); continue; }
            if (*c->p == '{' || *c->p == '[') depth++;
            else if (*c->p == '}' || *c->p == ']') { depth--; if (depth == 0) { c->p++; return; } }
            c->p++;
        }
        return;
    }
    while (*c->p && *c->p != ',' && *c->p != '}' && *c->p != ']') c->p++;
}

static long jint(jc_t *c) {
    jws(c);
    long v = 0; int neg = 0;
    if (*c->p == '-') { neg = 1; c->p++; }
    while (*c->p >= '0' && *c->p <= '9') v = v * 10 + (*c->p++ - '0');
    return neg ? -v : v;
}

static int b64v(unsigned char ch) {
    if (ch >= 'A' && ch <= 'Z') return ch - 'A';
    if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
    if (ch >= '0' && ch <= '9') return ch - '0' + 52;
    if (ch == '+') return 62;
    if (ch == '/') return 63;
    return -1;
}
static int b64_decode(const char *in, char *out, int cap) {
    int o = 0; unsigned acc = 0; int bits = 0;
    for (; *in && *in != '='; in++) {
        int v = b64v((unsigned char)*in);
        if (v < 0) continue;
        acc = (acc << 6) | (unsigned)v; bits += 6;
        if (bits >= 8) { bits -= 8; if (o < cap - 1) out[o++] = (char)((acc >> bits) & 0xFF); }
    }
    out[o] = 0;
    return o;
}

/* Walks an array of objects, handing (rank, value-of-`field`) to the callback. */
static void parse_entries(jc_t *c, const char *field, vox_tokenizer_t *tok, int is_vocab) {
    jws(c);
    if (*c->p != '[') { jskip(c); return; }
    c->p++;
    for (;;) {
        jws(c);
        if (*c->p == ',') { c->p++; continue; }
        if (*c->p != '{') break;
        c->p++;
        long rank = -1; char val[512]; val[0] = 0;
        for (;;) {
            jws(c);
            if (*c->p == ',') { c->p++; continue; }
            if (*c->p != '"') break;
            char key[48];
            jstr(c, key, sizeof key);
            jws(c);
            if (*c->p != ':') break;
            c->p++;
            if (!strcmp(key, "rank")) rank = jint(c);
            else if (!strcmp(key, field)) { jws(c); if (*c->p == '"') jstr(c, val, sizeof val); else jskip(c); }
            else jskip(c);
        }
        if (*c->p == '}') c->p++;
        if (rank < 0 || !val[0]) continue;
        if (is_vocab) {
            /* same bound as the reference (rank < MAX_VOCAB = 131072 - 1000, voxtral_tokenizer.c:263):
             * a hostile rank must neither overflow the capacity doubling nor ask for gigabytes */
            if (rank >= TEKKEN_MAX_VOCAB) continue;
            if (rank >= tok->cap_vocab) {
                int ncap = tok->cap_vocab ? tok->cap_vocab : 1 << 17;
                while (ncap <= rank) ncap *= 2;
                char **nv = (char **)realloc(tok->vocab, (size_t)ncap * sizeof(char *));
                if (!nv) continue;
                tok->vocab = nv;
                memset(tok->vocab + tok->cap_vocab, 0, (size_t)(ncap - tok->cap_vocab) * sizeof(char *));
                tok->cap_vocab = ncap;
            }
            char piece[PIECE_MAX];
            int n = b64_decode(val, piece, sizeof piece);
            char *np = (char *)malloc((size_t)n + 1);
            if (!np) continue;
            free(tok->vocab[rank]);
            tok->vocab[rank] = np;
            memcpy(tok->vocab[rank], piece, (size_t)n + 1);
            if (rank >= tok->n_vocab) tok->n_vocab = (int)rank + 1;
        } else if (rank < TEKKEN_SPECIAL) {
            free(tok->special[rank]);
            tok->special[rank] = strdup(val);
            if (rank >= tok->n_special) tok->n_special = (int)rank + 1;
        }
    }
    jws(c);
    if (*c->p == ']') c->p++;
}

vox_tokenizer_t *vox_tokenizer_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "vox_tokenizer_load: cannot open %s\n", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size <= 0) { fclose(f); return NULL; }
    char *json = (char *)malloc((size_t)size + 1);
    if (!json || fread(json, 1, (size_t)size, f) != (size_t)size) { fclose(f); free(json); return NULL; }
    fclose(f);
    json[size] = 0;

    vox_tokenizer_t *tok = (vox_tokenizer_t *)calloc(1, sizeof *tok);
    tok->special = (char **)calloc(TEKKEN_SPECIAL, sizeof(char *));
    jc_t c = {json};
    jws(&c);
    if (*c.p != '{') { free(json); vox_tokenizer_free(tok); return NULL; }
    c.p++;
    for (;;) {
        jws(&c);
        if (*c.p == ',') { c.p++; continue; }
        if (*c.p != '"') break;
        char key[64];
        jstr(&c, key, sizeof key);
        jws(&c);
        if (*c.p != ':') break;
        c.p++;
        if (!strcmp(key, "vocab")) parse_entries(&c, "token_bytes", tok, 1);
        else if (!strcmp(key, "special_tokens")) parse_entries(&c, "token_str", tok, 0);
        else jskip(&c);
    }
    free(json);
    if (vox_verbose >= 2)
        fprintf(stderr, "Tokenizer: %d vocab + %d special tokens\n", tok->n_vocab, tok->n_special);
    return tok;
}

void vox_tokenizer_free(vox_tokenizer_t *tok) {
    if (!tok) return;
    for (int i = 0; i < tok->cap_vocab; i++) free(tok->vocab[i]);
    free(tok->vocab);
    if (tok->special) for (int i = 0; i < TEKKEN_SPECIAL; i++) free(tok->special[i]);
    free(tok->special);
    free(tok);
}

const char *vox_tokenizer_decode(vox_tokenizer_t *tok, int id) {
    if (!tok) return NULL;
    if (id >= TEKKEN_SPECIAL && id < TEKKEN_SPECIAL + tok->n_vocab) return tok->vocab[id - TEKKEN_SPECIAL];
    if (id >= 0 && id < tok->n_special) return tok->special[id];
    return NULL;
}

char *vox_tokenizer_decode_seq(vox_tokenizer_t *tok, const int *ids, int n) {
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (ids[i] >= 0 && ids[i] < TEKKEN_SPECIAL) continue;   /* control tokens carry no text */
        const char *s = vox_tokenizer_decode(tok, ids[i]);
        if (s) total += strlen(s);
    }
    char *out = (char *)malloc(total + 1);
    size_t o = 0;
    for (int i = 0; i < n; i++) {
        if (ids[i] >= 0 && ids[i] < TEKKEN_SPECIAL) continue;
        const char *s = vox_tokenizer_decode(tok, ids[i]);
        if (s) { size_t l = strlen(s); memcpy(out + o, s, l); o += l; }
    }
    out[o] = 0;
    return out;
}

int vox_tokenizer_bos(vox_tokenizer_t *tok) { (void)tok; return 1; }
int vox_tokenizer_eos(vox_tokenizer_t *tok) { (void)tok; return 2; }
int vox_tokenizer_vocab_size(vox_tokenizer_t *tok) { (void)tok; return TEKKEN_SPECIAL + 130072; }
