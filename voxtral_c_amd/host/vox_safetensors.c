/* vox_safetensors.c — safetensors header index (see vox_safetensors.h). */
#include "vox_safetensors.h"
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

/* --- a very small JSON cursor: just enough for {"name":{"dtype":..,"shape":[..],
 *     "data_offsets":[a,b]}, "__metadata__":{...}} ------------------------------- */
typedef struct { const char *p, *end; int bad; } cur_t;

static void ws(cur_t *c) {
    while (c->p < c->end && (*c->p == ' ' || *c->p == '\n' || *c->p == '\t' || *c->p == '\r')) c->p++;
}
static int eat(cur_t *c, char ch) {
    ws(c);
    if (c->p < c->end && *c->p == ch) { c->p++; return 1; }
    return 0;
}
/* Returns a malloc'd, unescaped copy of the string at the cursor. */
static char *str(cur_t *c) {
    ws(c);
    if (c->p >= c->end || *c->p != '"') { c->bad = 1; return NULL; }
    const char *s = ++c->p;
    size_t n = 0;
    while (c->p < c->end && *c->p != '"') { if (*c->p == '\\' && c->p + 1 < c->end) c->p++; c->p++; n++; }
    if (c->p >= c->end) { c->bad = 1; return NULL; }
    char *out = (char *)malloc(n + 1);
    size_t o = 0;
    for (const char *q = s; q < c->p; q++) {
        if (*q == '\\' && q + 1 < c->p) {
            q++;
            out[o++] = (*q == 'n') ? '\n' : (*q == 't') ? '\t' : (*q == 'r') ? '\r' : *q;
        } else out[o++] = *q;
    }
    out[o] = 0;
    c->p++;
    return out;
}
static int64_t num(cur_t *c) {
    ws(c);
    int neg = 0; int64_t v = 0;
    if (c->p < c->end && *c->p == '-') { neg = 1; c->p++; }
    if (c->p >= c->end || *c->p < '0' || *c->p > '9') { c->bad = 1; return 0; }
    while (c->p < c->end && *c->p >= '0' && *c->p <= '9') v = v * 10 + (*c->p++ - '0');
    return neg ? -v : v;
}
static void skip(cur_t *c) {   /* skip any JSON value */
    ws(c);
    if (c->p >= c->end) { c->bad = 1; return; }
    if (*c->p == '"') { free(str(c)); return; }
    if (*c->p == '{' || *c->p == '[') {
        int depth = 0;
        do {
            if (*c->p == '"') { free(str(c)); continue; }
            if (*c->p == '{' || *c->p == '[') depth++;
            else if (*c->p == '}' || *c->p == ']') depth--;
            c->p++;
        } while (c->p < c->end && depth > 0);
        return;
    }
    while (c->p < c->end && *c->p != ',' && *c->p != '}' && *c->p != ']') c->p++;
}

static int cmp_name(const void *a, const void *b) {
    return strcmp(((const vox_st_tensor_t *)a)->name, ((const vox_st_tensor_t *)b)->name);
}

vox_st_file_t *vox_st_open(const char *path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { perror(path); return NULL; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); fprintf(stderr, "%s: not a safetensors file\n", path); return NULL; }
    void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { perror("mmap"); return NULL; }
    uint64_t hlen;
    memcpy(&hlen, map, 8);
    if (hlen > (uint64_t)st.st_size - 8) { munmap(map, (size_t)st.st_size); fprintf(stderr, "%s: bad header length\n", path); return NULL; }
    const uint8_t *payload = (const uint8_t *)map + 8 + hlen;
    const size_t payload_size = (size_t)st.st_size - 8 - (size_t)hlen;

    vox_st_file_t *f = (vox_st_file_t *)calloc(1, sizeof *f);
    f->map = map; f->map_size = (size_t)st.st_size;
    int cap = 0;
    cur_t c = {(const char *)map + 8, (const char *)map + 8 + hlen, 0};
    if (!eat(&c, '{')) c.bad = 1;
    while (!c.bad) {
        ws(&c);
        if (eat(&c, '}')) break;
        if (eat(&c, ',')) continue;
        char *name = str(&c);
        if (!name || !eat(&c, ':')) { free(name); c.bad = 1; break; }
        if (strcmp(name, "__metadata__") == 0) { free(name); skip(&c); continue; }
        vox_st_tensor_t t; memset(&t, 0, sizeof t);
        t.name = name; t.dtype = VOX_ST_OTHER;
        int64_t off0 = 0, off1 = 0;
        if (!eat(&c, '{')) { c.bad = 1; free(name); break; }
        while (!c.bad) {
            if (eat(&c, '}')) break;
            if (eat(&c, ',')) continue;
            char *key = str(&c);
            if (!key || !eat(&c, ':')) { free(key); c.bad = 1; break; }
            if (!strcmp(key, "dtype")) {
                char *d = str(&c);
                if (d) {
                    t.dtype = !strcmp(d, "BF16") ? VOX_ST_BF16 : !strcmp(d, "F16") ? VOX_ST_F16 :
                              !strcmp(d, "F32") ? VOX_ST_F32 : VOX_ST_OTHER;
                    free(d);
                }
            } else if (!strcmp(key, "shape")) {
                if (!eat(&c, '[')) c.bad = 1;
                while (!c.bad && !eat(&c, ']')) {
                    if (eat(&c, ',')) continue;
                    int64_t v = num(&c);
                    if (t.ndim < 8) t.shape[t.ndim++] = v;
                }
            } else if (!strcmp(key, "data_offsets")) {
                if (!eat(&c, '[')) c.bad = 1;
                off0 = num(&c); eat(&c, ','); off1 = num(&c);
                if (!eat(&c, ']')) c.bad = 1;
            } else skip(&c);
            free(key);
        }
        if (c.bad || off0 < 0 || off1 < off0 || (uint64_t)off1 > payload_size) {
            fprintf(stderr, "%s: tensor %s out of bounds\n", path, name);
            free(name); c.bad = 1; break;
        }
        t.data = payload + off0; t.nbytes = (size_t)(off1 - off0);
        {   /* shape and byte span must agree: vox_st_to_f32 and the uploads trust numel */
            int64_t n = 1; int okshape = 1;
            for (int i = 0; i < t.ndim; i++) {
                if (t.shape[i] < 0 || (t.shape[i] > 0 && n > ((int64_t)1 << 40) / t.shape[i])) { okshape = 0; break; }
                n *= t.shape[i];
            }
            const int64_t esz = (t.dtype == VOX_ST_F32) ? 4 : (t.dtype == VOX_ST_BF16 || t.dtype == VOX_ST_F16) ? 2 : 0;
            if (!okshape || (esz && (uint64_t)(n * esz) != (uint64_t)t.nbytes)) {
                fprintf(stderr, "%s: tensor %s: shape does not match its data span\n", path, name);
                free(name); c.bad = 1; break;
            }
        }
        if (f->n_tensors == cap) { cap = cap ? cap * 2 : 1024; f->tensors = (vox_st_tensor_t *)realloc(f->tensors, (size_t)cap * sizeof t); }
        f->tensors[f->n_tensors++] = t;
    }
    if (c.bad) { fprintf(stderr, "%s: cannot parse safetensors header\n", path); vox_st_close(f); return NULL; }
    if (f->n_tensors > 0) qsort(f->tensors, (size_t)f->n_tensors, sizeof(vox_st_tensor_t), cmp_name);
    return f;
}

void vox_st_close(vox_st_file_t *f) {
    if (!f) return;
    for (int i = 0; i < f->n_tensors; i++) free(f->tensors[i].name);
    free(f->tensors);
    if (f->map) munmap(f->map, f->map_size);
    free(f);
}

const vox_st_tensor_t *vox_st_find(const vox_st_file_t *f, const char *name) {
    int lo = 0, hi = f->n_tensors - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2, r = strcmp(name, f->tensors[mid].name);
        if (r == 0) return &f->tensors[mid];
        if (r < 0) hi = mid - 1; else lo = mid + 1;
    }
    return NULL;
}

int64_t vox_st_numel(const vox_st_tensor_t *t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; i++) n *= t->shape[i];
    return n;
}

static float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (!m) u = s;
        else { int sh = 0; while (!(m & 1024)) { m <<= 1; sh++; } u = s | ((uint32_t)(113 - sh) << 23) | ((m & 1023) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

float *vox_st_to_f32(const vox_st_tensor_t *t) {
    const int64_t n = vox_st_numel(t);
    float *out = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!out) return NULL;
    if (t->dtype == VOX_ST_F32) memcpy(out, t->data, (size_t)n * 4);
    else if (t->dtype == VOX_ST_BF16) {           /* memcpy reads: a hand-made header may leave the payload unaligned */
        const uint8_t *s = (const uint8_t *)t->data;
        for (int64_t i = 0; i < n; i++) { uint16_t h; memcpy(&h, s + 2 * i, 2); uint32_t u = (uint32_t)h << 16; memcpy(&out[i], &u, 4); }
    } else if (t->dtype == VOX_ST_F16) {
        const uint8_t *s = (const uint8_t *)t->data;
        for (int64_t i = 0; i < n; i++) { uint16_t h; memcpy(&h, s + 2 * i, 2); out[i] = half_to_float(h); }
    } else { free(out); return NULL; }
    return out;
}
