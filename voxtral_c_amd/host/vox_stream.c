/* vox_stream.c — the streaming transcription state machine.
 *
 * Same observable behaviour as the reference orchestrator (voxtral.c:413-1635): when the
 * encoder runs (312 mel frames for the first chunk, then the processing interval), how
 * the prompt is built, the greedy loop, token classification and queueing, alternative
 * tokens, flush/finish padding, the continuous-mode restart watchdogs and the stderr
 * statistics lines that benchmark.py parses.  What differs is where the data lives: the
 * reference shuttles mel frames, conv tails, encoder rows and adapter rows through host
 * buffers; here the host only keeps *counters* — all of those buffers are device
 * resident (csrc/vox_hip_engine.hip) and a feed() costs one H2D copy of audio samples and
 * one D2H copy of token ids.
 */
#include "vox_internal.h"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

extern int vox_verbose, vox_monitor;
void vox_kv_mirror_step(vox_ctx_t *ctx);
void vox_kv_mirror_prefill(vox_ctx_t *ctx, int seq_len);
void vox_enc_mirror_chunk(vox_ctx_t *ctx, int new_len);

/* Tekken special ids and stream policy constants (reference voxtral.c:362-386). */
enum { TOK_BOS = 1, TOK_EOS = 2, TOK_STREAMING_PAD = 32, TOK_TEXT_MIN = 1000 };
enum {
    SAMPLES_PER_TOKEN = 1280,
    RIGHT_PAD_BUFFER_TOKENS = 10,
    FIRST_CHUNK_MIN_MEL = 312,
    MAX_DECODE_KV = 2000,
    MAX_NONTEXT_STREAK = 64,
    MAX_NO_DECODE_SAMPLES = VOX_SAMPLE_RATE * 20,
    EMPTY_RESTARTS_FOR_FULL_RESET = 2,
    LEFT_PAD_TOKENS = 32
};
#define DEFAULT_INTERVAL_S 2.0f

typedef enum { CLS_TEXT, CLS_CONTROL, CLS_INVALID, CLS_EOS } tok_class_t;

struct vox_stream {
    vox_ctx_t *ctx;
    vox_hip_engine_t *eng;
    vox_tokenizer_t *tok;
    vox_mel_ctx_t *mel;
    int64_t samples_fed;
    int mel_cursor;             /* mel frames handed to the encoder so far */
    int stem_started;           /* first encoder chunk done */
    int min_new_mel;

    /* adapter rows are addressed logically; the rows themselves are on the device */
    int total_adapter, adapter_base;
    int dec_started, gen_pos, prev_token, eos_seen;
    int nontext_streak, text_since_restart, empty_restarts, waiting_prompt;
    int64_t last_decode_sample;
    int finished, continuous;
    int failed;                 /* a device-side failure left the stream's state inconsistent: feed / flush / finish return -1 from now on */

    /* ring of pending token strings, VOX_MAX_ALT slots per position */
    const char **q;
    int q_head, q_tail, q_cap;
    int n_alt;
    float alt_cutoff;

    float *logits;              /* host copy, only used in alt / recording mode */
    int *tok_scratch; int tok_scratch_cap;

    double enc_ms, dec_ms, prefill_ms;
    int n_generated, n_text;

    /* extensions (tests, tooling): opt-in id history, logit recording, teacher forcing */
    int *ids; int n_ids, ids_cap, ids_on;
    float *rec; int rec_rows, rec_cap;
    int n_steps;                /* decoder steps since vox_stream_init (restarts do not reset it) */
    const int *forced; int n_forced;
};

static double now_ms(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

static tok_class_t classify(vox_stream_t *s, int id) {
    if (id == TOK_EOS) return CLS_EOS;
    if (id < TOK_TEXT_MIN) return CLS_CONTROL;
    const char *p = vox_tokenizer_decode(s->tok, id);
    return (p && p[0]) ? CLS_TEXT : CLS_INVALID;    /* id 1000 = NUL byte = empty piece */
}

static void q_push(vox_stream_t *s, const char *alts[VOX_MAX_ALT]) {
    int next = (s->q_tail + 1) % s->q_cap;
    if (next == s->q_head) {
        const int ncap = s->q_cap * 2;
        const char **nq = (const char **)calloc((size_t)ncap * VOX_MAX_ALT, sizeof(char *));
        if (!nq) { fprintf(stderr, "vox_stream: out of memory growing the token queue; a token was dropped\n"); return; }
        int n = 0;
        for (int i = s->q_head; i != s->q_tail; i = (i + 1) % s->q_cap, n++)
            memcpy(&nq[n * VOX_MAX_ALT], &s->q[i * VOX_MAX_ALT], VOX_MAX_ALT * sizeof(char *));
        free(s->q);
        s->q = nq; s->q_head = 0; s->q_tail = n; s->q_cap = ncap;
        next = (s->q_tail + 1) % s->q_cap;
    }
    memcpy(&s->q[s->q_tail * VOX_MAX_ALT], alts, VOX_MAX_ALT * sizeof(char *));
    s->q_tail = next;
}

/* Alternatives for one position from the host copy of the logits (voxtral.c:911-966):
 * softmax, then up to n_alt-1 next-best text tokens whose 1 - p/p_best <= cutoff. */
static void fill_alts(vox_stream_t *s, int best, const char *alts[VOX_MAX_ALT]) {
    memset(alts, 0, VOX_MAX_ALT * sizeof(char *));
    alts[0] = vox_tokenizer_decode(s->tok, best);
    if (s->n_alt <= 1 || !s->logits) return;
    const int V = s->ctx->dims.vocab;
    float *p = s->logits;
    float mx = p[0];
    for (int i = 1; i < V; i++) if (p[i] > mx) mx = p[i];
    float sum = 0;
    for (int i = 0; i < V; i++) { p[i] = expf(p[i] - mx); sum += p[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < V; i++) p[i] *= inv;
    const float pbest = p[best];
    if (pbest <= 0) return;
    int picked[VOX_MAX_ALT], found = 1;
    picked[0] = best;
    while (found < s->n_alt) {
        int arg = -1; float pv = -1;
        for (int i = TOK_TEXT_MIN; i < V; i++) {
            if (i == best) continue;
            int dup = 0;
            for (int j = 1; j < found; j++) if (picked[j] == i) { dup = 1; break; }
            if (!dup && p[i] > pv) { pv = p[i]; arg = i; }
        }
        if (arg < 0 || 1.0f - pv / pbest > s->alt_cutoff) break;
        picked[found] = arg;
        alts[found++] = vox_tokenizer_decode(s->tok, arg);
    }
}

static int want_logits(const vox_stream_t *s) { return s->n_alt > 1 || s->rec_cap > 0; }
/* the host must see every token before the next step is enqueued */
static int step_by_step(const vox_stream_t *s) { return want_logits(s) || s->n_steps < s->n_forced; }

/* Records the engine's own greedy id of this step (opt-in) and returns the id the stream carries
 * forward: the engine's, or the caller's under teacher forcing (vox_stream_force_tokens). */
static int note_token(vox_stream_t *s, int id) {
    if (s->ids_on) {
        if (s->n_ids == s->ids_cap) {
            const int nc = s->ids_cap ? s->ids_cap * 2 : 1024;
            int *t = (int *)realloc(s->ids, (size_t)nc * sizeof(int));
            if (t) { s->ids = t; s->ids_cap = nc; }
        }
        if (s->n_ids < s->ids_cap) s->ids[s->n_ids++] = id;
    }
    if (s->rec_cap > 0 && s->rec_rows < s->rec_cap && s->logits)
        memcpy(s->rec + (size_t)s->rec_rows++ * s->ctx->dims.vocab, s->logits, (size_t)s->ctx->dims.vocab * sizeof(float));
    const int step = s->n_steps++;
    return step < s->n_forced ? s->forced[step] : id;
}

/* Book-keeping shared by the prefill token and every loop token. Returns the class; *id becomes
 * the token the stream carries forward (differs from the engine's only under teacher forcing). */
static tok_class_t account_token(vox_stream_t *s, int *idp) {
    s->n_generated++;
    s->last_decode_sample = s->samples_fed;
    const int id = *idp = note_token(s, *idp);
    const tok_class_t cls = classify(s, id);
    if (cls == CLS_TEXT) {
        const char *alts[VOX_MAX_ALT];
        fill_alts(s, id, alts);
        if (alts[0]) {
            q_push(s, alts);
            s->n_text++;
            s->text_since_restart = 1;
            s->empty_restarts = 0;
        }
        s->nontext_streak = 0;
    }
    return cls;
}

/* ---- resets (reference voxtral.c:734-780) ------------------------------------------ */
static void reset_decoder_state(vox_stream_t *s) {
    vox_hip_reset_decoder(s->eng);
    s->ctx->kv_cache_len = 0; s->ctx->kv_pos_offset = 0;
    s->total_adapter = 0; s->adapter_base = 0; s->gen_pos = 0;
    s->dec_started = 0; s->prev_token = TOK_BOS; s->eos_seen = 0;
    s->n_generated = 0; s->nontext_streak = 0; s->text_since_restart = 0; s->waiting_prompt = 0;
}

static int reset_full_state(vox_stream_t *s) {
    vox_mel_ctx_t *nm = vox_mel_ctx_init_engine(s->eng, LEFT_PAD_TOKENS * SAMPLES_PER_TOKEN, 1);
    if (!nm) return -1;
    vox_hip_reset_encoder(s->eng);
    vox_mel_free(s->mel);
    s->mel = nm;
    s->mel_cursor = 0;
    s->stem_started = 0;
    s->ctx->enc_kv_cache_len = 0; s->ctx->enc_kv_pos_offset = 0;
    reset_decoder_state(s);
    return 0;
}

/* ---- encoder side (reference stream_run_encoder, voxtral.c:783-907) ------------------ */
static void run_encoder(vox_stream_t *s) {
    const int total_mel = vox_mel_total_frames(s->mel);
    const int new_mel = total_mel - s->mel_cursor;
    const int need = s->stem_started ? s->min_new_mel : FIRST_CHUNK_MIN_MEL;
    if (new_mel < need && !s->finished) return;
    if (new_mel <= 0) return;

    const double t0 = now_ms();
    int new_mel_left = new_mel;
    if (s->ctx->n_shard_engines > 1 && !__atomic_load_n(&s->ctx->shard_disabled, __ATOMIC_RELAXED)) {
        /* a multi-device model: a large chunk - a stream's first one or any later one that starts on a token boundary - is
         * encoded by all GPUs together (vox_multi.c); whatever is left (< 8 frames) waits for the next chunk, or runs on the
         * stream's own engine as usual when the stream is being finished.  (main.c -i feeds files in 1 s pieces: after the
         * first chunk of ~3 s such a stream never has 16 tokens per engine in one chunk - it cannot benefit; one-feed clients,
         * vox_transcribe_audio and large processing intervals do.) */
        int mt = 0;
        const int used = vox_multi_encode_chunk(s->ctx, new_mel, &mt);
        if (used < 0) {
            /* By now the mel queue of the stream engine may be half consumed and adapter rows are accounted for that
             * were never written: there is nothing consistent left to retry on (the device mel queue is the only copy
             * of the frames).  Fail the stream for good - every later feed / flush / finish returns -1, like the
             * reference's error convention (voxtral.c:1237).  Later streams of this model do not shard (one flag write,
             * read at the start of run_encoder: no engine is torn down under a stream that may still use it). */
            fprintf(stderr, "vox_stream: sharded encoder failed (%s); this stream is dead, later streams of this model run on one GPU\n",
                    vox_hip_last_error());
            s->failed = 1;
            __atomic_store_n(&s->ctx->shard_disabled, 1, __ATOMIC_RELAXED);
            return;
        }
        if (used > 0) {
            /* the shards are only enqueued, and the stream engine does not wait for the others here: its decoder waits for each
             * shard's adapter rows when it reaches them (vox_hip_shard_end_push).  The encoder time accounted here is the host's
             * enqueue time; VOX_HIP_DISABLE=multi_overlap waits for the whole wavefront as round 3 did. */
            if (vox_hip_switch_disabled("multi_overlap")) vox_hip_sync(s->eng);
            vox_hip_add_encode_ms(s->eng, now_ms() - t0);
            __atomic_fetch_add(&s->ctx->n_sharded_chunks, 1, __ATOMIC_RELAXED);
            s->mel_cursor += used;
            s->stem_started = 1;
            vox_enc_mirror_chunk(s->ctx, used / 2);
            s->total_adapter += mt;
            new_mel_left = new_mel - used;
            if (vox_verbose >= 2)
                fprintf(stderr, "  Encoder sharded over %d GPUs: %d mel -> %d tokens (total adapter: %d)\n", s->ctx->n_shard_engines, used, mt, s->total_adapter);
            if (new_mel_left <= 0 || (new_mel_left < s->min_new_mel && !s->finished)) { s->enc_ms += now_ms() - t0; return; }
        }
    }
    int conv_rows = 0, residual = 0;
    const int new_tokens = vox_hip_stream_encode(s->eng, new_mel_left, &conv_rows, &residual);
    s->mel_cursor = total_mel;
    s->stem_started = 1;
    if (new_tokens < 0) {
        /* the device mel queue was the only copy of these frames and the encoder state is undefined now: the stream is dead
         * (feed / flush / finish return -1 from here on, the reference's error convention, voxtral.c:1237) */
        fprintf(stderr, "vox_stream: encoder failed: %s\n", vox_hip_last_error());
        s->failed = 1;
        return;
    }
    if (conv_rows <= 0) return;
    vox_enc_mirror_chunk(s->ctx, conv_rows);
    s->total_adapter += new_tokens;
    s->enc_ms += now_ms() - t0;
    if (vox_monitor) { fputs("\xe2\x96\xb6", stderr); fflush(stderr); }      /* ▶ encoder chunk */
    if (vox_verbose >= 2)
        fprintf(stderr, "  Encoder inc: %d mel -> %d conv -> %d usable (total adapter: %d, residual: %d)\n",
                new_mel_left, conv_rows, new_tokens * VOX_DOWNSAMPLE, s->total_adapter, residual);
}

/* ---- decoder side (reference stream_run_decoder, voxtral.c:969-1188) ----------------- */
static void run_decoder(vox_stream_t *s) {
    vox_ctx_t *ctx = s->ctx;
    const int prompt_len = 1 + LEFT_PAD_TOKENS + ctx->delay_tokens;
    const int have = s->total_adapter - s->adapter_base;

    if (!s->dec_started && have < prompt_len) {
        if (vox_monitor && !s->waiting_prompt) { fputs("\xe2\x8c\x9b", stderr); fflush(stderr); s->waiting_prompt = 1; }  /* ⌛ */
        return;
    }
    if (!s->dec_started) {
        s->waiting_prompt = 0;
        const double t0 = now_ms();
        extern void vox_hip_reset_decoder_kv(vox_hip_engine_t *);
        vox_hip_reset_decoder_kv(s->eng);
        ctx->kv_cache_len = 0; ctx->kv_pos_offset = 0;
        int tok = vox_hip_decoder_prefill_stream(s->eng, s->adapter_base, prompt_len, TOK_BOS, TOK_STREAMING_PAD,
                                                 want_logits(s) ? s->logits : NULL);
        if (tok < 0) { fprintf(stderr, "vox_stream: prefill failed: %s\n", vox_hip_last_error()); s->failed = 1; return; }
        vox_kv_mirror_prefill(ctx, prompt_len - 1);
        vox_kv_mirror_step(ctx);
        const tok_class_t cls = account_token(s, &tok);
        s->prev_token = tok;
        if (cls != CLS_TEXT && cls != CLS_EOS) s->nontext_streak++;
        if (tok == TOK_EOS) s->eos_seen = 1;
        s->gen_pos = s->adapter_base + prompt_len;
        s->dec_started = 1;
        const double dt = now_ms() - t0;
        s->dec_ms += dt; s->prefill_ms += dt;
        if (vox_monitor) { fputs("\xc2\xb7", stderr); fflush(stderr); }      /* · prefill */
    }

    if (s->dec_started && !s->eos_seen && s->gen_pos < s->total_adapter) {
        const double t0 = now_ms();
        const int before = s->n_generated;
        int n_text = 0, n_ctrl = 0, n_inval = 0, saw_eos = 0;
        const int V = ctx->dims.vocab;
        while (s->gen_pos < s->total_adapter && !saw_eos) {
            /* In plain mode the whole backlog is queued on the device in one go (the
             * previous token never leaves HBM); with alternatives / logit recording the
             * host needs each row, so go one position at a time. */
            int batch = s->total_adapter - s->gen_pos;
            if (step_by_step(s)) batch = 1;
            if (batch > s->tok_scratch_cap) {
                free(s->tok_scratch);
                s->tok_scratch = (int *)malloc((size_t)batch * sizeof(int));
                s->tok_scratch_cap = s->tok_scratch ? batch : 0;
                if (!s->tok_scratch) { s->failed = 1; return; }
            }
            const int got = vox_hip_decoder_run(s->eng, s->gen_pos, batch, s->prev_token, TOK_EOS, s->tok_scratch,
                                                want_logits(s) ? s->logits : NULL);
            if (got <= 0) {          /* the KV ring and the position counters no longer describe the same sequence */
                fprintf(stderr, "vox_stream: decode failed: %s\n", vox_hip_last_error());
                s->failed = 1;
                return;
            }
            (void)V;
            for (int i = 0; i < got; i++) {
                int tok = s->tok_scratch[i];
                vox_kv_mirror_step(ctx);
                const tok_class_t cls = account_token(s, &tok);
                s->prev_token = tok;
                if (cls == CLS_TEXT) n_text++;
                else if (cls == CLS_CONTROL) { s->nontext_streak++; n_ctrl++; }
                else if (cls == CLS_INVALID) { s->nontext_streak++; n_inval++; }
                s->gen_pos++;
                if (tok == TOK_EOS) { s->eos_seen = 1; saw_eos = 1; break; }
            }
        }
        if (s->n_generated > before) {
            const double dt = now_ms() - t0;
            s->dec_ms += dt;
            if (vox_monitor) {
                const int steps = s->n_generated - before;
                const int slow = dt / steps > 40;
                const char *sym, *sev = "";
                if (n_text > 0) sym = slow ? "\xe2\x96\xb8" : "\xe2\x96\xaa";           /* ▸ / ▪ */
                else if (n_inval > 0) sym = slow ? "\xe2\x9c\x98" : "\xe2\x9c\x97";     /* ✘ / ✗ */
                else if (n_ctrl > 0) sym = slow ? "\xe2\x96\xb9" : "\xe2\x96\xab";      /* ▹ / ▫ */
                else if (saw_eos) sym = "\xe2\x97\xa6";                                /* ◦ */
                else sym = "\xe2\x96\xaa";
                if (n_text == 0 && (n_ctrl > 0 || n_inval > 0)) {
                    if (s->nontext_streak >= MAX_NONTEXT_STREAK - 8) sev = "\xe2\x98\xa0";       /* ☠ */
                    else if (s->nontext_streak >= MAX_NONTEXT_STREAK / 2) sev = "\xe2\x9a\xa0";  /* ⚠ */
                }
                fprintf(stderr, "%s%s", sym, sev);
                fflush(stderr);
            }
        }
    }

    /* rows the decoder has consumed are dead (stream_adapter_compact, voxtral.c:718-731) */
    if (s->gen_pos > s->adapter_base) s->adapter_base = s->gen_pos;

    /* continuous-mode watchdogs (voxtral.c:1137-1187) */
    int why = 0;
    if (s->continuous) {
        if (s->eos_seen) why = 1;
        else if (s->dec_started && ctx->kv_cache_len > MAX_DECODE_KV) why = 2;
        else if (s->dec_started && s->nontext_streak >= MAX_NONTEXT_STREAK) why = 3;
        else if (!s->finished && s->samples_fed - s->last_decode_sample >= MAX_NO_DECODE_SAMPLES) why = 4;
    }
    if (why) {
        if (s->text_since_restart) s->empty_restarts = 0; else s->empty_restarts++;
        const int full = why >= 2 || s->empty_restarts >= EMPTY_RESTARTS_FOR_FULL_RESET;
        if (vox_monitor) {
            const char *sym = why == 1 ? "\xe2\x86\xba" : why == 2 ? "\xe2\x9f\xb3" : why == 3 ? "\xe2\x86\xaf" : "\xe2\x8c\x9a";
            fprintf(stderr, "%s%s", sym, full ? "\xe2\x99\xbb" : "\xe2\x9c\x82");
            fflush(stderr);
        }
        if (full) {
            if (reset_full_state(s) != 0) reset_decoder_state(s);
            s->empty_restarts = 0;
        } else reset_decoder_state(s);
        s->last_decode_sample = s->samples_fed;
    }
}

/* ---- public API ----------------------------------------------------------------------- */
vox_stream_t *vox_stream_init(vox_ctx_t *ctx) {
    if (!ctx || !ctx->engine) return NULL;
    vox_stream_t *s = (vox_stream_t *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->ctx = ctx;
    s->eng = (vox_hip_engine_t *)ctx->engine;
    /* The reference parses tekken.json in every vox_stream_init (voxtral.c:1197-1199); the table is
     * immutable, so the model keeps it (10+ ms per stream for the 131072-entry vocabulary). */
    if (!ctx->tokenizer) {
        char path[1024];
        snprintf(path, sizeof path, "%s/tekken.json", ctx->model_dir);
        ctx->tokenizer = vox_tokenizer_load(path);
    }
    s->tok = (vox_tokenizer_t *)ctx->tokenizer;
    if (!s->tok) { free(s); return NULL; }
    vox_hip_reset_encoder(s->eng);
    vox_hip_reset_decoder(s->eng);
    vox_hip_reset_timing(s->eng);
    ctx->enc_kv_cache_len = 0; ctx->enc_kv_pos_offset = 0;
    s->mel = vox_mel_ctx_init_engine(s->eng, LEFT_PAD_TOKENS * SAMPLES_PER_TOKEN, 1);
    s->q_cap = 256;
    s->q = (const char **)calloc((size_t)s->q_cap * VOX_MAX_ALT, sizeof(char *));
    s->logits = (float *)malloc((size_t)ctx->dims.vocab * sizeof(float));
    s->n_alt = 1;
    s->prev_token = TOK_BOS;
    s->min_new_mel = (int)(DEFAULT_INTERVAL_S * 100.0f);
    if (!s->mel || !s->q || !s->logits) { vox_stream_free(s); return NULL; }
    return s;
}

int vox_stream_feed(vox_stream_t *s, const float *samples, int n_samples) {
    if (!s || s->finished || s->failed || n_samples <= 0) return -1;
    vox_mel_feed(s->mel, samples, n_samples);
    s->samples_fed += n_samples;
    run_encoder(s);
    if (s->failed) return -1;
    run_decoder(s);
    return s->failed ? -1 : 0;
}

/* right padding: align to a token, then (delay+1) + 10 tokens of silence (voxtral.c:1593-1606) */
static void feed_right_padding(vox_stream_t *s) {
    const int align = (int)((SAMPLES_PER_TOKEN - (s->samples_fed % SAMPLES_PER_TOKEN)) % SAMPLES_PER_TOKEN);
    int remaining = align + ((s->ctx->delay_tokens + 1) + RIGHT_PAD_BUFFER_TOKENS) * SAMPLES_PER_TOKEN;
    static const float zeros[4096] = {0};
    while (remaining > 0) {
        const int n = remaining > 4096 ? 4096 : remaining;
        vox_mel_feed(s->mel, zeros, n);
        remaining -= n;
    }
}

int vox_stream_flush(vox_stream_t *s) {
    if (!s || s->finished || s->failed) return -1;
    feed_right_padding(s);
    const int saved = s->min_new_mel;
    s->min_new_mel = 1;
    run_encoder(s);
    if (!s->failed) run_decoder(s);
    s->min_new_mel = saved;
    return s->failed ? -1 : 0;
}

int vox_stream_finish(vox_stream_t *s) {
    if (!s || s->finished || s->failed) return -1;
    /* The reference flushes (padding -> encoder -> decoder) and then finishes the mel (-> encoder on the last frame -> decoder)
     * (voxtral.c:1608-1625).  Encoder and decoder outputs do not depend on how the frames are cut into chunks, so outside
     * continuous mode - whose watchdogs look at the state after every decoder run - the padding and the mel tail go through
     * ONE encoder pass and one decoder run: the separate 1-row encoder pass was 273 launches = 2.4 ms of pure launch latency. */
    if (s->continuous) vox_stream_flush(s);
    else feed_right_padding(s);
    s->finished = 1;
    vox_mel_finish(s->mel, 0);
    if (vox_verbose >= 2)
        fprintf(stderr, "Stream finished: %lld real samples (%.1f sec)\n", (long long)s->samples_fed,
                (double)s->samples_fed / VOX_SAMPLE_RATE);
    run_encoder(s);
    if (s->failed) return -1;
    run_decoder(s);
    return s->failed ? -1 : 0;
}

int vox_stream_get(vox_stream_t *s, const char **out, int max) {
    if (!s || max <= 0) return 0;
    int n = 0;
    while (n < max && s->q_head != s->q_tail) {
        out[n++] = s->q[s->q_head * VOX_MAX_ALT];
        s->q_head = (s->q_head + 1) % s->q_cap;
    }
    return n;
}

void vox_stream_set_alt(vox_stream_t *s, int n_alt, float cutoff) {
    if (!s) return;
    s->n_alt = n_alt < 1 ? 1 : n_alt > VOX_MAX_ALT ? VOX_MAX_ALT : n_alt;
    s->alt_cutoff = cutoff < 0 ? 0 : cutoff > 1 ? 1 : cutoff;
}

int vox_stream_get_alt(vox_stream_t *s, const char **out, int max_tokens, int n_alt) {
    if (!s || max_tokens <= 0 || n_alt <= 0) return 0;
    if (n_alt > VOX_MAX_ALT) n_alt = VOX_MAX_ALT;
    int n = 0;
    while (n < max_tokens && s->q_head != s->q_tail) {
        for (int a = 0; a < n_alt; a++) out[n * n_alt + a] = s->q[s->q_head * VOX_MAX_ALT + a];
        n++;
        s->q_head = (s->q_head + 1) % s->q_cap;
    }
    return n;
}

void vox_set_processing_interval(vox_stream_t *s, float seconds) {
    if (!s) return;
    if (seconds <= 0) seconds = 0;
    s->min_new_mel = (int)(seconds * 100.0f);    /* 100 mel frames per second */
    if (s->min_new_mel < 1) s->min_new_mel = 1;
}

void vox_stream_set_continuous(vox_stream_t *s, int enable) { if (s) s->continuous = enable; }

void vox_stream_free(vox_stream_t *s) {
    if (!s) return;
    if (vox_verbose >= 1) {   /* format parsed by the reference's benchmark.py:25-30 */
        fprintf(stderr, "Encoder: %d mel -> %d tokens (%.0f ms)\n", s->mel_cursor, s->total_adapter, s->enc_ms);
        if (s->n_text > 0) {
            const double gen = s->dec_ms - s->prefill_ms;
            fprintf(stderr, "Decoder: %d text tokens (%d steps) in %.0f ms (prefill %.0f ms + %.1f ms/step)\n",
                    s->n_text, s->n_generated, s->dec_ms, s->prefill_ms,
                    s->n_generated > 1 ? gen / (s->n_generated - 1) : 0);
        }
    }
    vox_mel_free(s->mel);
    free(s->q); free(s->logits); free(s->tok_scratch); free(s->ids); free(s->rec);
    free(s);
}

int vox_stream_token_ids(vox_stream_t *s, int *out_ids, int max) {
    if (!s) return 0;
    const int n = s->n_ids < max ? s->n_ids : max;
    if (out_ids && n > 0) memcpy(out_ids, s->ids, (size_t)n * sizeof(int));
    return out_ids ? n : s->n_ids;
}
void vox_stream_record_ids(vox_stream_t *s, int enable) { if (s) s->ids_on = enable != 0; }
void vox_stream_force_tokens(vox_stream_t *s, const int *ids, int n) {
    if (!s) return;
    s->forced = ids; s->n_forced = ids && n > 0 ? n : 0;
}
void vox_stream_record_logits(vox_stream_t *s, int max_rows) {
    if (!s || max_rows <= 0) return;
    free(s->rec);
    s->rec = (float *)malloc((size_t)max_rows * s->ctx->dims.vocab * sizeof(float));
    s->rec_cap = s->rec ? max_rows : 0;
    s->rec_rows = 0;
}
int vox_stream_recorded_logits(vox_stream_t *s, const float **rows_out) {
    if (!s) return 0;
    if (rows_out) *rows_out = s->rec;
    return s->rec_rows;
}

/* ---- convenience wrappers (reference voxtral.c:1338-1586) ------------------------------ */
static void trim(char *t) {
    size_t n = strlen(t), a = 0;
    while (a < n && isspace((unsigned char)t[a])) a++;
    while (n > a && isspace((unsigned char)t[n - 1])) n--;
    memmove(t, t + a, n - a);
    t[n - a] = 0;
}

typedef struct { char *p; size_t len, cap; int oom; } sbuf_t;
static int sb_init(sbuf_t *b) {
    b->p = (char *)malloc(1024); b->len = 0; b->cap = 1024; b->oom = 0;
    if (!b->p) return -1;
    b->p[0] = 0;
    return 0;
}
static void sb_drain(sbuf_t *b, vox_stream_t *s) {
    const char *tk[64];
    int n;
    while ((n = vox_stream_get(s, tk, 64)) > 0)
        for (int i = 0; i < n && !b->oom; i++) {
            const size_t l = strlen(tk[i]);
            if (b->len + l + 1 > b->cap) {
                size_t nc = b->cap;
                while (b->len + l + 1 > nc) nc *= 2;
                char *t = (char *)realloc(b->p, nc);
                if (!t) { b->oom = 1; break; }
                b->p = t; b->cap = nc;
            }
            memcpy(b->p + b->len, tk[i], l);
            b->len += l;
            b->p[b->len] = 0;
        }
}
/* the transcript, or NULL (and nothing leaked) if memory ran out while collecting it */
static char *sb_finish(sbuf_t *b) {
    if (b->oom) { fprintf(stderr, "vox_transcribe: out of memory\n"); free(b->p); return NULL; }
    trim(b->p);
    return b->p;
}

char *vox_transcribe_audio(vox_ctx_t *ctx, const float *samples, int n_samples) {
    vox_stream_t *s = vox_stream_init(ctx);
    if (!s) return NULL;
    sbuf_t b;
    if (sb_init(&b)) { vox_stream_free(s); return NULL; }
    const int rc_feed = vox_stream_feed(s, samples, n_samples);
    const int rc_fin = vox_stream_finish(s);
    sb_drain(&b, s);
    vox_stream_free(s);
    if (n_samples > 0 && (rc_feed != 0 || rc_fin != 0)) {      /* a device failure: no partial transcript passed off as a whole one */
        fprintf(stderr, "vox_transcribe: the stream failed on the device\n");
        free(b.p);
        return NULL;
    }
    return sb_finish(&b);
}

char *vox_transcribe(vox_ctx_t *ctx, const char *wav_path) {
    int n = 0;
    float *smp = vox_load_wav(wav_path, &n);
    if (!smp) { fprintf(stderr, "vox_transcribe: cannot load %s\n", wav_path); return NULL; }
    if (vox_verbose >= 1) fprintf(stderr, "Audio: %d samples (%.1f seconds)\n", n, (float)n / VOX_SAMPLE_RATE);
    char *t = vox_transcribe_audio(ctx, smp, n);
    free(smp);
    return t;
}

char *vox_transcribe_stdin(vox_ctx_t *ctx) {
    uint8_t head[4];
    if (fread(head, 1, 4, stdin) < 4) { fprintf(stderr, "vox_transcribe_stdin: not enough data on stdin\n"); return NULL; }
    if (!memcmp(head, "RIFF", 4)) {
        /* WAV: slurp, parse, run as one offline feed */
        if (vox_verbose >= 2) fprintf(stderr, "Detected WAV format on stdin\n");
        size_t cap = 1 << 20, size = 4;
        uint8_t *buf = (uint8_t *)malloc(cap);
        if (!buf) return NULL;
        memcpy(buf, head, 4);
        for (;;) {
            if (size == cap) { cap *= 2; uint8_t *t = (uint8_t *)realloc(buf, cap); if (!t) { free(buf); return NULL; } buf = t; }
            const size_t n = fread(buf + size, 1, cap - size, stdin);
            if (!n) break;
            size += n;
        }
        if (vox_verbose >= 2) fprintf(stderr, "Read %zu bytes from stdin\n", size);
        int n = 0;
        float *smp = vox_parse_wav_buffer(buf, size, &n);
        free(buf);
        if (!smp) { fprintf(stderr, "Invalid WAV data on stdin\n"); return NULL; }
        if (vox_verbose >= 1) fprintf(stderr, "Audio: %d samples (%.1f seconds)\n", n, (float)n / VOX_SAMPLE_RATE);
        char *t = vox_transcribe_audio(ctx, smp, n);
        free(smp);
        return t;
    }
    /* raw s16le: stream it */
    if (vox_verbose >= 2) fprintf(stderr, "Streaming raw s16le 16kHz mono from stdin\n");
    vox_stream_t *s = vox_stream_init(ctx);
    if (!s) return NULL;
    {
        int16_t v[2];
        memcpy(v, head, 4);
        const float f[2] = {v[0] / 32768.0f, v[1] / 32768.0f};
        vox_stream_feed(s, f, 2);
    }
    sbuf_t b;
    if (sb_init(&b)) { vox_stream_free(s); return NULL; }
    int16_t raw[4096];
    float fb[4096];
    for (;;) {
        const size_t n = fread(raw, sizeof(int16_t), 4096, stdin);
        if (!n) { vox_stream_finish(s); sb_drain(&b, s); break; }
        for (size_t i = 0; i < n; i++) fb[i] = raw[i] / 32768.0f;
        vox_stream_feed(s, fb, (int)n);
        sb_drain(&b, s);
    }
    vox_stream_free(s);
    return sb_finish(&b);
}
