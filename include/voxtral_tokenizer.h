/*
 * voxtral_tokenizer.h — Tekken tokenizer, decode only (reference voxtral_tokenizer.h:16-34).
 * ids 0..999 are special tokens, id >= 1000 maps to vocab[id-1000].token_bytes.
 */
#ifndef VOXTRAL_TOKENIZER_H
#define VOXTRAL_TOKENIZER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct vox_tokenizer vox_tokenizer_t;
vox_tokenizer_t *vox_tokenizer_load(const char *path);
void vox_tokenizer_free(vox_tokenizer_t *tok);
const char *vox_tokenizer_decode(vox_tokenizer_t *tok, int token_id);
char *vox_tokenizer_decode_seq(vox_tokenizer_t *tok, const int *tokens, int n_tokens);
int vox_tokenizer_bos(vox_tokenizer_t *tok);
int vox_tokenizer_eos(vox_tokenizer_t *tok);
int vox_tokenizer_vocab_size(vox_tokenizer_t *tok);
#ifdef __cplusplus
}
#endif
#endif
