/* vox_kernels.c — the reference's kernel-level API (voxtral_kernels.h) on the GPU.
 *
 * Every function keeps the reference's name, arguments and host-buffer contract
 * (/root/reference/voxtral_kernels.h:18-159) and forwards to the matching vox_hip_k_* /
 * vox_hip_linear_bf16 / vox_hip_causal_attention entry point of libvoxhip.so on a default
 * engine-less device context.  No arithmetic happens on the host.
 */
#include "vox_internal.h"
#include "../../include/voxtral_kernels.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

enum { EW_ADD = 0, EW_MUL, EW_AXPY, EW_SCALE, EW_SILU, EW_GELU };   /* csrc/vox_kernel_api.h */
int vox_hip_k_eltwise(vox_hip_engine_t *e, float *a, const float *b, float s, size_t n, int op);
int vox_hip_k_sgemm(vox_hip_engine_t *e, float *C, const float *A, const float *B, const float *bias, int M, int K, int N, int b_is_nk);
int vox_hip_k_conv1d(vox_hip_engine_t *e, float *out, const float *in, const float *weight, const float *bias, int c_in, int c_out,
                     int length, int ks, int stride, int pad_left, int out_len);
int vox_hip_k_rms_norm(vox_hip_engine_t *e, float *out, const float *x, const float *w, int seq, int hidden, float eps);
int vox_hip_k_softmax(vox_hip_engine_t *e, float *x, int rows, int cols);
int vox_hip_k_rope_freqs(vox_hip_engine_t *e, float *freqs, const int *pos, int seq, int dim, float theta);
int vox_hip_k_apply_rope(vox_hip_engine_t *e, float *x, const float *freqs, int seq, int heads, int head_dim);

static vox_hip_engine_t *dev(const char *who) {
    vox_hip_engine_t *e = vox_default_mel_engine();
    if (!e) fprintf(stderr, "%s: no HIP device context (%s); this library has no CPU fallback\n", who, vox_hip_last_error());
    return e;
}
#define DEV() vox_hip_engine_t *e = dev(__func__); if (!e) return
#define CHECK(call) do { if ((call) != 0) fprintf(stderr, "%s failed: %s\n", __func__, vox_hip_last_error()); } while (0)

void vox_add_inplace(float *a, const float *b, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, a, b, 0.f, (size_t)n, EW_ADD)); }
void vox_mul_inplace(float *a, const float *b, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, a, b, 0.f, (size_t)n, EW_MUL)); }
void vox_axpy(float *a, float scale, const float *b, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, a, b, scale, (size_t)n, EW_AXPY)); }
void vox_scale(float *x, float s, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, x, NULL, s, (size_t)n, EW_SCALE)); }
void vox_copy(float *dst, const float *src, int n) { if (n > 0) memcpy(dst, src, (size_t)n * sizeof(float)); }   /* voxtral_kernels.c:45-47 */
void vox_silu(float *x, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, x, NULL, 0.f, (size_t)n, EW_SILU)); }
void vox_gelu(float *x, int n) { if (n <= 0) return; DEV(); CHECK(vox_hip_k_eltwise(e, x, NULL, 0.f, (size_t)n, EW_GELU)); }
void vox_softmax(float *x, int rows, int cols) { if (rows <= 0 || cols <= 0) return; DEV(); CHECK(vox_hip_k_softmax(e, x, rows, cols)); }

void vox_matmul(float *C, const float *A, const float *B, int M, int K, int N) { DEV(); CHECK(vox_hip_k_sgemm(e, C, A, B, NULL, M, K, N, 0)); }
void vox_matmul_t(float *C, const float *A, const float *B, int M, int K, int N) { DEV(); CHECK(vox_hip_k_sgemm(e, C, A, B, NULL, M, K, N, 1)); }
void vox_linear(float *y, const float *x, const float *W, const float *b, int seq_len, int in_dim, int out_dim) {
    DEV(); CHECK(vox_hip_k_sgemm(e, y, x, W, b, seq_len, in_dim, out_dim, 1));
}
void vox_linear_nobias(float *y, const float *x, const float *W, int seq_len, int in_dim, int out_dim) {
    vox_linear(y, x, W, NULL, seq_len, in_dim, out_dim);
}
/* bf16 weights: the production kernels (GEMV for one row, MFMA GEMM otherwise; voxtral_kernels.c:197-253) */
void vox_linear_bf16(float *y, const float *x, const uint16_t *W_bf16, const float *b, int seq_len, int in_dim, int out_dim) {
    DEV(); CHECK(vox_hip_linear_bf16(e, y, x, W_bf16, b, seq_len, in_dim, out_dim, 0));
}
void vox_linear_nobias_bf16(float *y, const float *x, const uint16_t *W_bf16, int seq_len, int in_dim, int out_dim) {
    vox_linear_bf16(y, x, W_bf16, NULL, seq_len, in_dim, out_dim);
}
void vox_matmul_t_bf16(float *C, const float *A, const uint16_t *B_bf16, int M, int K, int N) { vox_linear_bf16(C, A, B_bf16, NULL, M, K, N); }

void vox_conv1d(float *out, const float *in, const float *weight, const float *bias, int channels_in, int channels_out,
                int length, int kernel_size, int stride, int padding) {
    if (stride <= 0) return;
    const int out_length = (length + 2 * padding - kernel_size) / stride + 1;              /* voxtral_kernels.c:268 */
    if (out_length <= 0) return;
    DEV(); CHECK(vox_hip_k_conv1d(e, out, in, weight, bias, channels_in, channels_out, length, kernel_size, stride, padding, out_length));
}
void vox_causal_conv1d(float *out, const float *in, const float *weight, const float *bias, int channels_in, int channels_out,
                       int length, int kernel_size, int stride) {
    if (stride <= 0) return;
    const int padding_total = kernel_size - stride;                                         /* voxtral_kernels.c:298-301 */
    const float n_frames = ((float)length - kernel_size + padding_total) / (float)stride + 1.0f;
    const int out_length = (int)ceilf(n_frames);
    if (out_length <= 0) return;
    DEV(); CHECK(vox_hip_k_conv1d(e, out, in, weight, bias, channels_in, channels_out, length, kernel_size, stride, padding_total, out_length));
}

void vox_rms_norm(float *out, const float *x, const float *weight, int seq_len, int hidden, float eps) {
    DEV(); CHECK(vox_hip_k_rms_norm(e, out, x, weight, seq_len, hidden, eps));
}
void vox_causal_attention(float *out, const float *Q, const float *K, const float *V, int seq_q, int seq_k, int n_heads,
                          int n_kv_heads, int head_dim, float scale, int window_size, int q_offset) {
    if (seq_q <= 0 || seq_k <= 0) return;
    DEV(); CHECK(vox_hip_causal_attention(e, out, Q, K, V, seq_q, seq_k, n_heads, n_kv_heads, head_dim, scale, window_size, q_offset));
}
void vox_compute_rope_freqs(float *freqs, const int *pos, int seq, int dim, float theta) {
    if (seq <= 0) return;
    DEV(); CHECK(vox_hip_k_rope_freqs(e, freqs, pos, seq, dim, theta));
}
void vox_apply_rope(float *x, const float *freqs, int seq, int heads, int head_dim) {
    if (seq <= 0) return;
    DEV(); CHECK(vox_hip_k_apply_rope(e, x, freqs, seq, heads, head_dim));
}
