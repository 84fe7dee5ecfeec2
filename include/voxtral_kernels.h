/*
 * voxtral_kernels.h — the reference's kernel-level API (reference voxtral_kernels.h:18-159), same names,
 * same argument meaning, host f32 buffers in row-major order in and out.
 *
 * The reference implements these on the CPU (voxtral_kernels.c); here every one of them runs as a HIP
 * kernel on the GPU through an engine-less default device context (the first call creates it on
 * VOX_DEVICE, default 0), so a client that calls any of them links against libvoxtral.so unchanged.
 * They are a compatibility / test surface: each call uploads its operands, runs, and downloads the
 * result (the streaming hot path never goes through here - it keeps everything in HBM, vox_hip.h).
 * Like the reference they return void; a device failure leaves the output untouched and prints on
 * stderr (vox_hip_last_error()).  tests/test_gpu_kernels_api.py checks each against oracle/_ref.
 */
#ifndef VOXTRAL_KERNELS_H
#define VOXTRAL_KERNELS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- basic operations (reference voxtral_kernels.h:18-22, voxtral_kernels.c:29-47) ---- */
void vox_add_inplace(float *a, const float *b, int n);              /* a += b            */
void vox_mul_inplace(float *a, const float *b, int n);              /* a *= b            */
void vox_axpy(float *a, float scale, const float *b, int n);        /* a += scale * b    */
void vox_scale(float *x, float s, int n);                           /* x *= s            */
void vox_copy(float *dst, const float *src, int n);                 /* host memcpy, as the reference */

/* ---- matrix operations (reference :28-67, voxtral_kernels.c:53-253) ---- */
void vox_matmul(float *C, const float *A, const float *B, int M, int K, int N);      /* A[M,K] @ B[K,N]   */
void vox_matmul_t(float *C, const float *A, const float *B, int M, int K, int N);    /* A[M,K] @ B[N,K]^T */
void vox_linear(float *y, const float *x, const float *W, const float *b, int seq_len, int in_dim, int out_dim);
void vox_linear_nobias(float *y, const float *x, const float *W, int seq_len, int in_dim, int out_dim);
void vox_linear_nobias_bf16(float *y, const float *x, const uint16_t *W_bf16, int seq_len, int in_dim, int out_dim);
void vox_linear_bf16(float *y, const float *x, const uint16_t *W_bf16, const float *b, int seq_len, int in_dim, int out_dim);
void vox_matmul_t_bf16(float *C, const float *A, const uint16_t *B_bf16, int M, int K, int N);

/* ---- 1-D convolution on channel-major data (reference :80-92, voxtral_kernels.c:255-340) ----
 * in [channels_in, length], weight [channels_out, channels_in, kernel_size], out [channels_out, out_length];
 * vox_conv1d: out_length = (length + 2*padding - kernel_size)/stride + 1;
 * vox_causal_conv1d: left pad = kernel_size - stride, out_length = ceil((length - kernel_size + left)/stride + 1). */
void vox_conv1d(float *out, const float *in, const float *weight, const float *bias, int channels_in, int channels_out,
                int length, int kernel_size, int stride, int padding);
void vox_causal_conv1d(float *out, const float *in, const float *weight, const float *bias, int channels_in,
                       int channels_out, int length, int kernel_size, int stride);

/* ---- normalisation / activations (reference :98-115, voxtral_kernels.c:346-406) ---- */
void vox_rms_norm(float *out, const float *x, const float *weight, int seq_len, int hidden, float eps);
void vox_silu(float *x, int n);
void vox_gelu(float *x, int n);                                     /* tanh approximation */
void vox_softmax(float *x, int rows, int cols);

/* ---- attention (reference :136-139, voxtral_kernels.c:412-482): GQA, window_size <= 0 = full causal,
 * query i sits at global position q_offset + i and sees keys max(0, p-window+1) .. min(p, seq_k-1) ---- */
void vox_causal_attention(float *out, const float *Q, const float *K, const float *V, int seq_q, int seq_k, int n_heads,
                          int n_kv_heads, int head_dim, float scale, int window_size, int q_offset);

/* ---- rotary embeddings (reference :150-157, voxtral_kernels.c:488-526) ---- */
void vox_compute_rope_freqs(float *freqs, const int *pos, int seq, int dim, float theta);   /* [seq, dim/2, (cos,sin)] */
void vox_apply_rope(float *x, const float *freqs, int seq, int heads, int head_dim);        /* pairs (2d, 2d+1), in place */

extern int vox_verbose;   /* 0 silent, 1 stats, 2 debug (reference voxtral.c:24) */
extern int vox_monitor;   /* --monitor glyph stream on stderr (reference voxtral.c:25) */
#ifdef __cplusplus
}
#endif
#endif
