/*
 * voxtral_kernels.h — kernel-level surface kept for source compatibility with clients
 * that include the reference header (main.c:9 does, for the two globals below).
 *
 * The reference implements these on the CPU (voxtral_kernels.c).  Here there is no CPU
 * math: the kernel-level entry points live in include/vox_hip.h (vox_hip_linear_bf16,
 * vox_hip_causal_attention, ...) and need an engine handle, so they are not redeclared
 * with the engine-less reference signatures.
 */
#ifndef VOXTRAL_KERNELS_H
#define VOXTRAL_KERNELS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
extern int vox_verbose;   /* 0 silent, 1 stats, 2 debug (reference voxtral.c:24) */
extern int vox_monitor;   /* --monitor glyph stream on stderr (reference voxtral.c:25) */
#ifdef __cplusplus
}
#endif
#endif
