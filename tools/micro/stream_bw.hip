// Micro-benchmark: what read bandwidth does a GEMV-shaped weight stream reach on MI355X, by access pattern?
// Each launch reads BYTES of a large rotating buffer (no cache reuse) with 16-byte non-temporal loads and xors them
// into a sink.  Patterns: how a workgroup's share is cut into per-wave streams, how many loads are in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// row pattern: wave w of block b streams rows (1 row = ROWB bytes contiguous), pieces of 1 KB per instruction
template <int THREADS, int NLOAD, int ROUNDS, bool CONTIG>
__global__ __launch_bounds__(THREADS) void k_stream(const unsigned char *base, size_t bytes_per_block, int rowb, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = THREADS / 64;
    const unsigned char *blk = base + (size_t)blockIdx.x * bytes_per_block;
    u32x4 acc = {0, 0, 0, 0};
    // a wave's share: bytes_per_block / WAVES = NLOAD * ROUNDS KB
    u32x4 r[NLOAD];
#pragma unroll 1
    for (int rd = 0; rd < ROUNDS; rd++) {
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int piece = rd * NLOAD + i;             // this wave's piece index, 1 KB each
            size_t off;
            if (CONTIG) off = ((size_t)piece * WAVES + wave) * 1024;               // waves interleaved at 1 KB: the block reads one contiguous run
            else off = (size_t)wave * (NLOAD * ROUNDS * 1024) + (size_t)piece * 1024;  // each wave its own contiguous run (row-like)
            r[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(blk + off) + lane);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; i++) acc ^= r[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}
template <int THREADS, int NLOAD, int ROUNDS, bool CONTIG>
static void run(const char *name, const unsigned char *buf, size_t bufbytes, unsigned *sink, int blocks) {
    const size_t per_block = (size_t)(THREADS / 64) * NLOAD * ROUNDS * 1024;
    const size_t per_launch = per_block * blocks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    size_t ofs = 0;
    for (int w = 0; w < 3; w++) { hipLaunchKernelGGL((k_stream<THREADS, NLOAD, ROUNDS, CONTIG>), dim3(blocks), dim3(THREADS), 0, 0, buf + ofs, per_block, 0, sink); ofs = (ofs + per_launch) % (bufbytes - per_launch); ofs &= ~(size_t)4095; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) { hipLaunchKernelGGL((k_stream<THREADS, NLOAD, ROUNDS, CONTIG>), dim3(blocks), dim3(THREADS), 0, 0, buf + ofs, per_block, 0, sink); ofs = (ofs + per_launch) % (bufbytes - per_launch); ofs &= ~(size_t)4095; }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-58s %7.1f MB/launch  %6.2f us/launch (incl. boundary)  %5.2f TB/s\n", name, per_launch / 1e6, us, per_launch / us / 1e6);
}
int main() {
    const size_t bufbytes = (size_t)6 << 30;
    unsigned char *buf; unsigned *sink;
    if (hipMalloc(&buf, bufbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 1 << 20); hipMemset(buf, 1, bufbytes);
    run<768, 12, 3, false>("256x768 thr, 3 rounds x 12 loads, wave-contiguous (w13x)", buf, bufbytes, sink, 256);
    run<768, 12, 3, true >("256x768 thr, 3 rounds x 12 loads, block-contiguous", buf, bufbytes, sink, 256);
    run<768, 36, 1, false>("256x768 thr, 36 loads up front, wave-contiguous", buf, bufbytes, sink, 256);
    run<768, 6, 6, false>("256x768 thr, 6 rounds x 6 loads, wave-contiguous", buf, bufbytes, sink, 256);
    run<1024, 9, 3, false>("256x1024 thr, 3 rounds x 9 loads, wave-contiguous", buf, bufbytes, sink, 256);
    run<512, 18, 3, false>("256x512 thr, 3 rounds x 18 loads, wave-contiguous", buf, bufbytes, sink, 256);
    run<256, 18, 2, false>("768x256 thr, 2 rounds x 18 loads (3 blocks per CU)", buf, bufbytes, sink, 768);
    run<256, 12, 3, true >("768x256 thr, 3 rounds x 12, block-contiguous", buf, bufbytes, sink, 768);
    run<768, 12, 12, false>("256x768 thr, 12 rounds x 12 loads (453 MB: steady state)", buf, bufbytes, sink, 256);
    run<768, 12, 12, true>("256x768 thr, 12 rounds x 12 loads, block-contiguous (453 MB)", buf, bufbytes, sink, 256);
    run<1024, 8, 16, true>("256x1024 thr, 16 rounds x 8 loads, block-contiguous (537 MB)", buf, bufbytes, sink, 256);
    run<256, 16, 8, true>("1024x256 thr, 8 rounds x 16, block-contiguous (537 MB)", buf, bufbytes, sink, 1024);
    return 0;
}
