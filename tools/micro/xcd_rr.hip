// Which XCD does block 0 of a launch land on, as a function of what was launched before it on the same stream?
// (Round 4: does the dispatcher's round-robin pointer persist across launches - next start = previous start + previous grid mod 8 -
// or does every launch start at a fixed XCD?)  hipcc --offload-arch=gfx950 -O2 -o xcd_rr xcd_rr.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_probe(unsigned *out, int slot) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;
        if (blockIdx.x < 16) out[slot * 16 + blockIdx.x] = xcc;
    }
}
int main() {
    const int grids[] = {256, 256, 25, 25, 192, 160, 40, 1, 1, 256, 7, 256, 1024, 3, 256, 400, 25, 416, 800, 160, 256};
    const int n = sizeof(grids) / sizeof(int);
    unsigned *d; hipMalloc(&d, n * 16 * 4); hipMemset(d, 0xff, n * 16 * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_probe, dim3(grids[i]), dim3(rep ? 512 : 256), 0, s, d, i);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(n * 16); hipMemcpy(h.data(), d, n * 16 * 4, hipMemcpyDeviceToHost);
        int pred = -1;
        for (int i = 0; i < n; i++) {
            printf("rep %d launch %2d grid %4d: block0 on XCD %u, blocks 0..7:", rep, i, grids[i], h[i * 16]);
            for (int b = 0; b < 8 && b < grids[i]; b++) printf(" %u", h[i * 16 + b]);
            if (pred >= 0) printf("   (pointer model predicts %d)", pred);
            printf("\n");
            pred = (h[i * 16] + grids[i]) % 8;
        }
    }
    // 2-D grid: is the linear order x-fastest?
    hipMemset(d, 0xff, n * 16 * 4);
    hipLaunchKernelGGL(k_probe, dim3(5, 3), dim3(256), 0, s, d, 0);
    hipStreamSynchronize(s);
    unsigned h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("grid (5,3): blockIdx.x 0..4 of row y=0 on XCDs %u %u %u %u %u\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
