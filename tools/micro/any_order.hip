// Does hipExtAnyOrderLaunch let a kernel start while its predecessor in the same stream is still running (gfx950)?
// K1: 128 workgroups spin for ~50 us; K2 (one workgroup) stamps its start.  Launched back to back: plain, then with the flag.
// build: hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long *t, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
__global__ void k_stamp(unsigned long long *t) { if (threadIdx.x == 0) t[2] = wall_clock64(); }
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 64); hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(d, 0, 64, s);
            hipLaunchKernelGGL(k_spin, dim3(128), dim3(64), 0, s, d, 5000ull);       // 50 us at 100 MHz
            if (mode == 0) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, d);
            else hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("%s: spin start 0, spin end %.2f us, second kernel start %.2f us  (err %s)\n", mode ? "any-order" : "plain    ",
                   (double)(h[1] - h[0]) / 100.0, (double)((long long)(h[2] - h[0])) / 100.0, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
