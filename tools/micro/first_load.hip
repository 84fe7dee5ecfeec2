// Micro-benchmark: what does the first global load of a kernel cost right after a kernel boundary on MI355X?
// (tuning aid for DESIGN.md section "kernel boundary"; build: hipcc --offload-arch=gfx950 -O2 first_load.hip -o first_load)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long wclk() { return __builtin_readcyclecounter() * 0 + wall_clock64(); }
__global__ void k_fill(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}
// each block: lane 0 of wave 0 times 3 dependent loads: (a) a line written by the previous kernel, (b) a line of a big read-only
// buffer never touched, (c) the same line as (a) again.  out[b] = {entry, t_a, t_b, t_c} in 10 ns ticks.
__global__ void k_probe(const float *fresh, const float *weights, unsigned long long *out, float *sink) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    float a = __builtin_nontemporal_load(fresh + (size_t)blockIdx.x * 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    float b = __builtin_nontemporal_load(weights + (size_t)blockIdx.x * 1536 * 36 + (size_t)(a != 12345.f ? 0 : 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    float c = __builtin_nontemporal_load(fresh + (size_t)blockIdx.x * 32 + 1 + (size_t)(b != 12345.f ? 0 : 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    out[4 * blockIdx.x] = t0; out[4 * blockIdx.x + 1] = t1 - t0; out[4 * blockIdx.x + 2] = t2 - t1; out[4 * blockIdx.x + 3] = t3 - t2;
    sink[blockIdx.x] = a + b + c;
}
int main() {
    const size_t nfresh = 1 << 20, nw = (size_t)128 << 20;
    float *fresh, *weights, *sink; unsigned long long *out;
    hipMalloc(&fresh, nfresh * 4); hipMalloc(&weights, nw * 4); hipMalloc(&sink, 4096 * 4); hipMalloc(&out, 4096 * 32);
    hipMemset(weights, 0, nw * 4);
    hipStream_t s; hipStreamCreate(&s);
    auto report = [&](const char *name) {
        std::vector<unsigned long long> h(256 * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> a, b, c;
        unsigned long long e0 = ~0ull, e1 = 0;
        for (int i = 0; i < 256; i++) { a.push_back(h[4 * i + 1] / 100.0); b.push_back(h[4 * i + 2] / 100.0); c.push_back(h[4 * i + 3] / 100.0); e0 = std::min(e0, h[4 * i]); e1 = std::max(e1, h[4 * i]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        printf("%-46s first load (fresh line)  min %.2f p50 %.2f max %.2f | weights line min %.2f p50 %.2f max %.2f | fresh again min %.2f p50 %.2f max %.2f us | entry skew %.2f us\n",
               name, a[0], a[128], a[255], b[0], b[128], b[255], c[0], c[128], c[255], (e1 - e0) / 100.0);
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, fresh, nfresh);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, s, fresh, weights + (size_t)rep * 1536 * 36 * 256, out, sink);
        hipStreamSynchronize(s);
        report("probe right behind a writer kernel:");
    }
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, s, fresh, weights + (size_t)2 * 1536 * 36 * 256, out, sink);
    hipStreamSynchronize(s);
    report("probe alone (stream idle before):");
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, s, fresh, weights + (size_t)(3 + rep) * 1536 * 36 * 256, out, sink);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, s, fresh, weights + (size_t)(5 + rep) * 1536 * 36 * 256, out, sink);
        hipStreamSynchronize(s);
        report("probe behind a probe (nothing written):");
    }
    return 0;
}
