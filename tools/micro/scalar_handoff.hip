// Can a hand-off be polled through the SCALAR memory path (s_load glc) instead of the vector path, so that the poll does not queue behind
// the weight loads a CU has in flight?  (Round 5.  Every hand-off of the decode step costs what is queued in front of it in the CU's vector
// memory FIFO - that FIFO is why bulk requests cannot be placed in front of a hand-off, DESIGN.md 8.6.  The scalar cache is a different
// path to L2.)  Two questions: (1) does s_load glc on one XCD SEE a write-through (sc1) vector store made on another XCD, and how fast;
// (2) what does a poll cost while the polling wave's CU streams weights.
//   hipcc --offload-arch=gfx950 -O2 -o scalar_handoff scalar_handoff.hip
// Workgroup 0 publishes a {epoch, value} granule T us after the start; workgroups 1 .. 255 (all XCDs) poll it - vector (sc1 load) or
// scalar (s_load_dwordx2 glc) - optionally issuing 64 KB of non-temporal weight loads per workgroup before EVERY poll.  Reported: time from
// the publish to each consumer's first sight of it (100 MHz wall clock), min / p50 / max over the consumers, and the bytes streamed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;

__device__ __forceinline__ u64 poll_vector(const u64 *g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 poll_scalar(const u64 *g) {
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(g) : "memory");
    return v;
}

// mode bit 0: scalar poll; bit 1: stream weights between polls; bit 2: scalar STORE for the publish (s_store_dwordx2 glc + s_dcache_wb)
__global__ __launch_bounds__(256) void k_handoff(u64 *g, u64 *out, const uint4 *bulk, size_t bulk_n, int mode, unsigned epoch, u64 delay_ticks, unsigned *sink) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const u64 t0 = wall_clock64();
    if (b == 0) {
        if (tid == 0) {
            while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
            const u64 val = ((u64)epoch << 32) | 0x1234u;
            const u64 tp = wall_clock64();
            if (mode & 4) {
                asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" :: "s"(val), "s"(g) : "memory");
            } else {
                __hip_atomic_store(g, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            out[0] = tp;
        }
        return;
    }
    unsigned acc = 0;
    u64 seen = 0, polls = 0, bytes = 0;
    const uint4 *src = bulk + ((size_t)b * 65536 + tid) % (bulk_n - 16 * 256 * 64);
    for (int it = 0; it < 4000; it++) {
        if (mode & 2) {       // 16 x 16 B per thread = 64 KB per workgroup in front of every poll
            uint4 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + (size_t)i * 256));
                w[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
            src += 16 * 256; if ((size_t)(src - bulk) > bulk_n - 16 * 256 * 64) src = bulk + tid;
            __builtin_amdgcn_sched_barrier(0);
            // the poll is issued BEHIND the loads; their data is consumed after it
            u64 v = (mode & 1) ? poll_scalar(g) : poll_vector(g);
            polls++;
            if ((unsigned)(v >> 32) == epoch && !seen) seen = wall_clock64();
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
            bytes += 16 * 16;
        } else {
            u64 v = (mode & 1) ? poll_scalar(g) : poll_vector(g);
            polls++;
            if ((unsigned)(v >> 32) == epoch && !seen) seen = wall_clock64();
            __builtin_amdgcn_s_sleep(2);
        }
        if (seen) break;
    }
    if (tid == 0) { out[b] = seen; out[256 + b] = polls; out[512 + b] = wall_clock64() - t0; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    u64 *g, *out; uint4 *bulk; unsigned *sink;
    const size_t bulk_n = (size_t)1 << 27;            // 2 GiB of uint4
    if (hipMalloc(&g, 64) != hipSuccess || hipMalloc(&out, 768 * 8) != hipSuccess || hipMalloc(&bulk, bulk_n * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(g, 0, 64); (void)hipMemset(bulk, 1, bulk_n * 16);
    const char *names[8] = {"vector store, vector poll, idle", "vector store, SCALAR poll, idle", "vector store, vector poll, streaming 64 KB per poll",
                            "vector store, SCALAR poll, streaming 64 KB per poll", "SCALAR store, vector poll, idle", "SCALAR store, SCALAR poll, idle",
                            "SCALAR store, vector poll, streaming", "SCALAR store, SCALAR poll, streaming"};
    unsigned epoch = 1;
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            (void)hipMemset(out, 0, 768 * 8);
            hipLaunchKernelGGL(k_handoff, dim3(256), dim3(256), 0, 0, g, out, (const uint4 *)bulk, bulk_n, mode, ++epoch, (u64)3000, sink);   // publish 30 us in
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed\n", mode); return 1; }
            std::vector<u64> h(768); (void)hipMemcpy(h.data(), out, 768 * 8, hipMemcpyDeviceToHost);
            std::vector<double> lat; int missed = 0; double polls = 0, total = 0;
            for (int b = 1; b < 256; b++) {
                if (!h[b]) { missed++; continue; }
                lat.push_back(((double)h[b] - (double)h[0]) / 100.0); polls += (double)h[256 + b]; total += (double)h[512 + b] / 100.0;
            }
            std::sort(lat.begin(), lat.end());
            if (lat.empty()) { printf("mode %d (%s) rep %d: NOBODY saw the granule (%d consumers gave up)\n", mode, names[mode], rep, missed); continue; }
            printf("mode %d (%s) rep %d: seen by %zu / 255, publish -> first sight us: min %.2f p50 %.2f p90 %.2f max %.2f; polls per consumer %.0f, us per poll %.2f\n",
                   mode, names[mode], rep, lat.size(), lat.front(), lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back(), polls / lat.size(),
                   total / std::max(1.0, polls));
        }
    }
    return 0;
}
