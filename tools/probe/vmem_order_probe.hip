// Probe: do a CU's vector-memory loads come back in issue order ACROSS waves?
//   hipcc --offload-arch=gfx950 -O3 -o vmem_order_probe vmem_order_probe.hip && ./vmem_order_probe
// One workgroup of two waves on one CU.  Wave 1 times single 16-byte loads from a small region that sits in L2 (warmed,
// larger than the 32 KB L1).  Wave 0 either idles (mode 0), streams 16-byte loads from a 2 GB buffer nobody touched --
// HBM misses -- (mode 1), or streams loads from the warm region (mode 2).  If returns are ordered per wave only, wave 1's
// latency is the same in all modes; if they are ordered per CU, its L2 hits wait behind wave 0's misses in mode 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void probe(const f32x4 *hot, const f32x4 *cold, size_t cold_n, int iters, int mode, long long *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    if (wave == 0) {
        if (mode != 0) {
            // keep ~8 loads in flight, back to back, for as long as wave 1 measures
            for (int it = 0; it < iters; ++it) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t q = (size_t)(it * 8 + k);
                    const f32x4 *p = mode == 1 ? cold + (q * 64 * 37 + (size_t)lane * 97) % cold_n : hot + ((q * 64 + lane) & 16383);
                    v[k] = *(const volatile f32x4 *)p;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sink += v[k][0];
            }
        }
    } else {
        long long tot = 0, mx = 0;
        for (int it = 0; it < iters; ++it) {
            const f32x4 *p = hot + ((it * 4099 + lane) & 16383);
            __builtin_amdgcn_s_waitcnt(0);
            const long long t0 = __builtin_amdgcn_s_memtime();
            const f32x4 v = *(const volatile f32x4 *)p;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t1 = __builtin_amdgcn_s_memtime();
            sink += v[0];
            tot += t1 - t0;
            mx = (t1 - t0) > mx ? (t1 - t0) : mx;
            __builtin_amdgcn_s_sleep(4);
        }
        if (lane == 0) { out[0] = tot; out[1] = mx; }
    }
    if (sink == 12345.678f) out[2] = 1;
}

int main() {
    const size_t hot_n = 16384, cold_n = (size_t)2 << 30 >> 4;        // 256 KB warm region, 2 GB cold buffer
    f32x4 *hot, *cold; long long *out, h[3];
    hipMalloc(&hot, hot_n * 16); hipMalloc(&cold, cold_n * 16); hipMalloc(&out, 24);
    hipMemset(hot, 0, hot_n * 16); hipMemset(cold, 0, cold_n * 16);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(out, 0, 24);
            probe<<<1, 128>>>(hot, cold, cold_n, 200, 2, out);            // warm the hot region into L2
            probe<<<1, 128>>>(hot, cold, cold_n, iters, mode, out);
            hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
            printf("mode %d (%s): wave 1's L2-hit load latency avg %.0f, max %lld s_memtime ticks\n", mode,
                   mode == 0 ? "wave 0 idle" : mode == 1 ? "wave 0 streams HBM misses" : "wave 0 streams L2 hits", (double)h[0] / iters, h[1]);
        }
    return 0;
}
