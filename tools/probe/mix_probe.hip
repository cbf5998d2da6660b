// Probe: cost, in MFMA-pipe time, of (a) packed-fp32 VALU (v_pk_fma_f32), (b) ds_read_b64, (c) global dword loads issued
// between v_mfma_f32_32x32x2_f32, with W waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mix_probe mix_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NPK, int NFMA, int NLDS, int NGL, int THREADS, int NACC>
__global__ __launch_bounds__(THREADS) void probe(float *out, const float *__restrict__ gin, int iters) {
    __shared__ f32x2 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = f32x2{1e-9f * i, 0.f};
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f32x2 v[8]; float s1[8];
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{a + i, a - i}; s1[i] = a * i; }
    const f32x2 bb = {b, b}, aa = {a, a};
    const float *gp = gin + threadIdx.x;
    int lo = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i & 7].x, s1[i & 7], acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NPK; ++k) v[(i + k) & 7] = __builtin_elementwise_fma(v[(i + k) & 7], bb, aa);
#pragma unroll
            for (int k = 0; k < NFMA; ++k) s1[(i + k) & 7] = __builtin_fmaf(s1[(i + k) & 7], b, a);
#pragma unroll
            for (int k = 0; k < NLDS; ++k) { f32x2 t = lds[(lo + 64 * ((i * NLDS + k + it) & 31))]; v[(i + k + 3) & 7] += t; }
#pragma unroll
            for (int k = 0; k < NGL; ++k) { s1[(i + k + 5) & 7] += gp[((i * NGL + k + it) & 63) * THREADS]; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + s1[i];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NPK, int NFMA, int NLDS, int NGL, int THREADS, int NACC>
void run(float *out, const float *gin, int nblk, const char *tag) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NPK, NFMA, NLDS, NGL, THREADS, NACC><<<nblk, THREADS>>>(out, gin, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NPK, NFMA, NLDS, NGL, THREADS, NACC><<<nblk, THREADS>>>(out, gin, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)nblk * (THREADS / 64) * iters * NACC;
    printf("%-44s pk=%d fma=%d lds=%d gl=%d  %8.3f ms  %6.1f TF/s\n", tag, NPK, NFMA, NLDS, NGL, ms, mf * 4096 / ms / 1e9);
}

int main() {
    float *out, *gin; hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&gin, 64 * 1024 * 4 + 4096); hipMemset(gin, 0, 64 * 1024 * 4 + 4096);
    // 8 waves/CU (2 per SIMD), 8 accumulators: today's Winograd kernel shape
    run<0, 0, 0, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 baseline");
    run<0, 2, 0, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 2 fma");
    run<2, 0, 0, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 2 pk");
    run<0, 4, 0, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 4 fma");
    run<4, 0, 0, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 4 pk");
    run<0, 0, 1, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 1 lds");
    run<0, 0, 2, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 2 lds");
    run<0, 0, 0, 1, 512, 8>(out, gin, 256, "2w/SIMD acc8 1 gl");
    run<0, 2, 1, 0, 512, 8>(out, gin, 256, "2w/SIMD acc8 2fma+1lds (~today)");
    // 12 waves/CU (3 per SIMD), 6 accumulators: F(4x4,3x3) candidate
    run<0, 0, 0, 0, 768, 6>(out, gin, 256, "3w/SIMD acc6 baseline");
    run<2, 0, 2, 0, 768, 6>(out, gin, 256, "3w/SIMD acc6 2pk+2lds");
    run<3, 0, 2, 0, 768, 6>(out, gin, 256, "3w/SIMD acc6 3pk+2lds");
    run<2, 0, 2, 1, 768, 6>(out, gin, 256, "3w/SIMD acc6 2pk+2lds+1gl");
    run<0, 4, 2, 1, 768, 6>(out, gin, 256, "3w/SIMD acc6 4fma+2lds+1gl");
    run<0, 5, 3, 1, 768, 6>(out, gin, 256, "3w/SIMD acc6 5fma+3lds+1gl");
    // 2 workgroups of 6 waves per CU
    run<2, 0, 2, 1, 384, 6>(out, gin, 512, "2x6w/CU acc6 2pk+2lds+1gl");
    return 0;
}
