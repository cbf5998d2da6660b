// Probe: does v_mfma_f32_32x32x2_f32 overlap with plain f32 VALU work (a) in the same wave,
// (b) in the partner wave on the same SIMD?   hipcc --offload-arch=gfx950 -O3 -o probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NVALU, int MODE>   // MODE 0: every wave MFMA + NVALU valu per MFMA; MODE 1: waves 0-3 MFMA only, waves 4-7 VALU only
__global__ __launch_bounds__(512, 2) void probe(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool do_mfma = MODE == 0 || wave < 4;
    const bool do_valu = MODE == 0 || wave >= 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (do_mfma) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int k = 0; k < NVALU; ++k) v[(i + k) & 7] = __builtin_fmaf(v[(i + k) & 7], b, a);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { s += v[i]; for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NVALU, int MODE>
float run(float *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NVALU, MODE><<<256, 512>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NVALU, MODE><<<256, 512>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;   // 8 MFMAs per iteration per wave
    const double mf = 256.0 * 8 * iters * 8;    // MFMAs (all 8 waves)
    printf("MODE 0 (every wave: MFMA + N VALU per MFMA), 2 waves/SIMD\n");
    float t0 = run<0, 0>(out, iters); printf(" N=0  %.3f ms  %.1f TF/s\n", t0, mf * 4096 / t0 / 1e9);
    float t1 = run<2, 0>(out, iters); printf(" N=2  %.3f ms (+%.1f%%)\n", t1, (t1 / t0 - 1) * 100);
    float t2 = run<4, 0>(out, iters); printf(" N=4  %.3f ms (+%.1f%%)\n", t2, (t2 / t0 - 1) * 100);
    float t3 = run<8, 0>(out, iters); printf(" N=8  %.3f ms (+%.1f%%)\n", t3, (t3 / t0 - 1) * 100);
    float t4 = run<16, 0>(out, iters); printf(" N=16 %.3f ms (+%.1f%%)\n", t4, (t4 / t0 - 1) * 100);
    printf("MODE 1 (waves 0-3 MFMA only, waves 4-7 N VALU per slot only)\n");
    float u0 = run<0, 1>(out, iters); printf(" N=0  %.3f ms  (half the MFMAs of MODE 0)\n", u0);
    float u1 = run<4, 1>(out, iters); printf(" N=4  %.3f ms (+%.1f%%)\n", u1, (u1 / u0 - 1) * 100);
    float u2 = run<16, 1>(out, iters); printf(" N=16 %.3f ms (+%.1f%%)\n", u2, (u2 / u0 - 1) * 100);
    float u3 = run<32, 1>(out, iters); printf(" N=32 %.3f ms (+%.1f%%)\n", u3, (u3 / u0 - 1) * 100);
    return 0;
}
