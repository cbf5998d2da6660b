// tap0.hip -- the bottom of the VGG backward in ONE pass over the two largest tensors of the step: the input gradient of
// conv1_1 (64 -> 3 channels at full resolution, second_approach.py:188 through utils.py:49 features[0..1]) fused with
// the style gradient of the relu1_1 tap (autograd of losses.py:36-39 through style_transfer.py:31-35).
//
// Unfused, the step streams the 64-channel gradient three times and the 64-channel activation twice:
//   gram_bwd  : g += coef * D F          reads F, reads + writes g       (3 x 64 x HW x 4 B per view)
//   dgrad     : gx = conv^T(g * [F > 0]) reads g, reads F                (2 x 64 x HW x 4 B)
// Here every pixel's 64-vector is read once (g and F, 2 x 64 x HW x 4 B) and both products run on the matrix pipe:
//   GEMM 1    t = g + coef * D F                  M = 64 channels, K = 64, N = pixels; the accumulators START from g
//   gate      t = F > 0 ? t : 0                   F is already in registers as GEMM 1's B operand
//   GEMM 2    Y[(tap, i)] = W'[(tap, i)][c] t[c]  M = 27 (9 taps x 3 image channels, padded to 32), K = 64, N = pixels;
//                                                 GEMM 1's accumulator registers ARE its B operands (the K order of
//                                                 both products is permuted to the MFMA accumulator row order, which a
//                                                 reduction does not notice), so nothing moves between the two
//   gather    gx[i][y][x] = sum over the 9 taps of Y[(tap, i)][y + ky - 1][x + kx - 1]   (second, tiny kernel)
// Y (27 planes) costs 27/64 of one gradient write + read; the pass is 1-D over pixels (no halo, no spatial tiling), and
// every sum has a fixed order (bitwise reproducible).  fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout, like gram.hip.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kOob = 0x80000000u;
constexpr int kTaps = 27;                 // (ky, kx, image channel)

// row of accumulator register v in lane half h of a 32x32 MFMA tile
__host__ __device__ constexpr int acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

// One wave = 32 * J consecutive pixels per iteration as J MFMA n-tiles (pixel p0 + J * lane31 + j), so every global access is
// a 4 * J-byte item per lane and a 128 * J-byte run per row per wave; the four waves of a workgroup take adjacent runs.
// J = 2: 8-byte accesses at 2 waves per SIMD (256 VGPRs); J = 1: 4-byte accesses at 3 waves per SIMD.
template <bool HAS_D, bool HAS_G, int J>
__global__ __launch_bounds__(256, J == 2 ? 3 : 4) void conv1_bwd_gemm_kernel(const float *__restrict__ g, const float *__restrict__ F,
                                                                         const float *__restrict__ D, float coef,
                                                                         const float *__restrict__ wd, float *__restrict__ Y,
                                                                         int HW, int iters) {
    __shared__ float A1s[HAS_D ? 2 * 32 * 64 : 64];   // [cout half][k step][lane]: coef * D[cout][k]
    __shared__ float A2s[32 * 64];                    // [k step][lane]: W'[(tap, i) = lane31][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.y;
    if (HAS_D) {
        const float *Dn = D + (size_t)n * 4096;
        for (int e = tid; e < 4096; e += 256) {
            const int ln = e & 63, s = (e >> 6) & 31, mb = e >> 11;
            A1s[e] = coef * Dn[((ln & 31) + 32 * mb) * 64 + 32 * (s >> 4) + acc_row(s & 15, ln >> 5)];
        }
    }
    for (int e = tid; e < 2048; e += 256) {
        const int ln = e & 63, t = e >> 6, row = ln & 31;
        const int c = 32 * (t >> 4) + acc_row(t & 15, ln >> 5);
        const int tap = row / 3, i = row - 3 * tap;
        A2s[e] = row < kTaps ? wd[((size_t)tap * 64 + c) * 128 + i] : 0.f;      // dgrad pack [tap'][64][128] of conv.hip
    }
    __syncthreads();

    const unsigned rowb = (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F + (size_t)n * 64 * HW), 0,
                                                                       64u * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(HAS_G ? g + (size_t)n * 64 * HW : F), 0, 64u * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)n * kTaps * HW, 0, (unsigned)kTaps * rowb,
                                                                       0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t &r, unsigned vo, unsigned so, float (&dst)[J]) __attribute__((always_inline)) {
        if (J == 2) {
            const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
            dst[0] = t[0]; dst[J - 1] = t[1];
        } else {
            dst[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
        }
    };
    for (int it = 0; it < iters; ++it) {
        const int px = ((blockIdx.x * iters + it) * 4 + wave) * (32 * J) + J * l31;     // J = 2: HW is even, px and px + 1 fall together
        const unsigned voff = px < HW ? (unsigned)px * 4u + (unsigned)lhi * 4u * rowb : kOob;
        float Fv[32][J];
        f32x16 acc1[J][2];          // [pixel j][channel half]
#pragma unroll
        for (int s = 0; s < 32; ++s) ld(rF, voff, (unsigned)(32 * (s >> 4) + acc_row(s & 15, 0)) * rowb, Fv[s]);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            float t[J];
#pragma unroll
            for (int j = 0; j < J; ++j) t[j] = 0.f;
            if (HAS_G) ld(rG, voff, (unsigned)(32 * (s >> 4) + acc_row(s & 15, 0)) * rowb, t);
#pragma unroll
            for (int j = 0; j < J; ++j) acc1[j][s >> 4][s & 15] = t[j];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the operand reads of the products below from being hoisted over the loads
        if (HAS_D) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float a0 = A1s[s * 64 + lane], a1 = A1s[2048 + s * 64 + lane];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    acc1[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, Fv[s][j], acc1[j][0], 0, 0, 0);
                    acc1[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, Fv[s][j], acc1[j][1], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc2[J];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const float a2 = A2s[t * 64 + lane];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float b = Fv[t][j] > 0.f ? acc1[j][t >> 4][t & 15] : 0.f;     // ReLU gate of relu1_1
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc2[j], 0, 0, 0);
            }
        }
        // rows acc_row(v, lhi) < 27 of Y: v = 15 holds rows 27 / 31, the upper lane half of v = 12..14 rows 28..30
#pragma unroll
        for (int v = 0; v < 15; ++v) {
            const unsigned vo = (v >= 12 && lhi) ? kOob : voff;
            if (J == 2) {
                const f32x2 o = {acc2[0][v], acc2[J - 1][v]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rY, vo, (unsigned)acc_row(v, 0) * rowb, 0);
            } else {
                const float o = acc2[0][v];      // (a bit_cast of the vector ELEMENT expression reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rY, vo, (unsigned)acc_row(v, 0) * rowb, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv1_bwd_gather_kernel(const float *__restrict__ Y, float *__restrict__ gx, int H,
                                                               int W) {
    const size_t HW = (size_t)H * W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const float *Yn = Y + (size_t)blockIdx.y * kTaps * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float *q = Yn + (size_t)((ky * 3 + kx) * 3) * HW + (size_t)yy * W + xx;
            a0 += q[0]; a1 += q[HW]; a2 += q[2 * HW];
        }
    float *o = gx + (size_t)blockIdx.y * 3 * HW + p;
    o[0] = a0; o[HW] = a1; o[2 * HW] = a2;
}

}  // namespace

extern "C" int st3d_conv1_bwd_supported(int H, int W) {
    const long hw = (long)H * W;
    return H > 0 && W > 0 && (hw % 2) == 0 && hw * 64 * 4 < (1L << 31);
}

extern "C" size_t st3d_conv1_bwd_workspace_bytes(int N, int H, int W) {
    return (size_t)(N > 0 ? N : 0) * kTaps * H * W * sizeof(float);
}

extern "C" int st3d_conv1_bwd(const float *gy, const float *act, const float *D, float coef, const float *w_dgrad_packed,
                              void *workspace, size_t workspace_bytes, float *gx, int N, int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(act && w_dgrad_packed && workspace && gx);
    ST3D_CHECK_ARG(gy || D);
    ST3D_CHECK_ARG(N > 0 && st3d_conv1_bwd_supported(H, W));
    ST3D_CHECK_ARG(workspace_bytes >= st3d_conv1_bwd_workspace_bytes(N, H, W));
    ST3D_CHECK_ARG((((uintptr_t)gy | (uintptr_t)act | (uintptr_t)workspace) & 7) == 0);
    hipStream_t s = st3d::as_stream(stream);
    const int HW = H * W;
    float *Y = reinterpret_cast<float *>(workspace);
    // ST3D_TAP0_J=1 / 2: pixels per lane (A/B runs)
    static const int J = [] { const char *e = getenv("ST3D_TAP0_J"); return e && e[0] == '1' ? 1 : 2; }();
    const int iters = HW >= 4096 ? 4 : 1;               // 128 * J * iters pixels per workgroup
    const dim3 grid(st3d::cdiv(HW, 128 * J * iters), N);
#define ST3D_TAP0_LAUNCH(JJ)                                                                                                  \
    do {                                                                                                                      \
        if (D && gy) conv1_bwd_gemm_kernel<true, true, JJ><<<grid, 256, 0, s>>>(gy, act, D, coef, w_dgrad_packed, Y, HW, iters);   \
        else if (D) conv1_bwd_gemm_kernel<true, false, JJ><<<grid, 256, 0, s>>>(gy, act, D, coef, w_dgrad_packed, Y, HW, iters);   \
        else conv1_bwd_gemm_kernel<false, true, JJ><<<grid, 256, 0, s>>>(gy, act, D, coef, w_dgrad_packed, Y, HW, iters);          \
    } while (0)
    if (J == 1) ST3D_TAP0_LAUNCH(1);
    else ST3D_TAP0_LAUNCH(2);
#undef ST3D_TAP0_LAUNCH
    ST3D_LAUNCH_CHECK();
    conv1_bwd_gather_kernel<<<dim3(st3d::cdiv(HW, 256), N), 256, 0, s>>>(Y, gx, H, W);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
