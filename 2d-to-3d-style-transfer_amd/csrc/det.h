// det.h -- order-independent (bitwise reproducible) scatter-add for the two gradient scatters of the render backward.
// Float atomics sum in whatever order the hardware serves them, so the texture / vertex gradient differs in its last
// bits from run to run.  The deterministic variants accumulate in 64-bit FIXED POINT instead: integer addition is
// associative, so any order -- lanes, tiles, views -- gives the same bits.  The scale is a power of two chosen from a
// bound on the magnitude of any partial sum (the sum of the absolute contributions, reduced in a fixed order by a first
// pass), so nothing can overflow and the quantisation step is 2^-60 of that bound -- far below fp32 resolution of the sums.
// A NaN or an infinity in the scattered values makes the bound non-finite: the header is then poisoned (inv = NaN) and the
// conversion pass writes NaN into every element -- a diverged run fails as loudly as it does with float atomics or in the
// reference, instead of the integer conversion laundering the NaN into a finite number.
//
// workspace layout: [DetHeader][partials float x det_partials][int64 accumulators x n]
#pragma once
#include "common.h"

namespace st3d_det {

struct DetHeader { double scale, inv; };

constexpr int kHeaderBytes = 16;

__host__ __device__ inline size_t workspace_bytes(size_t n_accum, size_t n_partials) {
    return kHeaderBytes + ((n_partials * sizeof(float) + 15) & ~(size_t)15) + n_accum * sizeof(long long);
}
inline float *partials_of(void *ws) { return reinterpret_cast<float *>(static_cast<char *>(ws) + kHeaderBytes); }
inline long long *accum_of(void *ws, size_t n_partials) {
    return reinterpret_cast<long long *>(static_cast<char *>(ws) + kHeaderBytes + ((n_partials * sizeof(float) + 15) & ~(size_t)15));
}

// bound = sum of the per-block partial sums (fixed tree, fp64) -> scale = 2^(60 - ceil(log2 bound))
static __global__ __launch_bounds__(256) void det_scale_kernel(const float *__restrict__ partials, int np, DetHeader *hdr) {
    __shared__ double s[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) v += (double)partials[i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double bound = s[0];
        int k = 0;
        if (bound > 0.0 && bound < 1e300) {
            int e;
            frexp(bound, &e);               // bound = m * 2^e, 0.5 <= m < 1  =>  bound < 2^e
            k = 60 - e;
        }
        if (k > 1000) k = 1000;
        if (k < -1000) k = -1000;
        hdr->scale = ldexp(1.0, k);
        hdr->inv = ldexp(1.0, -k);
        if (!(bound >= 0.0 && bound < 1e300)) {     // NaN / Inf upstream: poison the result (bound is a sum of |.|, never < 0)
            hdr->scale = 0.0;
            hdr->inv = __longlong_as_double(0x7ff8000000000000ll);
        }
    }
}

__device__ __forceinline__ long long det_quantise(float v, double scale) { return __double2ll_rn((double)v * scale); }

static __global__ __launch_bounds__(256) void det_convert_kernel(const long long *__restrict__ acc, size_t n, const DetHeader *hdr,
                                                                 int accumulate, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = (float)((double)acc[i] * hdr->inv);
    out[i] = accumulate ? out[i] + v : v;
}

__device__ __forceinline__ float det_block_sum(float v, float *s4) {      // 256 threads, fixed tree
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// partials[block] = sum |x| over the block's grid-stride share (the bound of the texture scatters: every contribution is
// a gradient value times blend and bilinear weights <= 1)
__attribute__((unused)) static __global__ __launch_bounds__(256) void det_abs_sum_kernel(const float *__restrict__ x, size_t n, float *__restrict__ partials) {
    __shared__ float s4[4];
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += fabsf(x[i]);
    const float t = det_block_sum(acc, s4);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

}  // namespace st3d_det
