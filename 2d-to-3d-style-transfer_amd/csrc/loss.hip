// loss.hip -- HBM-bound elementwise / reduction kernels of the step:
//   squared-difference sums (content MSE losses.py:31, per-layer style MSE losses.py:38,
//   masked MSE losses.py:71-75), their gradients, and the fused dense Adam update
//   (torch.optim.Adam defaults: utils.py:185-195, style_transfer.py:57).
// Coalesced 4 B/lane, grid-strided over <= 1024 workgroups; reductions are two-stage and
// ordered (per-workgroup partials, then one workgroup sums them in index order) so results are
// bitwise reproducible run to run -- no float atomics.
#include "common.h"

namespace {

constexpr int NPART = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ void block_store_partial(float v, float *partials) {
    __shared__ float s[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// partials[blk] = sum (a-b)^2 ; optionally D = a-b.  b has period nb.
__global__ __launch_bounds__(256) void sqdiff_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                     size_t nb, float *__restrict__ D, float *__restrict__ partials) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - b[nb == n ? i : i % nb];
        if (D) D[i] = d;
        acc += d * d;
    }
    block_store_partial(acc, partials);
}

__global__ __launch_bounds__(256) void finish_kernel(const float *__restrict__ partials, int np, float scale,
                                                     float *__restrict__ out) {
    __shared__ double s[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) v += (double)partials[i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = out[0] + (float)(s[0] * (double)scale);
}

__global__ __launch_bounds__(256) void axpy_diff_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                        float coef, int accumulate, float *__restrict__ g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = coef * (a[i] - b[i]);
        g[i] = accumulate ? g[i] + v : v;
    }
}

// masked MSE: d = r*m - t*m ; loss partial = d^2 ; grad_r = coef * d * m
__global__ __launch_bounds__(256) void masked_mse_kernel(const float *__restrict__ r, const float *__restrict__ t,
                                                         const float *__restrict__ m, int B, size_t HW, float coef,
                                                         float *__restrict__ gr, float *__restrict__ partials) {
    float acc = 0.f;
    const size_t n = (size_t)B * 3 * HW;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t b = i / (3 * HW), p = i % HW;
        const float mk = m[b * HW + p];
        const float d = r[i] * mk - t[i] * mk;
        acc += d * d;
        if (gr) gr[i] = coef * d * mk;
    }
    block_store_partial(acc, partials);
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, size_t n,
                                                   float step_size, float bc2_sqrt, float b1, float b2, float eps) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > NPART ? NPART : b));
}

}  // namespace

extern "C" int st3d_reduce_partials(void) { return NPART; }

extern "C" int st3d_sqdiff_sum(const float *a, const float *b, size_t n, size_t nb, float scale, float *D, float *partials,
                               float *loss_out, st3d_stream_t stream) {
    ST3D_CHECK_ARG(a && b && partials && loss_out);
    ST3D_CHECK_ARG(n > 0 && nb > 0 && n % nb == 0);
    hipStream_t s = st3d::as_stream(stream);
    const int gsz = grid_for(n);
    sqdiff_kernel<<<gsz, 256, 0, s>>>(a, b, n, nb, D, partials);
    ST3D_LAUNCH_CHECK();
    finish_kernel<<<1, 256, 0, s>>>(partials, gsz, scale, loss_out);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_axpy_diff(const float *a, const float *b, size_t n, float coef, int accumulate, float *g,
                              st3d_stream_t stream) {
    ST3D_CHECK_ARG(a && b && g && n > 0);
    axpy_diff_kernel<<<grid_for(n), 256, 0, st3d::as_stream(stream)>>>(a, b, n, coef, accumulate, g);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_masked_mse(const float *rendered, const float *target, const float *mask, int B, int S,
                               float *grad_rendered, float *partials, float *loss_out, st3d_stream_t stream) {
    ST3D_CHECK_ARG(rendered && target && mask && partials && loss_out);
    ST3D_CHECK_ARG(B > 0 && S > 0);
    hipStream_t s = st3d::as_stream(stream);
    const size_t HW = (size_t)S * S, n = (size_t)B * 3 * HW;
    const int gsz = grid_for(n);
    ST3D_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));
    masked_mse_kernel<<<gsz, 256, 0, s>>>(rendered, target, mask, B, HW, 2.0f / (float)n, grad_rendered, partials);
    ST3D_LAUNCH_CHECK();
    finish_kernel<<<1, 256, 0, s>>>(partials, gsz, 1.0f / (float)n, loss_out);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step,
                              float lr, float beta1, float beta2, float eps, st3d_stream_t stream) {
    ST3D_CHECK_ARG(param && grad && exp_avg && exp_avg_sq);
    ST3D_CHECK_ARG(n > 0 && step >= 1);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    adam_kernel<<<grid_for(n), 256, 0, st3d::as_stream(stream)>>>(param, grad, exp_avg, exp_avg_sq, n, step_size, bc2_sqrt,
                                                                 beta1, beta2, eps);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
