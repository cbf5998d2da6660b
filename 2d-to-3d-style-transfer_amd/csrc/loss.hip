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

// Several squared-difference sums in one launch pair (the loss tail of the plan: content + five style terms used to be
// thirteen tiny launches).  Block -> item through the prefix table; every item keeps the decomposition, the partials and
// the finishing tree it has alone (sqdiff_kernel + finish_kernel), and the items are folded into their slots in order,
// so the sums are bitwise what the separate launches give.
struct MultiArgs {
    const float *a[8]; const float *b[8]; float *D[8];
    size_t n[8], nb[8];
    float scale[8];
    int slot[8], first_block[9], count;
};

__global__ __launch_bounds__(256) void sqdiff_multi_kernel(const MultiArgs m, float *__restrict__ partials) {
    int k = 0;
    while (k + 1 < m.count && (int)blockIdx.x >= m.first_block[k + 1]) ++k;
    const int lb = blockIdx.x - m.first_block[k], gsz = m.first_block[k + 1] - m.first_block[k];
    const float *a = m.a[k], *b = m.b[k];
    float *D = m.D[k];
    const size_t n = m.n[k], nb = m.nb[k];
    float acc = 0.f;
    const size_t stride = (size_t)gsz * blockDim.x;
    for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - b[nb == n ? i : i % nb];
        if (D) D[i] = d;
        acc += d * d;
    }
    __shared__ float s[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)k * NPART + lb] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void finish_multi_kernel(const MultiArgs m, const float *__restrict__ partials, float *__restrict__ out,
                                                           int zero_first, int combine, float sw, float cw) {
    __shared__ double s[256];
    __shared__ float acc3[3];
    if (threadIdx.x < 3) acc3[threadIdx.x] = zero_first ? 0.f : out[threadIdx.x];
    __syncthreads();
    for (int k = 0; k < m.count; ++k) {
        const int np = m.first_block[k + 1] - m.first_block[k];
        double v = 0.0;
        for (int i = threadIdx.x; i < np; i += 256) v += (double)partials[(size_t)k * NPART + i];
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) acc3[m.slot[k]] = acc3[m.slot[k]] + (float)(s[0] * (double)m.scale[k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (combine) acc3[0] = cw * acc3[1] + sw * acc3[2];
        out[0] = acc3[0]; out[1] = acc3[1]; out[2] = acc3[2];
    }
}

// gate: also zero the result where a <= 0 (a is a post-ReLU activation: the ReLU gate of the gradient that lives in g)
__global__ __launch_bounds__(256) void axpy_diff_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                        float coef, int accumulate, int gate, float *__restrict__ g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float ai = a[i];
        const float v = coef * (ai - b[i]);
        const float t = accumulate ? g[i] + v : v;
        g[i] = (gate && !(ai > 0.f)) ? 0.f : t;
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

// ---- regularisers the reference defines but keeps switched off (losses.py:48-65) and the L2-to-original-texture
// idea of notes.txt:39: fused forward + gradient, ordered reductions like the rest of this file.

// masked anisotropic L1 total variation: partials[blk] = sum |I(y,x)-I(y+1,x)| m(y,x) m(y+1,x) + (same along x);
// partials[NPART + blk] = sum of the mask (counted once per pixel, channel 0 lanes only)
__global__ __launch_bounds__(256) void tv_kernel(const float *__restrict__ img, const float *__restrict__ mask, int B, int C,
                                                 int H, int W, float *__restrict__ partials) {
    float acc = 0.f, macc = 0.f;
    const size_t HW = (size_t)H * W, n = (size_t)B * C * HW;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t bc = i / HW, p = i - bc * HW, b = bc / C;
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const float *mb = mask + b * HW;
        const float a = img[i], m = mb[p];
        if (y + 1 < H) acc += fabsf(a - img[i + W]) * (m * mb[p + W]);
        if (x + 1 < W) acc += fabsf(a - img[i + 1]) * (m * mb[p + 1]);
        if (bc - b * C == 0) macc += m;
    }
    __shared__ float s2[4];
    macc = wave_sum(macc);
    if ((threadIdx.x & 63) == 0) s2[threadIdx.x >> 6] = macc;
    block_store_partial(acc, partials);             // (its barrier also publishes s2)
    if (threadIdx.x == 0) partials[NPART + blockIdx.x] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
}

__global__ __launch_bounds__(256) void tv_finish_kernel(const float *__restrict__ partials, int np, float *__restrict__ out) {
    __shared__ double s[256], sm[256];
    double v = 0.0, m = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) { v += (double)partials[i]; m += (double)partials[NPART + i]; }
    s[threadIdx.x] = v; sm[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s[threadIdx.x] += s[threadIdx.x + o]; sm[threadIdx.x] += sm[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(s[0] / sm[0]); out[1] = (float)sm[0]; }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// d loss / d img = (sum of the signs of the four differences the pixel takes part in, each weighted by its mask pair) / sum(mask)
__global__ __launch_bounds__(256) void tv_grad_kernel(const float *__restrict__ img, const float *__restrict__ mask, int B, int C,
                                                      int H, int W, const float *__restrict__ loss_and_msum,
                                                      float *__restrict__ g) {
    const size_t HW = (size_t)H * W, n = (size_t)B * C * HW;
    const float inv = 1.0f / loss_and_msum[1];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t bc = i / HW, p = i - bc * HW, b = bc / C;
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const float *mb = mask + b * HW;
        const float a = img[i], m = mb[p];
        float d = 0.f;
        if (y + 1 < H) d += sgn(a - img[i + W]) * (m * mb[p + W]);
        if (y > 0) d -= sgn(img[i - W] - a) * (mb[p - W] * m);
        if (x + 1 < W) d += sgn(a - img[i + 1]) * (m * mb[p + 1]);
        if (x > 0) d -= sgn(img[i - 1] - a) * (mb[p - 1] * m);
        g[i] = d * inv;
    }
}

// sum relu(t - 1) + relu(-t); gradient +1 above 1, -1 below 0
__global__ __launch_bounds__(256) void range_kernel(const float *__restrict__ t, size_t n, float *__restrict__ g,
                                                    float *__restrict__ partials) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = t[i];
        acc += fmaxf(v - 1.0f, 0.f) + fmaxf(-v, 0.f);
        if (g) g[i] = v > 1.0f ? 1.0f : (v < 0.f ? -1.0f : 0.f);
    }
    block_store_partial(acc, partials);
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

extern "C" int st3d_sqdiff_sum_multi(const st3d_sqdiff_item *items, int count, float *partials, float *loss_out3, int zero_first,
                                     int combine, float style_weight, float content_weight, st3d_stream_t stream) {
    ST3D_CHECK_ARG(items && partials && loss_out3 && count > 0 && count <= 8);
    MultiArgs m;
    memset(&m, 0, sizeof(m));
    m.count = count;
    for (int k = 0; k < count; ++k) {
        ST3D_CHECK_ARG(items[k].a && items[k].b && items[k].n > 0 && items[k].nb > 0 && items[k].n % items[k].nb == 0);
        ST3D_CHECK_ARG(items[k].slot >= 0 && items[k].slot <= 2);
        m.a[k] = items[k].a; m.b[k] = items[k].b; m.D[k] = items[k].D;
        m.n[k] = items[k].n; m.nb[k] = items[k].nb; m.scale[k] = items[k].scale; m.slot[k] = items[k].slot;
        m.first_block[k + 1] = m.first_block[k] + grid_for(items[k].n);
    }
    hipStream_t s = st3d::as_stream(stream);
    sqdiff_multi_kernel<<<m.first_block[count], 256, 0, s>>>(m, partials);
    ST3D_LAUNCH_CHECK();
    finish_multi_kernel<<<1, 256, 0, s>>>(m, partials, loss_out3, zero_first, combine, style_weight, content_weight);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_axpy_diff(const float *a, const float *b, size_t n, float coef, int accumulate, float *g,
                              st3d_stream_t stream) {
    ST3D_CHECK_ARG(a && b && g && n > 0);
    axpy_diff_kernel<<<grid_for(n), 256, 0, st3d::as_stream(stream)>>>(a, b, n, coef, accumulate, 0, g);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_axpy_diff_gated(const float *a, const float *b, size_t n, float coef, int accumulate, float *g,
                                    st3d_stream_t stream) {
    ST3D_CHECK_ARG(a && b && g && n > 0);
    axpy_diff_kernel<<<grid_for(n), 256, 0, st3d::as_stream(stream)>>>(a, b, n, coef, accumulate, 1, g);
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

extern "C" int st3d_tv_loss(const float *images, const float *masks, int B, int C, int H, int W, float *partials,
                            float *loss_and_mask_sum, float *grad_images, st3d_stream_t stream) {
    ST3D_CHECK_ARG(images && masks && partials && loss_and_mask_sum);
    ST3D_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = st3d::as_stream(stream);
    const size_t n = (size_t)B * C * H * W;
    const int gsz = grid_for(n);
    tv_kernel<<<gsz, 256, 0, s>>>(images, masks, B, C, H, W, partials);
    ST3D_LAUNCH_CHECK();
    tv_finish_kernel<<<1, 256, 0, s>>>(partials, gsz, loss_and_mask_sum);
    ST3D_LAUNCH_CHECK();
    if (grad_images) {
        tv_grad_kernel<<<gsz, 256, 0, s>>>(images, masks, B, C, H, W, loss_and_mask_sum, grad_images);
        ST3D_LAUNCH_CHECK();
    }
    return ST3D_OK;
}

extern "C" int st3d_range_loss(const float *values, size_t n, float *partials, float *loss_out, float *grad,
                               st3d_stream_t stream) {
    ST3D_CHECK_ARG(values && partials && loss_out && n > 0);
    hipStream_t s = st3d::as_stream(stream);
    const int gsz = grid_for(n);
    ST3D_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));
    range_kernel<<<gsz, 256, 0, s>>>(values, n, grad, partials);
    ST3D_LAUNCH_CHECK();
    finish_kernel<<<1, 256, 0, s>>>(partials, gsz, 1.0f, loss_out);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
