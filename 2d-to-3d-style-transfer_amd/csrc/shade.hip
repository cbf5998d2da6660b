// shade.hip -- fused texture sampling + ambient shading + K=1 softmax blend, forward and
// backward: the device work of PyTorch3D SoftPhongShader(AmbientLights) +
// TexturesUV.sample_textures as configured at first_approach.py:108-113 /
// second_approach.py:102-108, plus the RGB / mask extraction of utils.py:70-72.
//
// The reference runs ~15 elementwise/gather launches per view here and materialises
// (1,S,S,1,3) texels, (1,S,S,4) RGBA, permutes and stacks; these kernels read the fragments
// once and write NCHW RGB + mask directly (coalesced per colour plane).
// HBM-bound: algorithmic bytes per pixel = 24 B fragments + 16 B written (fwd),
// 24 + 12 B read (bwd) + <= 12 float atomics per covered pixel into the 3*T*T*4-byte map.
// Built with -ffp-contract=off (same operation sequence as oracle/raster_ref.c).
#include <type_traits>

#include "common.h"
#include "det.h"

namespace {

constexpr float kSigma = 1e-4f, kGamma = 1e-4f, kBlendEps = 1e-10f, kZnear = 1.0f, kZfar = 100.0f;

struct Footprint {
    int x0, x1, r0, r1;
    float wx0, wx1, wy0, wy1;
    bool vx0, vx1, vy0, vy1, cx, cy;
};

// UV -> bilinear footprint in ORIGINAL texture rows: grid = uv*2-1, map flipped vertically,
// grid_sample(bilinear, align_corners=True, padding_mode='border').
__device__ __forceinline__ Footprint uv_footprint(float u, float v, int T) {
    Footprint o;
    const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
    float ix = ((gx + 1.0f) / 2.0f) * (float)(T - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(T - 1);
    o.cx = false; o.cy = false;
    if (!(ix >= 0.f)) { ix = 0.f; o.cx = true; } else if (ix > (float)(T - 1)) { ix = (float)(T - 1); o.cx = true; }
    if (!(iy >= 0.f)) { iy = 0.f; o.cy = true; } else if (iy > (float)(T - 1)) { iy = (float)(T - 1); o.cy = true; }
    const float fx = floorf(ix), fy = floorf(iy);
    o.x0 = (int)fx; o.x1 = o.x0 + 1;
    const int yf0 = (int)fy, yf1 = yf0 + 1;
    o.wx1 = ix - fx; o.wx0 = 1.0f - o.wx1;
    o.wy1 = iy - fy; o.wy0 = 1.0f - o.wy1;
    o.vx0 = o.x0 >= 0 && o.x0 < T; o.vx1 = o.x1 >= 0 && o.x1 < T;
    o.vy0 = yf0 >= 0 && yf0 < T;   o.vy1 = yf1 >= 0 && yf1 < T;
    o.r0 = (T - 1) - yf0; o.r1 = (T - 1) - yf1;
    return o;
}

struct Blend { float prob, wnum, delta, denom; };

__device__ __forceinline__ Blend blend_k1(float dist, float z) {
    Blend o;
    o.prob = 1.0f / (1.0f + expf(dist / kSigma));
    const float z_inv = (kZfar - z) / (kZfar - kZnear);
    const float z_max = fmaxf(z_inv, kBlendEps);
    o.wnum = o.prob * expf((z_inv - z_max) / kGamma);
    o.delta = fmaxf(expf((kBlendEps - z_max) / kGamma), kBlendEps);
    o.denom = o.wnum + o.delta;
    return o;
}

__global__ __launch_bounds__(256) void shade_fwd_kernel(const int32_t *__restrict__ p2f, const float *__restrict__ bary,
                                                        const float *__restrict__ zbuf, const float *__restrict__ dists,
                                                        const float *__restrict__ uvs, const int32_t *__restrict__ fuv,
                                                        const float *__restrict__ tex, int B, int S, int T,
                                                        float *__restrict__ rgb, float *__restrict__ mask) {
    const size_t HW = (size_t)S * S;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i - b * HW;
    float *o = rgb + b * 3 * HW + p;
    const int f = p2f[i];
    if (f < 0) {
        o[0] = 1.f; o[HW] = 1.f; o[2 * HW] = 1.f; mask[i] = 0.f;
        return;
    }
    const float b0 = bary[3 * i], b1 = bary[3 * i + 1], b2 = bary[3 * i + 2];
    const int u0 = fuv[3 * f], u1 = fuv[3 * f + 1], u2 = fuv[3 * f + 2];
    const float u = b0 * uvs[2 * u0] + b1 * uvs[2 * u1] + b2 * uvs[2 * u2];
    const float v = b0 * uvs[2 * u0 + 1] + b1 * uvs[2 * u1 + 1] + b2 * uvs[2 * u2 + 1];
    const Footprint q = uv_footprint(u, v, T);
    const Blend bl = blend_k1(dists[i], zbuf[i]);
    const float w00 = q.wx0 * q.wy0, w01 = q.wx1 * q.wy0, w10 = q.wx0 * q.wy1, w11 = q.wx1 * q.wy1;
    const float *t00 = tex + ((size_t)q.r0 * T + q.x0) * 3, *t01 = tex + ((size_t)q.r0 * T + q.x1) * 3;
    const float *t10 = tex + ((size_t)q.r1 * T + q.x0) * 3, *t11 = tex + ((size_t)q.r1 * T + q.x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = 0.f;
        if (q.vy0 && q.vx0) t += t00[c] * w00;
        if (q.vy0 && q.vx1) t += t01[c] * w01;
        if (q.vy1 && q.vx0) t += t10[c] * w10;
        if (q.vy1 && q.vx1) t += t11[c] * w11;
        o[c * HW] = (bl.wnum * t + bl.delta * 1.0f) / bl.denom;
    }
    mask[i] = ((1.0f - (1.0f - bl.prob)) > 0.f) ? 1.f : 0.f;
}

// Texture-sampling backward.  One workgroup per 16x16-pixel tile of one view.  The <= 12 bilinear contributions of a
// pixel are not sent to HBM one float atomic each (neighbouring pixels hit the same texels: ~6 M contended L2 atomics
// per step at config 2): they are first summed per texel in an LDS table (open addressing on the texel index,
// ds_add_f32) and each distinct texel of the tile then costs three global atomics.  d/d(u,v) and d/d(bary) (vertex
// path) are per-pixel outputs written directly.
constexpr int kTexSlots = 2048;          // >= 4 x 256 footprint corners: the probe always terminates

// DET 0: float LDS table + float global atomics (fast default).  DET 1: the same binning in 64-bit fixed point (LDS and
// global integer atomics; `gtex` is then the int64 accumulator array and `det` holds the power-of-two scale): bitwise
// reproducible whatever the order (det.h).
template <int DET>
__global__ __launch_bounds__(256) void shade_bwd_kernel(const float *__restrict__ grad_rgb, const int32_t *__restrict__ p2f,
                                                        const float *__restrict__ bary, const float *__restrict__ zbuf,
                                                        const float *__restrict__ dists, const float *__restrict__ uvs,
                                                        const int32_t *__restrict__ fuv, const float *__restrict__ tex,
                                                        int B, int S, int T, int tiles_x, float *__restrict__ gtex,
                                                        float *__restrict__ guv, float *__restrict__ gbary,
                                                        const st3d_det::DetHeader *__restrict__ det) {
    typedef typename std::conditional<DET != 0, unsigned long long, float>::type acc_t;
    __shared__ int s_key[kTexSlots];
    __shared__ acc_t s_acc[kTexSlots][3];
    const int tid = threadIdx.x;
    const double dscale = DET ? det->scale : 1.0;
    if (gtex) {
        for (int e = tid; e < kTexSlots; e += 256) s_key[e] = -1;
        for (int e = tid; e < kTexSlots * 3; e += 256) (&s_acc[0][0])[e] = (acc_t)0;
        __syncthreads();
    }
    const size_t HW = (size_t)S * S;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int yi = ty * 16 + (tid >> 4), xi = tx * 16 + (tid & 15);
    const bool in_img = yi < S && xi < S;
    const size_t p = (size_t)yi * S + xi, i = (size_t)b * HW + p;
    const int f = in_img ? p2f[i] : -1;
    if (in_img && f < 0) {
        if (guv) { guv[2 * i] = 0.f; guv[2 * i + 1] = 0.f; }
        if (gbary) { gbary[3 * i] = 0.f; gbary[3 * i + 1] = 0.f; gbary[3 * i + 2] = 0.f; }
    }
    if (f >= 0) {
        const bool want_uv = guv || gbary;
        const float b0 = bary[3 * i], b1 = bary[3 * i + 1], b2 = bary[3 * i + 2];
        const int u0 = fuv[3 * f], u1 = fuv[3 * f + 1], u2 = fuv[3 * f + 2];
        const float u = b0 * uvs[2 * u0] + b1 * uvs[2 * u1] + b2 * uvs[2 * u2];
        const float v = b0 * uvs[2 * u0 + 1] + b1 * uvs[2 * u1 + 1] + b2 * uvs[2 * u2 + 1];
        const Footprint q = uv_footprint(u, v, T);
        const Blend bl = blend_k1(dists[i], zbuf[i]);
        const float k = bl.wnum / bl.denom;
        const float w00 = q.wx0 * q.wy0, w01 = q.wx1 * q.wy0, w10 = q.wx0 * q.wy1, w11 = q.wx1 * q.wy1;
        const int e00 = q.r0 * T + q.x0, e01 = q.r0 * T + q.x1, e10 = q.r1 * T + q.x0, e11 = q.r1 * T + q.x1;
        const float *g = grad_rgb + (size_t)b * 3 * HW + p;
        const float gk[3] = {g[0] * k, g[HW] * k, g[2 * HW] * k};
        if (want_uv) {
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t00 = (q.vy0 && q.vx0) ? tex[(size_t)e00 * 3 + c] : 0.f, t01 = (q.vy0 && q.vx1) ? tex[(size_t)e01 * 3 + c] : 0.f;
                const float t10 = (q.vy1 && q.vx0) ? tex[(size_t)e10 * 3 + c] : 0.f, t11 = (q.vy1 && q.vx1) ? tex[(size_t)e11 * 3 + c] : 0.f;
                gix += gk[c] * ((t01 - t00) * q.wy0 + (t11 - t10) * q.wy1);
                giy += gk[c] * ((t10 - t00) * q.wx0 + (t11 - t01) * q.wx1);
            }
            const float gu = q.cx ? 0.f : gix * (float)(T - 1);
            const float gv = q.cy ? 0.f : giy * (float)(T - 1);
            if (guv) { guv[2 * i] = gu; guv[2 * i + 1] = gv; }
            if (gbary) {      // uv = sum_i b_i * uv_i
                gbary[3 * i] = gu * uvs[2 * u0] + gv * uvs[2 * u0 + 1];
                gbary[3 * i + 1] = gu * uvs[2 * u1] + gv * uvs[2 * u1 + 1];
                gbary[3 * i + 2] = gu * uvs[2 * u2] + gv * uvs[2 * u2 + 1];
            }
        }
        if (gtex) {
            auto deposit = [&](int texel, float w) __attribute__((always_inline)) {
                int slot = (int)(((unsigned)texel * 2654435761u) >> 21) & (kTexSlots - 1);
                for (;;) {
                    const int prev = atomicCAS(&s_key[slot], -1, texel);
                    if (prev == -1 || prev == texel) break;
                    slot = (slot + 1) & (kTexSlots - 1);
                }
                if (DET) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        atomicAdd(reinterpret_cast<unsigned long long *>(&s_acc[slot][c]),
                                  (unsigned long long)st3d_det::det_quantise(gk[c] * w, dscale));
                } else {
                    atomicAdd(reinterpret_cast<float *>(&s_acc[slot][0]), gk[0] * w);
                    atomicAdd(reinterpret_cast<float *>(&s_acc[slot][1]), gk[1] * w);
                    atomicAdd(reinterpret_cast<float *>(&s_acc[slot][2]), gk[2] * w);
                }
            };
            if (q.vy0 && q.vx0) deposit(e00, w00);
            if (q.vy0 && q.vx1) deposit(e01, w01);
            if (q.vy1 && q.vx0) deposit(e10, w10);
            if (q.vy1 && q.vx1) deposit(e11, w11);
        }
    }
    if (gtex) {
        __syncthreads();
        for (int e = tid; e < kTexSlots * 3; e += 256) {
            const int slot = e / 3, c = e - slot * 3;
            const int texel = s_key[slot];
            if (texel < 0) continue;
            const acc_t v = s_acc[slot][c];
            if (v == (acc_t)0) continue;
            if (DET) atomicAdd(reinterpret_cast<unsigned long long *>(gtex) + (size_t)texel * 3 + c, (unsigned long long)v);
            else atomicAdd(gtex + (size_t)texel * 3 + c, (float)v);
        }
    }
}

// out = img*mask + bg*(1-mask)   (utils.py:23,27); bg == nullptr: out = img*mask
__global__ __launch_bounds__(256) void background_kernel(const float *__restrict__ img, const float *__restrict__ mask,
                                                         const float *__restrict__ bg, int bg_batch, int B, size_t HW,
                                                         float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * 3 * HW) return;
    const size_t b = i / (3 * HW), r = i - b * 3 * HW, p = r % HW;
    const float m = mask[b * HW + p];
    float v = img[i] * m;
    if (bg) v = v + bg[(bg_batch == 1 ? 0 : b * 3 * HW) + r] * (1.0f - m);
    out[i] = v;
}

}  // namespace

extern "C" int st3d_shade_fwd(const int32_t *pix_to_face, const float *bary, const float *zbuf, const float *dists,
                              const float *verts_uvs, const int32_t *faces_uvs, const float *texture, int B, int S, int T,
                              int F, int VT, float *rgb, float *mask, st3d_stream_t stream) {
    ST3D_CHECK_ARG(pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture && rgb && mask);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && F > 0 && VT > 0);
    const size_t n = (size_t)B * S * S;
    shade_fwd_kernel<<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(pix_to_face, bary, zbuf, dists, verts_uvs,
                                                                                   faces_uvs, texture, B, S, T, rgb, mask);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_shade_bwd(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                              const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                              int B, int S, int T, int F, int VT, float *grad_texture, float *grad_uv,
                              float *grad_bary, st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_rgb && pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture);
    ST3D_CHECK_ARG(grad_texture || grad_uv || grad_bary);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && F > 0 && VT > 0);
    const size_t n = (size_t)B * S * S;
    const int tiles = (S + 15) / 16;
    shade_bwd_kernel<0><<<dim3(tiles * tiles, B), 256, 0, st3d::as_stream(stream)>>>(
        grad_rgb, pix_to_face, bary, zbuf, dists, verts_uvs, faces_uvs, texture, B, S, T, tiles, grad_texture, grad_uv, grad_bary,
        nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

namespace {
// (bound on any texel-channel sum: sum over the batch of |grad_rgb| -- blend weight and bilinear weights are <= 1: st3d_det::det_abs_sum_kernel)
constexpr int kDetPartials = 1024;
}  // namespace

extern "C" size_t st3d_shade_bwd_det_workspace_bytes(int T) {
    return st3d_det::workspace_bytes((size_t)T * T * 3, kDetPartials);
}

// st3d_shade_bwd with a bitwise reproducible texture gradient: fixed-point accumulation (det.h); measured no slower than the
// float-atomic scatter (0.128 vs 0.137 ms at config 2), so it is the default of the Python host (st3d/ops.py).  grad_texture is ACCUMULATED into, like st3d_shade_bwd.
extern "C" int st3d_shade_bwd_det(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                                  const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                                  int B, int S, int T, int F, int VT, float *grad_texture, float *grad_uv, float *grad_bary,
                                  void *workspace, size_t workspace_bytes, st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_rgb && pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture && workspace);
    ST3D_CHECK_ARG(grad_texture);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && F > 0 && VT > 0);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_shade_bwd_det_workspace_bytes(T) && ((uintptr_t)workspace & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    auto *hdr = reinterpret_cast<st3d_det::DetHeader *>(workspace);
    float *partials = st3d_det::partials_of(workspace);
    long long *acc = st3d_det::accum_of(workspace, kDetPartials);
    const size_t npx = (size_t)B * 3 * S * S, nacc = (size_t)T * T * 3;
    st3d_det::det_abs_sum_kernel<<<kDetPartials, 256, 0, s>>>(grad_rgb, npx, partials);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_scale_kernel<<<1, 256, 0, s>>>(partials, kDetPartials, hdr);
    ST3D_LAUNCH_CHECK();
    ST3D_HIP(hipMemsetAsync(acc, 0, nacc * sizeof(long long), s));
    const int tiles = (S + 15) / 16;
    shade_bwd_kernel<1><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_rgb, pix_to_face, bary, zbuf, dists, verts_uvs, faces_uvs,
                                                                texture, B, S, T, tiles, reinterpret_cast<float *>(acc), grad_uv,
                                                                grad_bary, hdr);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_convert_kernel<<<st3d::cdiv((long)nacc, 256), 256, 0, s>>>(acc, nacc, hdr, 1, grad_texture);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_apply_background(const float *img, const float *mask, const float *bg, int bg_batch, int B, int S,
                                     float *out, st3d_stream_t stream) {
    ST3D_CHECK_ARG(img && mask && out);
    ST3D_CHECK_ARG(B > 0 && S > 0 && (bg_batch == 1 || bg_batch == B));
    const size_t n = (size_t)B * 3 * S * S;
    background_kernel<<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(img, mask, bg, bg_batch, B,
                                                                                   (size_t)S * S, out);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
