/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or
 * called from the product path; see DESIGN.md "Oracle").
 *
 * Plain-C CPU restatement of the render half of the reference hot path:
 *   utils.py:65-77 render_meshes  ->  PyTorch3D MeshRasterizer + SoftPhongShader +
 *   TexturesUV.sample_textures + softmax_rgb_blend, with the settings the reference
 *   fixes at first_approach.py:107-113 / second_approach.py:101-108
 *   (blur_radius=0.0, faces_per_pixel=1, AmbientLights, FoVPerspectiveCameras defaults).
 *
 * PyTorch3D is a third-party dependency that is NOT vendored under /root/reference and is
 * not installed (version unpinned: no requirements/lock file).  The arithmetic below
 * restates its published naive-CPU algorithm (rasterize_meshes_cpu.cpp, geometry_utils.h,
 * blending.py softmax_rgb_blend, TexturesUV.sample_textures + ATen grid_sampler_2d) as
 * summarised in SURVEY.md Appendix A.  PARITY UNPINNED for this file: the reference holds
 * no golden vectors for the render path; it is pinned by analytic tests only
 * (tests/test_oracle_raster.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so the HIP
 * kernels, built the same way, can be compared bit-for-bit on the integer/geometry outputs).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K_EPS 1e-8f

/* ---------------------------------------------------------------- cameras (A.1) */

/* verts (V,3) world; R (3,3) row-major, T (3): X_view = X_world . R + T  (row-vector
 * convention, utils.py:142-149,161-168).  out (V,3) = (x_ndc, y_ndc, z_view) with
 * x_ndc = s*x_v/z_v, s = 1/tan(fov/2) (FoVPerspectiveCameras defaults fov=60deg, aspect 1). */
void ref_project_verts(const float *verts, int V, const float *R, const float *T,
                       float inv_tan_half_fov, float *out)
{
    for (int v = 0; v < V; ++v) {
        const float x = verts[3 * v + 0], y = verts[3 * v + 1], z = verts[3 * v + 2];
        const float xv = x * R[0] + y * R[3] + z * R[6] + T[0];
        const float yv = x * R[1] + y * R[4] + z * R[7] + T[1];
        const float zv = x * R[2] + y * R[5] + z * R[8] + T[2];
        out[3 * v + 0] = (inv_tan_half_fov * xv) / zv;
        out[3 * v + 1] = (inv_tan_half_fov * yv) / zv;
        out[3 * v + 2] = zv;
    }
}

/* ---------------------------------------------------------------- geometry (A.2) */

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by)
{
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float point_line_dist2(float px, float py, float ax, float ay, float bx, float by)
{
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= K_EPS) {
        const float dx = px - bx, dy = py - by;
        return dx * dx + dy * dy;
    }
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

static inline float pix_to_ndc(int i, int S)
{
    return -1.0f + (2.0f * (float)i + 1.0f) / (float)S;
}

/*
 * Naive rasteriser, K = faces_per_pixel = 1, perspective_correct = 1, no bary clipping,
 * no back-face culling, bbox padded by sqrt(blur_radius).
 * verts_ndc (V,3) from ref_project_verts; faces (F,3) int32.
 * Outputs (S,S): pix_to_face int32 (-1 = background), zbuf, bary (S,S,3), dists (signed
 * squared distance to the nearest edge; negative inside); -1 fill elsewhere.
 */
static void rasterize_impl(const float *verts_ndc, const int32_t *faces, int F, int S,
                           float blur_radius, int nthreads, int row_lists,
                           int32_t *pix_to_face, float *zbuf, float *bary, float *dists)
{
    const float pad = sqrtf(blur_radius);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int yi = 0; yi < S; ++yi) {
        const float yf = pix_to_ndc(S - 1 - yi, S);
        /* row_lists: the faces whose padded y-extent contains this row's centre, in face order -- the SAME y test
         * the per-pixel loop below repeats, evaluated once per row, so the result is identical to the plain
         * all-faces loop (checked bit for bit in tests/test_oracle_raster.py); it only skips work. */
        int32_t *row = NULL;
        int nrow = F;
        if (row_lists) {
            row = (int32_t *)malloc((size_t)(F > 0 ? F : 1) * sizeof(int32_t));
            nrow = 0;
            for (int f = 0; f < F; ++f) {
                const float y0 = verts_ndc[3 * faces[3 * f] + 1], y1 = verts_ndc[3 * faces[3 * f + 1] + 1];
                const float y2 = verts_ndc[3 * faces[3 * f + 2] + 1];
                const float ymin = fminf(y0, fminf(y1, y2)) - pad, ymax = fmaxf(y0, fmaxf(y1, y2)) + pad;
                if (yf > ymax || yf < ymin) continue;
                row[nrow++] = f;
            }
        }
        for (int xi = 0; xi < S; ++xi) {
            const float xf = pix_to_ndc(S - 1 - xi, S);
            int best_f = -1;
            float best_z = 0.f, best_d = 0.f, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
            for (int k = 0; k < nrow; ++k) {
                const int f = row ? row[k] : k;
                const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
                const float x0 = verts_ndc[3 * i0], y0 = verts_ndc[3 * i0 + 1], z0 = verts_ndc[3 * i0 + 2];
                const float x1 = verts_ndc[3 * i1], y1 = verts_ndc[3 * i1 + 1], z1 = verts_ndc[3 * i1 + 2];
                const float x2 = verts_ndc[3 * i2], y2 = verts_ndc[3 * i2 + 1], z2 = verts_ndc[3 * i2 + 2];
                const float xmin = fminf(x0, fminf(x1, x2)) - pad, xmax = fmaxf(x0, fmaxf(x1, x2)) + pad;
                const float ymin = fminf(y0, fminf(y1, y2)) - pad, ymax = fmaxf(y0, fmaxf(y1, y2)) + pad;
                if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                const float zmax = fmaxf(z0, fmaxf(z1, z2));
                if (zmax < K_EPS) continue;                       /* face fully behind the camera */
                const float face_area = edge_fn(x2, y2, x0, y0, x1, y1);
                if (face_area <= K_EPS && face_area >= -K_EPS) continue;
                const float area = face_area + K_EPS;
                const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                /* perspective correction */
                const float t0 = w0 * z1 * z2;
                const float t1 = z0 * w1 * z2;
                const float t2 = z0 * z1 * w2;
                const float den = fmaxf(t0 + t1 + t2, K_EPS);
                const float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
                const float pz = b0 * z0 + b1 * z1 + b2 * z2;
                if (pz < 0.f) continue;
                const int inside = (b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f);
                if (!inside) {
                    /* blur_radius == 0: an outside pixel can never satisfy d < blur. */
                    if (blur_radius <= 0.f) continue;
                }
                const float d01 = point_line_dist2(xf, yf, x0, y0, x1, y1);
                const float d12 = point_line_dist2(xf, yf, x1, y1, x2, y2);
                const float d20 = point_line_dist2(xf, yf, x2, y2, x0, y0);
                const float d = fminf(d01, fminf(d12, d20));
                if (!inside && d >= blur_radius) continue;
                const float sd = inside ? -d : d;
                if (best_f < 0 || pz < best_z) {            /* ties keep the smaller face index */
                    best_f = f; best_z = pz; best_d = sd; bw0 = b0; bw1 = b1; bw2 = b2;
                }
            }
            const size_t p = (size_t)yi * S + xi;
            if (best_f >= 0) {
                pix_to_face[p] = best_f; zbuf[p] = best_z; dists[p] = best_d;
                bary[3 * p] = bw0; bary[3 * p + 1] = bw1; bary[3 * p + 2] = bw2;
            } else {
                pix_to_face[p] = -1; zbuf[p] = -1.f; dists[p] = -1.f;
                bary[3 * p] = -1.f; bary[3 * p + 1] = -1.f; bary[3 * p + 2] = -1.f;
            }
        }
        free(row);
    }
}

void ref_rasterize(const float *verts_ndc, const int32_t *faces, int F, int S,
                   float blur_radius, int nthreads,
                   int32_t *pix_to_face, float *zbuf, float *bary, float *dists)
{
    rasterize_impl(verts_ndc, faces, F, S, blur_radius, nthreads, 1, pix_to_face, zbuf, bary, dists);
}

/* the plain loop over all faces for every pixel (what PyTorch3D's CPU rasteriser does) */
void ref_rasterize_naive(const float *verts_ndc, const int32_t *faces, int F, int S,
                         float blur_radius, int nthreads,
                         int32_t *pix_to_face, float *zbuf, float *bary, float *dists)
{
    rasterize_impl(verts_ndc, faces, F, S, blur_radius, nthreads, 0, pix_to_face, zbuf, bary, dists);
}

/* ---------------------------------------------------------------- texture sampling (A.3) */

typedef struct { int x0, x1, r0, r1; float wx0, wx1, wy0, wy1; int vx0, vx1, vy0, vy1; } bilerp_t;

/* UV -> bilinear footprint in ORIGINAL texture rows (map flipped vertically before
 * grid_sample(align_corners=True, padding_mode='border')). */
static inline void uv_footprint(float u, float v, int T, bilerp_t *o, float *ix_out, float *iy_out,
                                int *clamped_x, int *clamped_y)
{
    const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
    float ix = ((gx + 1.0f) / 2.0f) * (float)(T - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(T - 1);
    *clamped_x = 0; *clamped_y = 0;
    if (!(ix >= 0.f)) { ix = 0.f; *clamped_x = 1; } else if (ix > (float)(T - 1)) { ix = (float)(T - 1); *clamped_x = 1; }
    if (!(iy >= 0.f)) { iy = 0.f; *clamped_y = 1; } else if (iy > (float)(T - 1)) { iy = (float)(T - 1); *clamped_y = 1; }
    const float fx = floorf(ix), fy = floorf(iy);
    o->x0 = (int)fx; o->x1 = o->x0 + 1;
    const int yf0 = (int)fy, yf1 = yf0 + 1;            /* rows of the FLIPPED map */
    o->wx1 = ix - fx; o->wx0 = 1.0f - o->wx1;         /* = (x1 - ix) */
    o->wy1 = iy - fy; o->wy0 = 1.0f - o->wy1;
    o->vx0 = o->x0 >= 0 && o->x0 < T; o->vx1 = o->x1 >= 0 && o->x1 < T;
    o->vy0 = yf0 >= 0 && yf0 < T;     o->vy1 = yf1 >= 0 && yf1 < T;
    o->r0 = (T - 1) - yf0; o->r1 = (T - 1) - yf1;     /* original rows */
    *ix_out = ix; *iy_out = iy;
}

/* blend constants: softmax_rgb_blend with BlendParams defaults sigma=gamma=1e-4, bg=(1,1,1),
 * znear=1, zfar=100 (SoftPhongShader defaults), K=1. */
#define BLEND_SIGMA 1e-4f
#define BLEND_GAMMA 1e-4f
#define BLEND_EPS   1e-10f
#define ZNEAR 1.0f
#define ZFAR  100.0f

static inline void blend_k1(float dist, float z, float *prob_out, float *wnum_out, float *delta_out, float *denom_out)
{
    const float prob = 1.0f / (1.0f + expf(dist / BLEND_SIGMA));   /* sigmoid(-dist/sigma) */
    const float z_inv = (ZFAR - z) / (ZFAR - ZNEAR);
    const float z_max = fmaxf(z_inv, BLEND_EPS);
    const float wnum = prob * expf((z_inv - z_max) / BLEND_GAMMA);
    const float delta = fmaxf(expf((BLEND_EPS - z_max) / BLEND_GAMMA), BLEND_EPS);
    *prob_out = prob; *wnum_out = wnum; *delta_out = delta; *denom_out = wnum + delta;
}

/*
 * Fused shade: UV interpolation -> bilinear sample -> ambient (colour = texel) -> softmax
 * blend (K=1) -> CHW RGB + mask exactly as utils.py:70-72 lays them out.
 * texture (T,T,3) HWC; verts_uvs (VT,2); faces_uvs (F,3) int32.
 * rgb (3,S,S), mask (S,S) = (alpha > 0).
 */
void ref_shade_fwd(const int32_t *pix_to_face, const float *bary, const float *zbuf, const float *dists,
                   const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                   int S, int T, float *rgb, float *mask)
{
    const size_t HW = (size_t)S * S;
    for (size_t p = 0; p < HW; ++p) {
        const int f = pix_to_face[p];
        if (f < 0) {
            rgb[p] = 1.f; rgb[HW + p] = 1.f; rgb[2 * HW + p] = 1.f; mask[p] = 0.f;
            continue;
        }
        const float b0 = bary[3 * p], b1 = bary[3 * p + 1], b2 = bary[3 * p + 2];
        const int u0 = faces_uvs[3 * f], u1 = faces_uvs[3 * f + 1], u2 = faces_uvs[3 * f + 2];
        const float u = b0 * verts_uvs[2 * u0] + b1 * verts_uvs[2 * u1] + b2 * verts_uvs[2 * u2];
        const float v = b0 * verts_uvs[2 * u0 + 1] + b1 * verts_uvs[2 * u1 + 1] + b2 * verts_uvs[2 * u2 + 1];
        bilerp_t q; float ix, iy; int cx, cy;
        uv_footprint(u, v, T, &q, &ix, &iy, &cx, &cy);
        float prob, wnum, delta, denom;
        blend_k1(dists[p], zbuf[p], &prob, &wnum, &delta, &denom);
        for (int c = 0; c < 3; ++c) {
            float t = 0.f;
            if (q.vy0 && q.vx0) t += texture[((size_t)q.r0 * T + q.x0) * 3 + c] * (q.wx0 * q.wy0);
            if (q.vy0 && q.vx1) t += texture[((size_t)q.r0 * T + q.x1) * 3 + c] * (q.wx1 * q.wy0);
            if (q.vy1 && q.vx0) t += texture[((size_t)q.r1 * T + q.x0) * 3 + c] * (q.wx0 * q.wy1);
            if (q.vy1 && q.vx1) t += texture[((size_t)q.r1 * T + q.x1) * 3 + c] * (q.wx1 * q.wy1);
            rgb[c * HW + p] = (wnum * t + delta * 1.0f) / denom;
        }
        mask[p] = ((1.0f - (1.0f - prob)) > 0.f) ? 1.f : 0.f;
    }
}

/*
 * Backward of ref_shade_fwd w.r.t. the texture map (double accumulation, then cast):
 * grad_rgb (3,S,S) -> grad_texture (T,T,3) ACCUMULATED (+=) so several views can be summed.
 * Optionally (grad_uv != NULL) also d loss / d (u,v) per pixel (S,S,2) for the vertex path
 * (bilinear derivative, zero where the coordinate was clamped by padding_mode='border').
 */
void ref_shade_bwd(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                   const float *dists, const float *verts_uvs, const int32_t *faces_uvs,
                   const float *texture, int S, int T, double *grad_texture, float *grad_uv)
{
    const size_t HW = (size_t)S * S;
    for (size_t p = 0; p < HW; ++p) {
        const int f = pix_to_face[p];
        if (grad_uv) { grad_uv[2 * p] = 0.f; grad_uv[2 * p + 1] = 0.f; }
        if (f < 0) continue;
        const float b0 = bary[3 * p], b1 = bary[3 * p + 1], b2 = bary[3 * p + 2];
        const int u0 = faces_uvs[3 * f], u1 = faces_uvs[3 * f + 1], u2 = faces_uvs[3 * f + 2];
        const float u = b0 * verts_uvs[2 * u0] + b1 * verts_uvs[2 * u1] + b2 * verts_uvs[2 * u2];
        const float v = b0 * verts_uvs[2 * u0 + 1] + b1 * verts_uvs[2 * u1 + 1] + b2 * verts_uvs[2 * u2 + 1];
        bilerp_t q; float ix, iy; int cx, cy;
        uv_footprint(u, v, T, &q, &ix, &iy, &cx, &cy);
        float prob, wnum, delta, denom;
        blend_k1(dists[p], zbuf[p], &prob, &wnum, &delta, &denom);
        const float k = wnum / denom;                  /* d rgb / d texel */
        double gix = 0.0, giy = 0.0;
        for (int c = 0; c < 3; ++c) {
            const float g = grad_rgb[c * HW + p] * k;
            float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
            if (q.vy0 && q.vx0) { grad_texture[((size_t)q.r0 * T + q.x0) * 3 + c] += (double)(g * (q.wx0 * q.wy0)); t00 = texture[((size_t)q.r0 * T + q.x0) * 3 + c]; }
            if (q.vy0 && q.vx1) { grad_texture[((size_t)q.r0 * T + q.x1) * 3 + c] += (double)(g * (q.wx1 * q.wy0)); t01 = texture[((size_t)q.r0 * T + q.x1) * 3 + c]; }
            if (q.vy1 && q.vx0) { grad_texture[((size_t)q.r1 * T + q.x0) * 3 + c] += (double)(g * (q.wx0 * q.wy1)); t10 = texture[((size_t)q.r1 * T + q.x0) * 3 + c]; }
            if (q.vy1 && q.vx1) { grad_texture[((size_t)q.r1 * T + q.x1) * 3 + c] += (double)(g * (q.wx1 * q.wy1)); t11 = texture[((size_t)q.r1 * T + q.x1) * 3 + c]; }
            /* d sample / d ix, d sample / d iy (flipped-map coordinates) */
            gix += (double)g * ((double)(t01 - t00) * q.wy0 + (double)(t11 - t10) * q.wy1);
            giy += (double)g * ((double)(t10 - t00) * q.wx0 + (double)(t11 - t01) * q.wx1);
        }
        if (grad_uv) {
            /* ix = u*(T-1); iy_flipped = v*(T-1) */
            grad_uv[2 * p]     = cx ? 0.f : (float)(gix * (double)(T - 1));
            grad_uv[2 * p + 1] = cy ? 0.f : (float)(giy * (double)(T - 1));
        }
    }
}

/* ---------------------------------------------------------------- Adam (torch.optim.Adam, utils.py:185-195) */

/* One dense Adam step with torch defaults (amsgrad=False, weight_decay=0, maximize=False):
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
 *   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),  bc_i = 1 - b_i^step. */
void ref_adam_step(float *p, const float *g, float *m, float *v, size_t n, int step,
                   float lr, float b1, float b2, float eps)
{
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    for (size_t i = 0; i < n; ++i) {
        const float gi = g[i];
        m[i] = m[i] + (gi - m[i]) * (1.0f - b1);          /* torch: exp_avg.lerp_(grad, 1-beta1) */
        v[i] = v[i] * b2 + (1.0f - b2) * gi * gi;        /* mul_(beta2).addcmul_(g, g, 1-beta2) */
        const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* ---------------------------------------------------------------- vertex path
 * Backward of the rasteriser's perspective-correct barycentrics w.r.t. the projected vertices,
 * and of the projection w.r.t. the world vertices -- what autograd does in the reference when
 * optimization_target is 'mesh'/'both' (utils.py:187-195, second_approach.py:188) through
 * PyTorch3D's RasterizeMeshesBackward (grad_bary path; the gradients through zbuf/dists are
 * O(1e-10) via the K=1 blend, SURVEY.md A.4, and are not propagated).
 * Accumulation in double.  PARITY UNPINNED (see the header); pinned by the fp64 autograd
 * restatement in tests/test_oracle_vertex_path.py. */

/* grad_uv (S,S,2) -> grad_bary (S,S,3): uv = sum_i b_i * uv_i */
void ref_uv_to_bary_grad(const float *grad_uv, const int32_t *pix_to_face, const float *verts_uvs,
                         const int32_t *faces_uvs, int S, float *grad_bary)
{
    const size_t HW = (size_t)S * S;
    for (size_t p = 0; p < HW; ++p) {
        const int f = pix_to_face[p];
        for (int i = 0; i < 3; ++i) {
            float g = 0.f;
            if (f >= 0) {
                const int ui = faces_uvs[3 * f + i];
                g = grad_uv[2 * p] * verts_uvs[2 * ui] + grad_uv[2 * p + 1] * verts_uvs[2 * ui + 1];
            }
            grad_bary[3 * p + i] = g;
        }
    }
}

void ref_raster_bwd(const float *grad_bary, const int32_t *pix_to_face, const float *verts_ndc,
                    const int32_t *faces, int S, double *grad_verts_ndc /* (V,3) accumulated */)
{
    for (int yi = 0; yi < S; ++yi) {
        const float yf = pix_to_ndc(S - 1 - yi, S);
        for (int xi = 0; xi < S; ++xi) {
            const size_t p = (size_t)yi * S + xi;
            const int f = pix_to_face[p];
            if (f < 0) continue;
            const float xf = pix_to_ndc(S - 1 - xi, S);
            const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
            const double x0 = verts_ndc[3 * i0], y0 = verts_ndc[3 * i0 + 1], z0 = verts_ndc[3 * i0 + 2];
            const double x1 = verts_ndc[3 * i1], y1 = verts_ndc[3 * i1 + 1], z1 = verts_ndc[3 * i1 + 2];
            const double x2 = verts_ndc[3 * i2], y2 = verts_ndc[3 * i2 + 1], z2 = verts_ndc[3 * i2 + 2];
            const double px = xf, py = yf;
            const double A = ((x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)) + (double)K_EPS;
            const double e0 = (px - x1) * (y2 - y1) - (py - y1) * (x2 - x1);
            const double e1 = (px - x2) * (y0 - y2) - (py - y2) * (x0 - x2);
            const double e2 = (px - x0) * (y1 - y0) - (py - y0) * (x1 - x0);
            const double w0 = e0 / A, w1 = e1 / A, w2 = e2 / A;
            const double t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
            const double den = t0 + t1 + t2;
            if (!(den > (double)K_EPS)) continue;                 /* clamped denominator: no gradient */
            const double b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
            const double g0 = grad_bary[3 * p], g1 = grad_bary[3 * p + 1], g2 = grad_bary[3 * p + 2];
            const double gs = g0 * b0 + g1 * b1 + g2 * b2;
            const double dt0 = (g0 - gs) / den, dt1 = (g1 - gs) / den, dt2 = (g2 - gs) / den;
            const double dw0 = dt0 * z1 * z2, dw1 = dt1 * z0 * z2, dw2 = dt2 * z0 * z1;
            const double dz0 = dt1 * w1 * z2 + dt2 * z1 * w2;
            const double dz1 = dt0 * w0 * z2 + dt2 * z0 * w2;
            const double dz2 = dt0 * w0 * z1 + dt1 * z0 * w1;
            const double de0 = dw0 / A, de1 = dw1 / A, de2 = dw2 / A;
            const double dA = -(dw0 * w0 + dw1 * w1 + dw2 * w2) / A;
            /* E(p;a,b) = (px-ax)(by-ay) - (py-ay)(bx-ax):
             *   dE/dax = py-by, dE/day = bx-px, dE/dbx = -(py-ay), dE/dby = px-ax,
             *   dE/dpx = by-ay, dE/dpy = -(bx-ax) */
            double gx0 = 0, gy0 = 0, gx1 = 0, gy1 = 0, gx2 = 0, gy2 = 0;
            /* e0 = E(p; v1, v2) */
            gx1 += de0 * (py - y2); gy1 += de0 * (x2 - px); gx2 += de0 * -(py - y1); gy2 += de0 * (px - x1);
            /* e1 = E(p; v2, v0) */
            gx2 += de1 * (py - y0); gy2 += de1 * (x0 - px); gx0 += de1 * -(py - y2); gy0 += de1 * (px - x2);
            /* e2 = E(p; v0, v1) */
            gx0 += de2 * (py - y1); gy0 += de2 * (x1 - px); gx1 += de2 * -(py - y0); gy1 += de2 * (px - x0);
            /* A = E(v2; v0, v1) + eps */
            gx0 += dA * (y2 - y1); gy0 += dA * (x1 - x2); gx1 += dA * -(y2 - y0); gy1 += dA * (x2 - x0);
            gx2 += dA * (y1 - y0); gy2 += dA * -(x1 - x0);
            grad_verts_ndc[3 * i0] += gx0; grad_verts_ndc[3 * i0 + 1] += gy0; grad_verts_ndc[3 * i0 + 2] += dz0;
            grad_verts_ndc[3 * i1] += gx1; grad_verts_ndc[3 * i1 + 1] += gy1; grad_verts_ndc[3 * i1 + 2] += dz1;
            grad_verts_ndc[3 * i2] += gx2; grad_verts_ndc[3 * i2 + 1] += gy2; grad_verts_ndc[3 * i2 + 2] += dz2;
        }
    }
}

/* backward of ref_project_verts w.r.t. the world vertices; grad_verts (V,3) accumulated */
void ref_project_verts_bwd(const float *verts, int V, const float *R, const float *T, float s,
                           const double *grad_ndc, double *grad_verts)
{
    for (int v = 0; v < V; ++v) {
        const double x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
        const double xv = x * R[0] + y * R[3] + z * R[6] + T[0];
        const double yv = x * R[1] + y * R[4] + z * R[7] + T[1];
        const double zv = x * R[2] + y * R[5] + z * R[8] + T[2];
        const double gxn = grad_ndc[3 * v], gyn = grad_ndc[3 * v + 1], gz = grad_ndc[3 * v + 2];
        const double dxv = s * gxn / zv, dyv = s * gyn / zv;
        const double dzv = gz - s * (gxn * xv + gyn * yv) / (zv * zv);
        grad_verts[3 * v + 0] += dxv * R[0] + dyv * R[1] + dzv * R[2];
        grad_verts[3 * v + 1] += dxv * R[3] + dyv * R[4] + dzv * R[5];
        grad_verts[3 * v + 2] += dxv * R[6] + dyv * R[7] + dzv * R[8];
    }
}

/* ---------------------------------------------------------------- general soft rasteriser
 * K = faces_per_pixel nearest faces per pixel, blur_radius >= 0, optional barycentric clipping
 * (PyTorch3D clips when blur_radius > 0 unless told otherwise): the general form of ref_rasterize
 * the reference does NOT use (it fixes K = 1, blur = 0) but BASELINE.json's north star names.
 * Outputs (S,S,K) sorted by ascending depth, -1 filled.  PARITY UNPINNED (see the header). */
void ref_rasterize_k2(const float *verts_ndc, const int32_t *faces, int F, int S, int K,
                      float blur_radius, int clip_bary, int cull_backfaces, int perspective_correct, int nthreads,
                      int32_t *pix_to_face, float *zbuf, float *bary, float *dists);

void ref_rasterize_k(const float *verts_ndc, const int32_t *faces, int F, int S, int K,
                     float blur_radius, int clip_bary, int nthreads,
                     int32_t *pix_to_face, float *zbuf, float *bary, float *dists)
{
    ref_rasterize_k2(verts_ndc, faces, F, S, K, blur_radius, clip_bary, 0, 1, nthreads, pix_to_face, zbuf, bary, dists);
}

/* ... with RasterizationSettings.cull_backfaces (a face whose signed NDC area is negative is skipped) and
 * perspective_correct (0: the screen-space barycentrics are used as they are) */
void ref_rasterize_k2(const float *verts_ndc, const int32_t *faces, int F, int S, int K,
                      float blur_radius, int clip_bary, int cull_backfaces, int perspective_correct, int nthreads,
                      int32_t *pix_to_face, float *zbuf, float *bary, float *dists)
{
    const float pad = sqrtf(blur_radius);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int yi = 0; yi < S; ++yi) {
        const float yf = pix_to_ndc(S - 1 - yi, S);
        for (int xi = 0; xi < S; ++xi) {
            const float xf = pix_to_ndc(S - 1 - xi, S);
            int qf[16]; float qz[16], qd[16], qb[16][3];
            int qn = 0;
            for (int f = 0; f < F; ++f) {
                const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
                const float x0 = verts_ndc[3 * i0], y0 = verts_ndc[3 * i0 + 1], z0 = verts_ndc[3 * i0 + 2];
                const float x1 = verts_ndc[3 * i1], y1 = verts_ndc[3 * i1 + 1], z1 = verts_ndc[3 * i1 + 2];
                const float x2 = verts_ndc[3 * i2], y2 = verts_ndc[3 * i2 + 1], z2 = verts_ndc[3 * i2 + 2];
                const float xmin = fminf(x0, fminf(x1, x2)) - pad, xmax = fmaxf(x0, fmaxf(x1, x2)) + pad;
                const float ymin = fminf(y0, fminf(y1, y2)) - pad, ymax = fmaxf(y0, fmaxf(y1, y2)) + pad;
                if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                if (fmaxf(z0, fmaxf(z1, z2)) < K_EPS) continue;
                const float face_area = edge_fn(x2, y2, x0, y0, x1, y1);
                if (face_area <= K_EPS && face_area >= -K_EPS) continue;
                if (cull_backfaces && face_area < 0.f) continue;
                const float area = face_area + K_EPS;
                const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                float b0 = w0, b1 = w1, b2 = w2;
                if (perspective_correct) {
                    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
                    const float den = fmaxf(t0 + t1 + t2, K_EPS);
                    b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
                }
                float c0 = b0, c1 = b1, c2 = b2;
                if (clip_bary) {
                    c0 = fminf(fmaxf(b0, 0.f), 1.f); c1 = fminf(fmaxf(b1, 0.f), 1.f); c2 = fminf(fmaxf(b2, 0.f), 1.f);
                    const float s = fmaxf(c0 + c1 + c2, K_EPS);
                    c0 /= s; c1 /= s; c2 /= s;
                }
                const float pz = c0 * z0 + c1 * z1 + c2 * z2;
                if (pz < 0.f) continue;
                const int inside = (b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f);
                const float d01 = point_line_dist2(xf, yf, x0, y0, x1, y1);
                const float d12 = point_line_dist2(xf, yf, x1, y1, x2, y2);
                const float d20 = point_line_dist2(xf, yf, x2, y2, x0, y0);
                const float d = fminf(d01, fminf(d12, d20));
                if (!inside && d >= blur_radius) continue;
                const float sd = inside ? -d : d;
                /* insert into the K-best list (ascending z; equal z keeps the earlier face first) */
                int pos = qn;
                while (pos > 0 && pz < qz[pos - 1]) --pos;
                if (pos >= K) continue;
                const int last = (qn < K) ? qn : K - 1;
                for (int j = last; j > pos; --j) {
                    qf[j] = qf[j - 1]; qz[j] = qz[j - 1]; qd[j] = qd[j - 1];
                    qb[j][0] = qb[j - 1][0]; qb[j][1] = qb[j - 1][1]; qb[j][2] = qb[j - 1][2];
                }
                qf[pos] = f; qz[pos] = pz; qd[pos] = sd; qb[pos][0] = c0; qb[pos][1] = c1; qb[pos][2] = c2;
                if (qn < K) ++qn;
            }
            const size_t p = ((size_t)yi * S + xi) * K;
            for (int k = 0; k < K; ++k) {
                if (k < qn) {
                    pix_to_face[p + k] = qf[k]; zbuf[p + k] = qz[k]; dists[p + k] = qd[k];
                    bary[3 * (p + k)] = qb[k][0]; bary[3 * (p + k) + 1] = qb[k][1]; bary[3 * (p + k) + 2] = qb[k][2];
                } else {
                    pix_to_face[p + k] = -1; zbuf[p + k] = -1.f; dists[p + k] = -1.f;
                    bary[3 * (p + k)] = -1.f; bary[3 * (p + k) + 1] = -1.f; bary[3 * (p + k) + 2] = -1.f;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------- near-plane clipping (PyTorch3D renderer/mesh/clip.py)
 * MeshRasterizer clips every mesh against z = z_clip_value (znear / 2 for perspective cameras) before rasterising:
 *   no vertex behind the plane (z < z_clip)  -> the face as it is
 *   all three behind                          -> removed
 *   ONE behind (p1)                           -> the quadrilateral p4 p2 p3 p5 as two triangles t1 = (p4, p2, p5), t2 = (p5, p2, p3)
 *   TWO behind (p1 = the one in front)        -> the triangle (p1, p4, p5)
 * with p2, p3 the face's other vertices in cyclic order after p1, p4 on edge p1-p2 and p5 on edge p1-p3 at depth z_clip:
 *   w2 = (z1 - z_clip) / (z1 - z2), w3 likewise; x/y of p4 interpolated in VIEW space when perspective_correct
 *   (x4 = ((1 - w2) x1 z1 + w2 x2 z2) / z_clip: NDC x/y are already divided by depth), linearly in NDC otherwise.
 * The rasteriser then works on the clipped triangles; two halves of one quadrilateral never both enter a pixel's K list
 * (the nearer-in-the-plane one, by unsigned edge distance, replaces the other), and the outputs are converted back:
 * pix_to_face = the ORIGINAL face, bary = bary_clipped . M with M the rows of barycentric coordinates of the clipped
 * triangle's vertices in the original face (p4 = (1 - w2) p1 + w2 p2, ...); zbuf and dists stay those of the clipped
 * triangle.  PARITY UNPINNED (PyTorch3D absent): restated from the published algorithm; analytic pins in
 * tests/test_oracle_soft.py.
 *
 * One face -> two record slots (2f, 2f + 1): tri[9] = x0 y0 z0 x1 y1 z1 x2 y2 z2, code 0 = empty, 1 = unclipped,
 * 2 + p1 + 3 * kind with kind 0 = t1, 1 = t2 (one vertex behind), 2 = the two-behind triangle; w[2] = (w2, w3). */
void ref_clip_face(const float v[9], float z_clip, int perspective_correct, float tri[2][9], int code[2], float w[2])
{
    code[0] = code[1] = 0; w[0] = w[1] = 0.f;
    const int behind[3] = {v[2] < z_clip, v[5] < z_clip, v[8] < z_clip};
    const int nb = behind[0] + behind[1] + behind[2];
    if (nb == 0) { memcpy(tri[0], v, 9 * sizeof(float)); code[0] = 1; return; }
    if (nb == 3) return;
    int i1 = 0;                                   /* the vertex alone on its side of the plane */
    for (int i = 0; i < 3; ++i) if (behind[i] == (nb == 1)) i1 = i;
    const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    const float *p1 = v + 3 * i1, *p2 = v + 3 * i2, *p3 = v + 3 * i3;
    const float w2 = (p1[2] - z_clip) / (p1[2] - p2[2]);
    const float w3 = (p1[2] - z_clip) / (p1[2] - p3[2]);
    float p4[3], p5[3];
    if (perspective_correct) {
        p4[0] = ((1.0f - w2) * (p1[0] * p1[2]) + w2 * (p2[0] * p2[2])) / z_clip;
        p4[1] = ((1.0f - w2) * (p1[1] * p1[2]) + w2 * (p2[1] * p2[2])) / z_clip;
        p5[0] = ((1.0f - w3) * (p1[0] * p1[2]) + w3 * (p3[0] * p3[2])) / z_clip;
        p5[1] = ((1.0f - w3) * (p1[1] * p1[2]) + w3 * (p3[1] * p3[2])) / z_clip;
    } else {
        p4[0] = (1.0f - w2) * p1[0] + w2 * p2[0]; p4[1] = (1.0f - w2) * p1[1] + w2 * p2[1];
        p5[0] = (1.0f - w3) * p1[0] + w3 * p3[0]; p5[1] = (1.0f - w3) * p1[1] + w3 * p3[1];
    }
    p4[2] = z_clip; p5[2] = z_clip;
    w[0] = w2; w[1] = w3;
    if (nb == 1) {
        memcpy(tri[0], p4, 12); memcpy(tri[0] + 3, p2, 12); memcpy(tri[0] + 6, p5, 12);
        memcpy(tri[1], p5, 12); memcpy(tri[1] + 3, p2, 12); memcpy(tri[1] + 6, p3, 12);
        code[0] = 2 + i1; code[1] = 2 + i1 + 3;
    } else {
        memcpy(tri[0], p1, 12); memcpy(tri[0] + 3, p4, 12); memcpy(tri[0] + 6, p5, 12);
        code[0] = 2 + i1 + 6;
    }
}

/* rows of M: barycentric coordinates (in the original face) of the clipped triangle's three vertices */
void ref_clip_conversion(int code, float w2, float w3, float M[3][3])
{
    memset(M, 0, 9 * sizeof(float));
    if (code <= 1) { M[0][0] = M[1][1] = M[2][2] = 1.f; return; }
    const int i1 = (code - 2) % 3, kind = (code - 2) / 3, i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    float b4[3] = {0, 0, 0}, b5[3] = {0, 0, 0}, b1[3] = {0, 0, 0}, b2[3] = {0, 0, 0}, b3[3] = {0, 0, 0};
    b4[i1] = 1.0f - w2; b4[i2] = w2; b5[i1] = 1.0f - w3; b5[i3] = w3; b1[i1] = 1.f; b2[i2] = 1.f; b3[i3] = 1.f;
    const float *rows[3];
    if (kind == 0) { rows[0] = b4; rows[1] = b2; rows[2] = b5; }
    else if (kind == 1) { rows[0] = b5; rows[1] = b2; rows[2] = b3; }
    else { rows[0] = b1; rows[1] = b4; rows[2] = b5; }
    for (int r = 0; r < 3; ++r) memcpy(M[r], rows[r], 12);
}

/* ref_rasterize_k2 + near-plane clipping at z_clip (< 0: off).  frag_slot (S,S,K) or NULL: the record slot 2 f + sub the
 * fragment came from (the HIP backward needs it; -1 = empty). */
void ref_rasterize_k3(const float *verts_ndc, const int32_t *faces, int F, int S, int K,
                      float blur_radius, int clip_bary, int cull_backfaces, int perspective_correct, float z_clip,
                      int nthreads, int32_t *pix_to_face, float *zbuf, float *bary, float *dists, int32_t *frag_slot)
{
    const int N = 2 * F;
    float *tri = (float *)malloc((size_t)N * 9 * sizeof(float));
    int *code = (int *)malloc((size_t)N * sizeof(int));
    float *ww = (float *)malloc((size_t)N * 2 * sizeof(float));
    for (int f = 0; f < F; ++f) {
        float v[9], t[2][9], w[2];
        int c[2];
        for (int k = 0; k < 3; ++k) memcpy(v + 3 * k, verts_ndc + 3 * faces[3 * f + k], 12);
        if (z_clip >= 0.f) ref_clip_face(v, z_clip, perspective_correct, t, c, w);
        else { memcpy(t[0], v, 36); c[0] = 1; c[1] = 0; w[0] = w[1] = 0.f; }
        for (int s = 0; s < 2; ++s) {
            memcpy(tri + (size_t)(2 * f + s) * 9, t[s], 36);
            code[2 * f + s] = c[s]; ww[2 * (2 * f + s)] = w[0]; ww[2 * (2 * f + s) + 1] = w[1];
        }
    }
    const float pad = sqrtf(blur_radius);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int yi = 0; yi < S; ++yi) {
        const float yf = pix_to_ndc(S - 1 - yi, S);
        for (int xi = 0; xi < S; ++xi) {
            const float xf = pix_to_ndc(S - 1 - xi, S);
            int qf[16]; float qz[16], qd[16], qb[16][3];
            int qn = 0;
            for (int sl = 0; sl < N; ++sl) {
                if (!code[sl]) continue;
                const float *t = tri + (size_t)sl * 9;
                const float x0 = t[0], y0 = t[1], z0 = t[2], x1 = t[3], y1 = t[4], z1 = t[5], x2 = t[6], y2 = t[7], z2 = t[8];
                const float xmin = fminf(x0, fminf(x1, x2)) - pad, xmax = fmaxf(x0, fmaxf(x1, x2)) + pad;
                const float ymin = fminf(y0, fminf(y1, y2)) - pad, ymax = fmaxf(y0, fmaxf(y1, y2)) + pad;
                if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                if (fmaxf(z0, fmaxf(z1, z2)) < K_EPS) continue;
                const float face_area = edge_fn(x2, y2, x0, y0, x1, y1);
                if (face_area <= K_EPS && face_area >= -K_EPS) continue;
                if (cull_backfaces && face_area < 0.f) continue;
                const float area = face_area + K_EPS;
                const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                float b0 = w0, b1 = w1, b2 = w2;
                if (perspective_correct) {
                    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
                    const float den = fmaxf(t0 + t1 + t2, K_EPS);
                    b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
                }
                float c0 = b0, c1 = b1, c2 = b2;
                if (clip_bary) {
                    c0 = fminf(fmaxf(b0, 0.f), 1.f); c1 = fminf(fmaxf(b1, 0.f), 1.f); c2 = fminf(fmaxf(b2, 0.f), 1.f);
                    const float s = fmaxf(c0 + c1 + c2, K_EPS);
                    c0 /= s; c1 /= s; c2 /= s;
                }
                const float pz = c0 * z0 + c1 * z1 + c2 * z2;
                if (pz < 0.f) continue;
                const int inside = (b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f);
                const float d01 = point_line_dist2(xf, yf, x0, y0, x1, y1);
                const float d12 = point_line_dist2(xf, yf, x1, y1, x2, y2);
                const float d20 = point_line_dist2(xf, yf, x2, y2, x0, y0);
                const float d = fminf(d01, fminf(d12, d20));
                if (!inside && d >= blur_radius) continue;
                const float sd = inside ? -d : d;
                /* the other half of a split quadrilateral already in the list?  keep the one nearer in the image plane */
                const int kind = code[sl] >= 2 ? (code[sl] - 2) / 3 : -1;
                if (kind == 0 || kind == 1) {
                    const int other = sl ^ 1;
                    int at = -1;
                    for (int j = 0; j < qn; ++j) if (qf[j] == other) at = j;
                    if (at >= 0) {
                        if (!(d < fabsf(qd[at]))) continue;
                        for (int j = at; j + 1 < qn; ++j) {            /* drop it; the new one is inserted below */
                            qf[j] = qf[j + 1]; qz[j] = qz[j + 1]; qd[j] = qd[j + 1];
                            qb[j][0] = qb[j + 1][0]; qb[j][1] = qb[j + 1][1]; qb[j][2] = qb[j + 1][2];
                        }
                        --qn;
                    }
                }
                int pos = qn;
                while (pos > 0 && pz < qz[pos - 1]) --pos;
                if (pos >= K) continue;
                const int last = (qn < K) ? qn : K - 1;
                for (int j = last; j > pos; --j) {
                    qf[j] = qf[j - 1]; qz[j] = qz[j - 1]; qd[j] = qd[j - 1];
                    qb[j][0] = qb[j - 1][0]; qb[j][1] = qb[j - 1][1]; qb[j][2] = qb[j - 1][2];
                }
                qf[pos] = sl; qz[pos] = pz; qd[pos] = sd; qb[pos][0] = c0; qb[pos][1] = c1; qb[pos][2] = c2;
                if (qn < K) ++qn;
            }
            const size_t p = ((size_t)yi * S + xi) * K;
            for (int k = 0; k < K; ++k) {
                if (k < qn) {
                    const int sl = qf[k];
                    float M[3][3];
                    ref_clip_conversion(code[sl], ww[2 * sl], ww[2 * sl + 1], M);
                    pix_to_face[p + k] = sl >> 1; zbuf[p + k] = qz[k]; dists[p + k] = qd[k];
                    for (int i = 0; i < 3; ++i)
                        bary[3 * (p + k) + i] = code[sl] <= 1 ? qb[k][i]
                                                              : (qb[k][0] * M[0][i] + qb[k][1] * M[1][i]) + qb[k][2] * M[2][i];
                    if (frag_slot) frag_slot[p + k] = sl;
                } else {
                    pix_to_face[p + k] = -1; zbuf[p + k] = -1.f; dists[p + k] = -1.f;
                    bary[3 * (p + k)] = -1.f; bary[3 * (p + k) + 1] = -1.f; bary[3 * (p + k) + 2] = -1.f;
                    if (frag_slot) frag_slot[p + k] = -1;
                }
            }
        }
    }
    free(tri); free(code); free(ww);
}
