// soft.hip -- the GENERAL soft renderer: K = faces_per_pixel nearest faces per pixel, blur_radius
// >= 0 (faces contribute up to sqrt(blur) outside their edges), barycentric clipping, and
// softmax_rgb_blend over the K layers (sigma, gamma, background), forward and backward
// (texture, barycentric, depth and edge-distance gradients -> vertices).
// The reference fixes K = 1 / blur = 0 (first_approach.py:107, second_approach.py:101) and runs on
// the specialised kernels of raster.hip / shade.hip; this file is the general form named by the
// north star ("soft rasteriser ... soft-aggregation z-blend").  Same structure:
// one 256-thread workgroup per 16x16 tile, order-preserving ballot compaction of the faces whose
// padded bbox touches the tile into an LDS list, every lane (= pixel) walks the list keeping its K
// nearest candidates in registers (insertion into a depth-sorted list, K is a template parameter
// so the list never leaves the VGPRs).  HBM-bound: S*S*K*24 B of fragments per view.
// Built with -ffp-contract=off: same operation order as oracle/raster_ref.c:ref_rasterize_k.
#include <type_traits>

#include "common.h"
#include "det.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr int TILE = 16;
constexpr int LIST_CAP = 512;
constexpr float kBlendEps = 1e-10f, kZnear = 1.0f, kZfar = 100.0f;

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }
__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float pld2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= kEps) {
        const float dx = px - bx, dy = py - by;
        return dx * dx + dy * dy;
    }
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

// ------------------------------------------------------------------------------------------ near-plane clipping
// PyTorch3D clips every mesh against z = z_clip_value (znear / 2 for perspective cameras) before rasterising
// (renderer/mesh/clip.py; oracle/raster_ref.c:ref_clip_face is the CPU restatement, same operation order): a face with one
// vertex behind the plane becomes the quadrilateral p4 p2 p3 p5 = triangles t1 (p4, p2, p5) and t2 (p5, p2, p3), with two
// behind the triangle (p1, p4, p5), with three behind nothing.  Two record slots per face: 2 f (the face itself, t1, or the
// two-behind triangle) and 2 f + 1 (t2).  code: 0 empty, 1 unclipped, 2 + p1 + 3 * kind (kind 0 = t1, 1 = t2, 2 = two behind).
struct Clipped { float t[2][9]; int code[2]; float w2, w3; };

__device__ __forceinline__ void clip_face_dev(const float v[9], float z_clip, int persp, Clipped &o) {
    o.code[0] = o.code[1] = 0; o.w2 = o.w3 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { o.t[0][k] = 0.f; o.t[1][k] = 0.f; }
    const bool b0 = v[2] < z_clip, b1 = v[5] < z_clip, b2 = v[8] < z_clip;
    const int nb = (int)b0 + (int)b1 + (int)b2;
    if (nb == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) o.t[0][k] = v[k];
        o.code[0] = 1;
        return;
    }
    if (nb == 3) return;
    const bool lone = (nb == 1);                   // the vertex alone on its side of the plane
    const int i1 = (b0 == lone) ? 0 : ((b1 == lone) ? 1 : 2);      // (the host restatement keeps the LAST match; one match exists)
    const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    float p1[3], p2[3], p3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p1[k] = i1 == 0 ? v[k] : (i1 == 1 ? v[3 + k] : v[6 + k]);
        p2[k] = i2 == 0 ? v[k] : (i2 == 1 ? v[3 + k] : v[6 + k]);
        p3[k] = i3 == 0 ? v[k] : (i3 == 1 ? v[3 + k] : v[6 + k]);
    }
    const float w2 = (p1[2] - z_clip) / (p1[2] - p2[2]);
    const float w3 = (p1[2] - z_clip) / (p1[2] - p3[2]);
    float p4[3], p5[3];
    if (persp) {
        p4[0] = ((1.0f - w2) * (p1[0] * p1[2]) + w2 * (p2[0] * p2[2])) / z_clip;
        p4[1] = ((1.0f - w2) * (p1[1] * p1[2]) + w2 * (p2[1] * p2[2])) / z_clip;
        p5[0] = ((1.0f - w3) * (p1[0] * p1[2]) + w3 * (p3[0] * p3[2])) / z_clip;
        p5[1] = ((1.0f - w3) * (p1[1] * p1[2]) + w3 * (p3[1] * p3[2])) / z_clip;
    } else {
        p4[0] = (1.0f - w2) * p1[0] + w2 * p2[0]; p4[1] = (1.0f - w2) * p1[1] + w2 * p2[1];
        p5[0] = (1.0f - w3) * p1[0] + w3 * p3[0]; p5[1] = (1.0f - w3) * p1[1] + w3 * p3[1];
    }
    p4[2] = z_clip; p5[2] = z_clip;
    o.w2 = w2; o.w3 = w3;
    if (nb == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o.t[0][k] = p4[k]; o.t[0][3 + k] = p2[k]; o.t[0][6 + k] = p5[k];
            o.t[1][k] = p5[k]; o.t[1][3 + k] = p2[k]; o.t[1][6 + k] = p3[k];
        }
        o.code[0] = 2 + i1; o.code[1] = 2 + i1 + 3;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { o.t[0][k] = p1[k]; o.t[0][3 + k] = p4[k]; o.t[0][6 + k] = p5[k]; }
        o.code[0] = 2 + i1 + 6;
    }
}

// out = c . M, M = rows of barycentric coordinates (in the original face) of the clipped triangle's vertices:
//   t1 (kind 0) rows (b4, b2, b5); t2 (kind 1) rows (b5, b2, b3); two-behind (kind 2) rows (b1, b4, b5)
//   b4 = (1 - w2) e_i1 + w2 e_i2, b5 = (1 - w3) e_i1 + w3 e_i3.  Written out per original vertex in the summation order
//   (row0 + row1) + row2 of the CPU restatement (products with the matrix's zeros and ones are exact and omitted).
__device__ __forceinline__ void clip_convert(int code, float w2, float w3, float c0, float c1, float c2, float out[3]) {
    if (code <= 1) { out[0] = c0; out[1] = c1; out[2] = c2; return; }
    const int i1 = (code - 2) % 3, kind = (code - 2) / 3, i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    float e1, e2, e3;
    if (kind == 0) { e1 = c0 * (1.0f - w2) + c2 * (1.0f - w3); e2 = c0 * w2 + c1; e3 = c2 * w3; }
    else if (kind == 1) { e1 = c0 * (1.0f - w3); e2 = c1; e3 = c0 * w3 + c2; }
    else { e1 = (c0 + c1 * (1.0f - w2)) + c2 * (1.0f - w3); e2 = c1 * w2; e3 = c2 * w3; }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (k == i1) ? e1 : ((k == i2) ? e2 : e3);
}

// record: [x0 y0 z0 x1][y1 z1 x2 y2][z2 code w2 w3]; two per face
__global__ void face_setup_clip_kernel(const float *__restrict__ ndc, const int32_t *__restrict__ faces, int B, int V, int F,
                                       float z_clip, int persp, float4 *__restrict__ rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * F) return;
    const int b = i / F, f = i - b * F;
    const float *vb = ndc + (size_t)b * V * 3;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float v[9] = {vb[3 * i0], vb[3 * i0 + 1], vb[3 * i0 + 2], vb[3 * i1], vb[3 * i1 + 1], vb[3 * i1 + 2],
                        vb[3 * i2], vb[3 * i2 + 1], vb[3 * i2 + 2]};
    Clipped c;
    clip_face_dev(v, z_clip, persp, c);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float *t = c.t[s];
        const float zmax = fmaxf(t[2], fmaxf(t[5], t[8]));
        const float area = edge_fn(t[6], t[7], t[0], t[1], t[3], t[4]);
        const bool valid = c.code[s] != 0 && !(zmax < kEps) && !(area <= kEps && area >= -kEps);
        const size_t o = ((size_t)b * F + f) * 2 + s;
        rec[3 * o + 0] = make_float4(t[0], t[1], t[2], t[3]);
        rec[3 * o + 1] = make_float4(t[4], t[5], t[6], t[7]);
        rec[3 * o + 2] = make_float4(t[8], valid ? (float)c.code[s] : 0.f, c.w2, c.w3);
    }
}

// face records as in raster.hip: [x0 y0 z0 x1][y1 z1 x2 y2][z2 valid 0 0]; with clipping two per face (rpf = 2), see above
template <int K>
__global__ __launch_bounds__(256) void raster_k_kernel(const float4 *__restrict__ rec, int F, int S, float blur, int clip,
                                                       int cull, int persp, int rpf, int32_t *__restrict__ pix_to_face,
                                                       float *__restrict__ zbuf, float *__restrict__ bary,
                                                       float *__restrict__ dists, int32_t *__restrict__ frag_slot) {
    // F counts RECORDS here (rpf per face: 2 with near-plane clipping, else 1)
    __shared__ float s_face[LIST_CAP][10];
    __shared__ int s_fidx[LIST_CAP];
    __shared__ int s_wcnt[4];

    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = blockIdx.x * TILE + (tid & (TILE - 1));
    const int py = blockIdx.y * TILE + (tid >> 4);
    const bool in_img = px < S && py < S;
    const float xf = pix_to_ndc(S - 1 - px, S), yf = pix_to_ndc(S - 1 - py, S);
    const float pad = sqrtf(blur);
    const int px_hi = min(blockIdx.x * TILE + TILE - 1, S - 1), py_hi = min(blockIdx.y * TILE + TILE - 1, S - 1);
    const float tx_max = pix_to_ndc(S - 1 - blockIdx.x * TILE, S), tx_min = pix_to_ndc(S - 1 - px_hi, S);
    const float ty_max = pix_to_ndc(S - 1 - blockIdx.y * TILE, S), ty_min = pix_to_ndc(S - 1 - py_hi, S);
    const float4 *rb = rec + (size_t)b * F * 3;

    int qf[K]; float qz[K], qd[K], qb0[K], qb1[K], qb2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { qf[k] = -1; qz[k] = 3.0e38f; qd[k] = -1.f; qb0[k] = qb1[k] = qb2[k] = -1.f; }

    int count = 0;
    for (int base = 0; base < F; base += 256) {
        const int f = base + tid;
        bool hit = false;
        float4 r0, r1, r2;
        if (f < F) {
            r0 = rb[3 * (size_t)f]; r1 = rb[3 * (size_t)f + 1]; r2 = rb[3 * (size_t)f + 2];
            const float xmin = fminf(r0.x, fminf(r0.w, r1.z)) - pad, xmax = fmaxf(r0.x, fmaxf(r0.w, r1.z)) + pad;
            const float ymin = fminf(r0.y, fminf(r1.x, r1.w)) - pad, ymax = fmaxf(r0.y, fmaxf(r1.x, r1.w)) + pad;
            hit = (r2.y != 0.f) && !(tx_min > xmax || tx_max < xmin || ty_min > ymax || ty_max < ymin);
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = count;
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        if (hit) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_fidx[slot] = f;
            s_face[slot][0] = r0.x; s_face[slot][1] = r0.y; s_face[slot][2] = r0.z;
            s_face[slot][3] = r0.w; s_face[slot][4] = r1.x; s_face[slot][5] = r1.y;
            s_face[slot][6] = r1.z; s_face[slot][7] = r1.w; s_face[slot][8] = r2.x;
            s_face[slot][9] = r2.y;                     // record code (clipping: which kind of sub-triangle)
        }
        count += total;
        __syncthreads();
        if (count > LIST_CAP - 256 || base + 256 >= F) {
            for (int i = 0; i < count; ++i) {
                const float x0 = s_face[i][0], y0 = s_face[i][1], z0 = s_face[i][2];
                const float x1 = s_face[i][3], y1 = s_face[i][4], z1 = s_face[i][5];
                const float x2 = s_face[i][6], y2 = s_face[i][7], z2 = s_face[i][8];
                const float xmin = fminf(x0, fminf(x1, x2)) - pad, xmax = fmaxf(x0, fmaxf(x1, x2)) + pad;
                const float ymin = fminf(y0, fminf(y1, y2)) - pad, ymax = fmaxf(y0, fmaxf(y1, y2)) + pad;
                if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                const float face_area = edge_fn(x2, y2, x0, y0, x1, y1);
                if (cull && face_area < 0.f) continue;      // cull_backfaces: the face winds away from the camera
                const float area = face_area + kEps;
                const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                float b0 = w0, b1 = w1, b2 = w2;            // perspective_correct = False: screen-space barycentrics
                if (persp) {
                    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
                    const float den = fmaxf(t0 + t1 + t2, kEps);
                    b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
                }
                float c0 = b0, c1 = b1, c2 = b2;
                if (clip) {
                    c0 = fminf(fmaxf(b0, 0.f), 1.f); c1 = fminf(fmaxf(b1, 0.f), 1.f); c2 = fminf(fmaxf(b2, 0.f), 1.f);
                    const float s = fmaxf(c0 + c1 + c2, kEps);
                    c0 /= s; c1 /= s; c2 /= s;
                }
                const float pz = c0 * z0 + c1 * z1 + c2 * z2;
                if (pz < 0.f) continue;
                const bool inside = (b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f);
                const float d = fminf(pld2(xf, yf, x0, y0, x1, y1), fminf(pld2(xf, yf, x1, y1, x2, y2), pld2(xf, yf, x2, y2, x0, y0)));
                if (!inside && d >= blur) continue;
                if (rpf == 2) {
                    // the other half of a split quadrilateral already in the list?  keep the one nearer in the image plane
                    const int code = (int)s_face[i][9];
                    if (code >= 2 && code < 8) {
                        const int other = s_fidx[i] ^ 1;
                        int at = -1;
#pragma unroll
                        for (int k = 0; k < K; ++k) if (qf[k] == other) at = k;
                        if (at >= 0) {
                            float od = 0.f;
#pragma unroll
                            for (int k = 0; k < K; ++k) if (k == at) od = qd[k];
                            if (!(d < fabsf(od))) continue;
#pragma unroll
                            for (int k = 0; k + 1 < K; ++k)
                                if (k >= at) { qf[k] = qf[k + 1]; qz[k] = qz[k + 1]; qd[k] = qd[k + 1]; qb0[k] = qb0[k + 1]; qb1[k] = qb1[k + 1]; qb2[k] = qb2[k + 1]; }
                            qf[K - 1] = -1; qz[K - 1] = 3.0e38f; qd[K - 1] = -1.f; qb0[K - 1] = qb1[K - 1] = qb2[K - 1] = -1.f;
                        }
                    }
                }
                if (!(pz < qz[K - 1])) continue;          // not among the K nearest (ties keep the earlier face)
                // sorted insertion, fully unrolled: the list stays in registers
                // (once the new fragment is placed every later slot shifts unconditionally, so equal-z entries
                // already in the list keep their order)
                int cf = s_fidx[i]; float cz = pz, cd = inside ? -d : d, e0 = c0, e1 = c1, e2 = c2;
                bool placed = false;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (placed || cz < qz[k]) {
                        placed = true;
                        const int tf = qf[k]; const float tz = qz[k], td = qd[k], u0 = qb0[k], u1 = qb1[k], u2 = qb2[k];
                        qf[k] = cf; qz[k] = cz; qd[k] = cd; qb0[k] = e0; qb1[k] = e1; qb2[k] = e2;
                        cf = tf; cz = tz; cd = td; e0 = u0; e1 = u1; e2 = u2;
                    }
                }
            }
            count = 0;
            __syncthreads();
        }
    }
    if (!in_img) return;
    const size_t p = (((size_t)b * S + py) * S + px) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = qf[k] >= 0;
        zbuf[p + k] = ok ? qz[k] : -1.f;
        dists[p + k] = qd[k];
        if (rpf == 2) {                     // back to the original face: index, and barycentrics through the clip's matrix
            float o3[3] = {qb0[k], qb1[k], qb2[k]};
            if (ok) {
                const float4 r2 = rb[3 * (size_t)qf[k] + 2];
                clip_convert((int)r2.y, r2.z, r2.w, qb0[k], qb1[k], qb2[k], o3);
            }
            pix_to_face[p + k] = ok ? (qf[k] >> 1) : -1;
            bary[3 * (p + k)] = o3[0]; bary[3 * (p + k) + 1] = o3[1]; bary[3 * (p + k) + 2] = o3[2];
        } else {
            pix_to_face[p + k] = qf[k];
            bary[3 * (p + k)] = qb0[k]; bary[3 * (p + k) + 1] = qb1[k]; bary[3 * (p + k) + 2] = qb2[k];
        }
        if (frag_slot) frag_slot[p + k] = qf[k];
    }
}

// ------------------------------------------------------------------------------------------ shading
struct Footprint { int x0, x1, r0, r1; float wx0, wx1, wy0, wy1; bool vx0, vx1, vy0, vy1, cx, cy; };
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

struct SoftArgs {
    const int32_t *p2f; const float *bary, *zbuf, *dists, *uvs; const int32_t *fuv; const float *tex;
    int B, S, T, K; float sigma, gamma, bg0, bg1, bg2;
};

// one thread per pixel, loop over the K layers twice (z_max first).  MODE 0: forward (rgb, alpha), threads along rows;
// MODE 1: backward (texture scatter + per-layer d/d bary, d/d zbuf, d/d dists), one workgroup per 16x16-pixel tile:
// neighbouring pixels (and layers) share texels, so the contributions are first summed per texel in an LDS table (open
// addressing on the texel index) and each distinct texel of the tile then costs three global atomics -- what
// shade_bwd_kernel does for the specialised K = 1 path.  Round 2 issued one global float atomic per contribution: with a
// mesh that fills the screen (config 5 after the vertices reach the camera) that was 12 ms per step.  DET 1: the same in
// 64-bit fixed point (det.h; gtex is then the int64 accumulator array): bitwise reproducible.  A contribution that finds
// no slot within kSoftProbe steps (K layers can touch more texels than the table holds) goes to global memory directly.
constexpr int kSoftTexSlots = 2048, kSoftProbe = 24;

template <int MODE, int DET = 0>
__global__ __launch_bounds__(256) void soft_shade_kernel(const SoftArgs a, float *__restrict__ rgb, float *__restrict__ alpha_out,
                                                         const float *__restrict__ grad_rgb, float *__restrict__ gtex,
                                                         float *__restrict__ gbary, float *__restrict__ gz, float *__restrict__ gd,
                                                         int tiles_x, const st3d_det::DetHeader *__restrict__ det) {
    typedef typename std::conditional<DET != 0, unsigned long long, float>::type acc_t;
    __shared__ int s_key[MODE == 1 ? kSoftTexSlots : 1];
    __shared__ acc_t s_acc[MODE == 1 ? kSoftTexSlots : 1][3];
    const size_t HW = (size_t)a.S * a.S;
    size_t i, b, p;
    bool live;
    const bool binned = MODE == 1 && gtex != nullptr;
    if (MODE == 1) {
        const int tid = threadIdx.x;
        if (binned) {
            for (int e = tid; e < kSoftTexSlots; e += 256) s_key[e] = -1;
            for (int e = tid; e < kSoftTexSlots * 3; e += 256) (&s_acc[0][0])[e] = (acc_t)0;
            __syncthreads();
        }
        const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
        const int yi = ty * 16 + (tid >> 4), xi = tx * 16 + (tid & 15);
        live = yi < a.S && xi < a.S;
        b = blockIdx.y; p = (size_t)yi * a.S + xi; i = b * HW + p;
    } else {
        i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= (size_t)a.B * HW) return;
        b = i / HW; p = i - b * HW; live = true;
    }
    const double dscale = (MODE == 1 && DET) ? det->scale : 1.0;
    // texel (T x T index) += v[0..2] * w
    auto deposit = [&](int texel, const float *v, float w) __attribute__((always_inline)) {
        int slot = (int)(((unsigned)texel * 2654435761u) >> 21) & (kSoftTexSlots - 1);
        bool found = false;
        for (int tries = 0; tries < kSoftProbe; ++tries) {
            const int prev = atomicCAS(&s_key[slot], -1, texel);
            if (prev == -1 || prev == texel) { found = true; break; }
            slot = (slot + 1) & (kSoftTexSlots - 1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (DET) {
                const unsigned long long q = (unsigned long long)st3d_det::det_quantise(v[c] * w, dscale);
                if (found) atomicAdd(reinterpret_cast<unsigned long long *>(&s_acc[slot][c]), q);
                else atomicAdd(reinterpret_cast<unsigned long long *>(gtex) + (size_t)texel * 3 + c, q);
            } else {
                if (found) atomicAdd(reinterpret_cast<float *>(&s_acc[slot][c]), v[c] * w);
                else atomicAdd(gtex + (size_t)texel * 3 + c, v[c] * w);
            }
        }
    };
    if (live) {
    const int K = a.K, T = a.T;
    const float zr = 1.0f / (kZfar - kZnear);
    // pass 1: z_max (argmax = first maximum, as torch.max) and alpha
    float z_max = 0.f; int kmax = -1; float keep = 1.f;
    for (int k = 0; k < K; ++k) {
        const int f = a.p2f[i * K + k];
        if (f < 0) continue;
        const float z_inv = (kZfar - a.zbuf[i * K + k]) * zr;
        if (kmax < 0 || z_inv > z_max) { if (kmax < 0 || z_inv > z_max) { z_max = z_inv; kmax = k; } }
        const float prob = 1.0f / (1.0f + expf(a.dists[i * K + k] / a.sigma));
        keep *= (1.0f - prob);
    }
    if (kmax < 0) z_max = 0.f;                     // all masked: z_inv * mask = 0
    const bool zclamped = !(z_max > kBlendEps);
    if (zclamped) z_max = kBlendEps;
    const float dexp = expf((kBlendEps - z_max) / a.gamma);
    const bool dclamped = !(dexp > kBlendEps);
    const float delta = dclamped ? kBlendEps : dexp;
    // pass 2: weights and colours
    float wsum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const int f = a.p2f[i * K + k];
        if (f < 0) continue;
        const float prob = 1.0f / (1.0f + expf(a.dists[i * K + k] / a.sigma));
        const float z_inv = (kZfar - a.zbuf[i * K + k]) * zr;
        const float w = prob * expf((z_inv - z_max) / a.gamma);
        const float b0 = a.bary[3 * (i * K + k)], b1 = a.bary[3 * (i * K + k) + 1], b2 = a.bary[3 * (i * K + k) + 2];
        const int u0 = a.fuv[3 * f], u1 = a.fuv[3 * f + 1], u2 = a.fuv[3 * f + 2];
        const float u = b0 * a.uvs[2 * u0] + b1 * a.uvs[2 * u1] + b2 * a.uvs[2 * u2];
        const float v = b0 * a.uvs[2 * u0 + 1] + b1 * a.uvs[2 * u1 + 1] + b2 * a.uvs[2 * u2 + 1];
        const Footprint q = uv_footprint(u, v, T);
        const float w00 = q.wx0 * q.wy0, w01 = q.wx1 * q.wy0, w10 = q.wx0 * q.wy1, w11 = q.wx1 * q.wy1;
        const float *t00 = a.tex + ((size_t)q.r0 * T + q.x0) * 3, *t01 = a.tex + ((size_t)q.r0 * T + q.x1) * 3;
        const float *t10 = a.tex + ((size_t)q.r1 * T + q.x0) * 3, *t11 = a.tex + ((size_t)q.r1 * T + q.x1) * 3;
        float t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            if (q.vy0 && q.vx0) s += t00[c] * w00;
            if (q.vy0 && q.vx1) s += t01[c] * w01;
            if (q.vy1 && q.vx0) s += t10[c] * w10;
            if (q.vy1 && q.vx1) s += t11[c] * w11;
            t[c] = s;
        }
        wsum += w; c0 += w * t[0]; c1 += w * t[1]; c2 += w * t[2];
    }
    const float denom = wsum + delta;
    const float r0 = (c0 + delta * a.bg0) / denom, r1 = (c1 + delta * a.bg1) / denom, r2 = (c2 + delta * a.bg2) / denom;
    if (MODE == 0) {
        float *o = rgb + b * 3 * HW + p;
        o[0] = r0; o[HW] = r1; o[2 * HW] = r2;
        alpha_out[i] = 1.0f - keep;
        return;
    }
    // ---- backward
    const float *g = grad_rgb + b * 3 * HW + p;
    const float g0 = g[0], g1 = g[HW], g2 = g[2 * HW];
    const float ddelta = (g0 * (a.bg0 - r0) + g1 * (a.bg1 - r1) + g2 * (a.bg2 - r2)) / denom;
    float dzmax = dclamped ? 0.f : ddelta * (-delta / a.gamma);
    // pass 3a: d w_k -> d z_max contributions
    for (int k = 0; k < K; ++k) {
        const int f = a.p2f[i * K + k];
        if (gbary) { gbary[3 * (i * K + k)] = 0.f; gbary[3 * (i * K + k) + 1] = 0.f; gbary[3 * (i * K + k) + 2] = 0.f; }
        if (gz) gz[i * K + k] = 0.f;
        if (gd) gd[i * K + k] = 0.f;
        if (f < 0) continue;
    }
    // pass 3b: per-layer gradients (recompute the layer, then scatter)
    float dz_acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int f = a.p2f[i * K + k];
        if (f < 0) continue;
        const float prob = 1.0f / (1.0f + expf(a.dists[i * K + k] / a.sigma));
        const float z_inv = (kZfar - a.zbuf[i * K + k]) * zr;
        const float e = expf((z_inv - z_max) / a.gamma);
        const float w = prob * e;
        const float b0 = a.bary[3 * (i * K + k)], b1 = a.bary[3 * (i * K + k) + 1], b2 = a.bary[3 * (i * K + k) + 2];
        const int u0 = a.fuv[3 * f], u1 = a.fuv[3 * f + 1], u2 = a.fuv[3 * f + 2];
        const float u = b0 * a.uvs[2 * u0] + b1 * a.uvs[2 * u1] + b2 * a.uvs[2 * u2];
        const float v = b0 * a.uvs[2 * u0 + 1] + b1 * a.uvs[2 * u1 + 1] + b2 * a.uvs[2 * u2 + 1];
        const Footprint q = uv_footprint(u, v, T);
        const float w00 = q.wx0 * q.wy0, w01 = q.wx1 * q.wy0, w10 = q.wx0 * q.wy1, w11 = q.wx1 * q.wy1;
        const size_t o00 = ((size_t)q.r0 * T + q.x0) * 3, o01 = ((size_t)q.r0 * T + q.x1) * 3;
        const size_t o10 = ((size_t)q.r1 * T + q.x0) * 3, o11 = ((size_t)q.r1 * T + q.x1) * 3;
        const float kw = w / denom;           // d rgb / d colour_k
        const float gc[3] = {g0, g1, g2};
        const float rr[3] = {r0, r1, r2};
        float gix = 0.f, giy = 0.f, dw = 0.f;
        if (binned) {
            const float gk3[3] = {g0 * kw, g1 * kw, g2 * kw};
            if (q.vy0 && q.vx0) deposit((int)(o00 / 3), gk3, w00);
            if (q.vy0 && q.vx1) deposit((int)(o01 / 3), gk3, w01);
            if (q.vy1 && q.vx0) deposit((int)(o10 / 3), gk3, w10);
            if (q.vy1 && q.vx1) deposit((int)(o11 / 3), gk3, w11);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
            const float gck = gc[c] * kw;
            if (q.vy0 && q.vx0) t00 = a.tex[o00 + c];
            if (q.vy0 && q.vx1) t01 = a.tex[o01 + c];
            if (q.vy1 && q.vx0) t10 = a.tex[o10 + c];
            if (q.vy1 && q.vx1) t11 = a.tex[o11 + c];
            const float tc = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;
            dw += gc[c] * (tc - rr[c]) / denom;
            gix += gck * ((t01 - t00) * q.wy0 + (t11 - t10) * q.wy1);
            giy += gck * ((t10 - t00) * q.wx0 + (t11 - t01) * q.wx1);
        }
        if (gbary) {
            const float gu = q.cx ? 0.f : gix * (float)(T - 1), gv = q.cy ? 0.f : giy * (float)(T - 1);
            gbary[3 * (i * K + k)] = gu * a.uvs[2 * u0] + gv * a.uvs[2 * u0 + 1];
            gbary[3 * (i * K + k) + 1] = gu * a.uvs[2 * u1] + gv * a.uvs[2 * u1 + 1];
            gbary[3 * (i * K + k) + 2] = gu * a.uvs[2 * u2] + gv * a.uvs[2 * u2 + 1];
        }
        // w = prob * exp((z_inv - z_max)/gamma)
        const float dprob = dw * e;
        const float dzinv = dw * w / a.gamma;
        dzmax -= dzinv;
        if (gd) gd[i * K + k] = dprob * prob * (1.0f - prob) * (-1.0f / a.sigma);
        if (gz) gz[i * K + k] = -dzinv * zr;          // z_inv = (zfar - z)/(zfar - znear)
        (void)dz_acc;
    }
    // z_max = max_k z_inv_k (clamped): its gradient goes to the arg-max layer
    if (gz && kmax >= 0 && !zclamped) gz[i * K + kmax] += -dzmax * zr;
    }   // live
    if (binned) {
        __syncthreads();
        for (int e = threadIdx.x; e < kSoftTexSlots * 3; e += 256) {
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

// ------------------------------------------------------------------------------------------ raster backward
// d loss / d (clipped barycentrics, depth, signed distance) per (pixel, layer) -> projected vertices
// the nine contributions of fragment i (= (pixel, layer)) of face f to the face's three projected vertices, in the face's
// own vertex order: c9 = {d/dx0, d/dy0, d/dz0, d/dx1, ...}
__device__ __forceinline__ void raster_k_frag_grad(const float *__restrict__ gbary, const float *__restrict__ gzb,
                                                   const float *__restrict__ gdist, const float *__restrict__ vb,
                                                   const int32_t *__restrict__ faces, int S, int clip, int persp,
                                                   const int32_t *__restrict__ frag_slot, float z_clip, size_t i, int f, int xi,
                                                   int yi, float c9[9]) {
    const float px = pix_to_ndc(S - 1 - xi, S), py = pix_to_ndc(S - 1 - yi, S);
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    float x0 = vb[3 * i0], y0 = vb[3 * i0 + 1], z0 = vb[3 * i0 + 2];
    float x1 = vb[3 * i1], y1 = vb[3 * i1 + 1], z1 = vb[3 * i1 + 2];
    float x2 = vb[3 * i2], y2 = vb[3 * i2 + 1], z2 = vb[3 * i2 + 2];
    // near-plane clipping: the fragment lives on a clipped sub-triangle of the face; redo the clip (same arithmetic as
    // face_setup_clip_kernel), differentiate on the sub-triangle, then carry the gradient back to the face's vertices
    const float ov[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
    int ccode = 1; float cw2 = 0.f, cw3 = 0.f;
    if (frag_slot) {
        Clipped cl;
        clip_face_dev(ov, z_clip, persp, cl);
        const int sub = frag_slot[i] & 1;
        ccode = cl.code[sub]; cw2 = cl.w2; cw3 = cl.w3;
        x0 = cl.t[sub][0]; y0 = cl.t[sub][1]; z0 = cl.t[sub][2]; x1 = cl.t[sub][3]; y1 = cl.t[sub][4]; z1 = cl.t[sub][5];
        x2 = cl.t[sub][6]; y2 = cl.t[sub][7]; z2 = cl.t[sub][8];
    }
    const float A = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
    const float w0 = edge_fn(px, py, x1, y1, x2, y2) / A;
    const float w1 = edge_fn(px, py, x2, y2, x0, y0) / A;
    const float w2 = edge_fn(px, py, x0, y0, x1, y1) / A;
    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
    const float den = t0 + t1 + t2;
    float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f, dz0 = 0.f, dz1 = 0.f, dz2 = 0.f;
    const float b0 = persp ? t0 / fmaxf(den, kEps) : w0, b1 = persp ? t1 / fmaxf(den, kEps) : w1,
                b2 = persp ? t2 / fmaxf(den, kEps) : w2;
    // ---- clipped barycentrics + depth
    float c0 = b0, c1 = b1, c2 = b2, s = 1.f;
    if (clip) {
        c0 = fminf(fmaxf(b0, 0.f), 1.f); c1 = fminf(fmaxf(b1, 0.f), 1.f); c2 = fminf(fmaxf(b2, 0.f), 1.f);
        s = fmaxf(c0 + c1 + c2, kEps);
        c0 /= s; c1 /= s; c2 /= s;
    }
    const float gzk = gzb ? gzb[i] : 0.f;
    float gq0 = gbary ? gbary[3 * i] : 0.f, gq1 = gbary ? gbary[3 * i + 1] : 0.f, gq2 = gbary ? gbary[3 * i + 2] : 0.f;
    const float go[3] = {gq0, gq1, gq2};             // d loss / d (barycentrics in the ORIGINAL face)
    int k1 = 0, k2 = 1, k3 = 2, kind = -1;
    if (ccode > 1) {                                 // bary_orig = c . M  =>  d c = M . g
        k1 = (ccode - 2) % 3; kind = (ccode - 2) / 3; k2 = (k1 + 1) % 3; k3 = (k1 + 2) % 3;
        const float g1 = go[k1], g2 = go[k2], g3 = go[k3];
        const float g4 = (1.0f - cw2) * g1 + cw2 * g2, g5 = (1.0f - cw3) * g1 + cw3 * g3;
        if (kind == 0) { gq0 = g4; gq1 = g2; gq2 = g5; }
        else if (kind == 1) { gq0 = g5; gq1 = g2; gq2 = g3; }
        else { gq0 = g1; gq1 = g4; gq2 = g5; }
    }
    float dc0 = gq0 + gzk * z0;
    float dc1 = gq1 + gzk * z1;
    float dc2 = gq2 + gzk * z2;
    dz0 += gzk * c0; dz1 += gzk * c1; dz2 += gzk * c2;
    float db0 = dc0, db1 = dc1, db2 = dc2;
    if (clip) {
        const float dot = dc0 * c0 + dc1 * c1 + dc2 * c2;
        const bool live = (fminf(fmaxf(b0, 0.f), 1.f) + fminf(fmaxf(b1, 0.f), 1.f) + fminf(fmaxf(b2, 0.f), 1.f)) > kEps;
        const float e0 = live ? (dc0 - dot) / s : 0.f, e1 = live ? (dc1 - dot) / s : 0.f, e2 = live ? (dc2 - dot) / s : 0.f;
        db0 = (b0 > 0.f && b0 < 1.f) ? e0 : 0.f;
        db1 = (b1 > 0.f && b1 < 1.f) ? e1 : 0.f;
        db2 = (b2 > 0.f && b2 < 1.f) ? e2 : 0.f;
    }
    if (!persp || den > kEps) {
        float dw0 = db0, dw1 = db1, dw2 = db2;          // perspective_correct = False: b = w
        if (persp) {
            const float gs = db0 * b0 + db1 * b1 + db2 * b2;
            const float dt0 = (db0 - gs) / den, dt1 = (db1 - gs) / den, dt2 = (db2 - gs) / den;
            dw0 = dt0 * z1 * z2; dw1 = dt1 * z0 * z2; dw2 = dt2 * z0 * z1;
            dz0 += dt1 * w1 * z2 + dt2 * z1 * w2;
            dz1 += dt0 * w0 * z2 + dt2 * z0 * w2;
            dz2 += dt0 * w0 * z1 + dt1 * z0 * w1;
        }
        const float de0 = dw0 / A, de1 = dw1 / A, de2 = dw2 / A;
        const float dA = -(dw0 * w0 + dw1 * w1 + dw2 * w2) / A;
        gx0 += de1 * -(py - y2) + de2 * (py - y1) + dA * (y2 - y1);
        gy0 += de1 * (px - x2) + de2 * (x1 - px) + dA * (x1 - x2);
        gx1 += de0 * (py - y2) + de2 * -(py - y0) + dA * -(y2 - y0);
        gy1 += de0 * (x2 - px) + de2 * (px - x0) + dA * (x2 - x0);
        gx2 += de0 * -(py - y1) + de1 * (py - y0) + dA * (y1 - y0);
        gy2 += de0 * (px - x1) + de1 * (x0 - px) + dA * -(x1 - x0);
    }
    // ---- signed squared distance to the nearest edge (first minimum), projection parameter constant
    if (gdist) {
        const bool inside = (b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f);
        const float gdd = inside ? -gdist[i] : gdist[i];
        const float d01 = pld2(px, py, x0, y0, x1, y1), d12 = pld2(px, py, x1, y1, x2, y2), d20 = pld2(px, py, x2, y2, x0, y0);
        int e = 0; float dm = d01;
        if (d12 < dm) { dm = d12; e = 1; }
        if (d20 < dm) { dm = d20; e = 2; }
        const float ax = e == 0 ? x0 : (e == 1 ? x1 : x2), ay = e == 0 ? y0 : (e == 1 ? y1 : y2);
        const float bx = e == 0 ? x1 : (e == 1 ? x2 : x0), by = e == 0 ? y1 : (e == 1 ? y2 : y0);
        const float bax = bx - ax, bay = by - ay, l2 = bax * bax + bay * bay;
        float gax, gay, gbx, gby;
        if (l2 <= kEps) {
            gax = 0.f; gay = 0.f; gbx = gdd * 2.f * (bx - px); gby = gdd * 2.f * (by - py);
        } else {
            float t = (bax * (px - ax) + bay * (py - ay)) / l2;
            t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
            const float qx = ax + t * bax - px, qy = ay + t * bay - py;
            gax = gdd * (1.f - t) * 2.f * qx; gay = gdd * (1.f - t) * 2.f * qy;
            gbx = gdd * t * 2.f * qx; gby = gdd * t * 2.f * qy;
        }
        if (e == 0) { gx0 += gax; gy0 += gay; gx1 += gbx; gy1 += gby; }
        else if (e == 1) { gx1 += gax; gy1 += gay; gx2 += gbx; gy2 += gby; }
        else { gx2 += gax; gy2 += gay; gx0 += gbx; gy0 += gby; }
    }
    if (ccode > 1) {
        // sub-triangle vertices q0 q1 q2 -> the face's p1 p2 p3 (indices k1 k2 k3 in the face) through p4 = cut(p1, p2, w2),
        // p5 = cut(p1, p3, w3), w2 = (z1 - zc) / (z1 - z2), w3 likewise; M's dependence on w2 / w3 included
        const float gq[3][3] = {{gx0, gy0, dz0}, {gx1, gy1, dz1}, {gx2, gy2, dz2}};
        float gp[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};      // [p1, p2, p3][x, y, z]
        const float P[3][3] = {{ov[3 * k1], ov[3 * k1 + 1], ov[3 * k1 + 2]}, {ov[3 * k2], ov[3 * k2 + 1], ov[3 * k2 + 2]},
                               {ov[3 * k3], ov[3 * k3 + 1], ov[3 * k3 + 2]}};
        float dw2 = 0.f, dw3 = 0.f;
        // which sub-triangle vertex is what: 0 = p1, 1 = p2, 2 = p3, 4 = p4, 5 = p5
        const int who[3] = {kind == 0 ? 4 : (kind == 1 ? 5 : 0), kind == 2 ? 4 : 1, kind == 1 ? 2 : 5};
        // the clipped (and possibly clamped / renormalised) coordinates c the forward multiplied into M
        float cc0 = b0, cc1 = b1, cc2 = b2;
        if (clip) { cc0 = c0; cc1 = c1; cc2 = c2; }
        const float cq[3] = {cc0, cc1, cc2};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int wq = who[q];
            if (wq < 3) {
                gp[wq][0] += gq[q][0]; gp[wq][1] += gq[q][1]; gp[wq][2] += gq[q][2];
            } else {
                const int o = (wq == 4) ? 1 : 2;                 // the edge's far end: p2 for p4, p3 for p5
                const float w = (wq == 4) ? cw2 : cw3;
                float dw = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {                    // x, y
                    const float g = gq[q][a];
                    if (persp) {
                        gp[0][a] += g * (1.0f - w) * P[0][2] / z_clip; gp[0][2] += g * (1.0f - w) * P[0][a] / z_clip;
                        gp[o][a] += g * w * P[o][2] / z_clip;          gp[o][2] += g * w * P[o][a] / z_clip;
                        dw += g * (P[o][a] * P[o][2] - P[0][a] * P[0][2]) / z_clip;
                    } else {
                        gp[0][a] += g * (1.0f - w); gp[o][a] += g * w;
                        dw += g * (P[o][a] - P[0][a]);
                    }
                }
                // M's row for this vertex: (1 - w) e_p1 + w e_far
                dw += cq[q] * (go[(wq == 4) ? k2 : k3] - go[k1]);
                if (wq == 4) dw2 += dw; else dw3 += dw;
            }
        }
        const float d12 = P[0][2] - P[1][2], d13 = P[0][2] - P[2][2];
        gp[0][2] += dw2 * (z_clip - P[1][2]) / (d12 * d12) + dw3 * (z_clip - P[2][2]) / (d13 * d13);
        gp[1][2] += dw2 * (P[0][2] - z_clip) / (d12 * d12);
        gp[2][2] += dw3 * (P[0][2] - z_clip) / (d13 * d13);
        const int ks[3] = {k1, k2, k3};
#pragma unroll
        for (int r = 0; r < 3; ++r) {          // gp[r] belongs to the face's vertex number ks[r]
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (ks[r] == q) c9[3 * q + c] = gp[r][c];
        }
        return;
    }
    c9[0] = gx0; c9[1] = gy0; c9[2] = dz0; c9[3] = gx1; c9[4] = gy1; c9[5] = dz1; c9[6] = gx2; c9[7] = gy2; c9[8] = dz2;
}

// One workgroup per 16x16-pixel tile, every thread walks its pixel's K layers.  The nine contributions of a fragment are
// summed per FACE in an LDS table (open addressing on the face index) and one set of nine global atomics per (tile, face)
// goes out -- what raster_bwd_kernel does for the specialised K = 1 path (round 2 issued nine global float atomics per
// fragment: 20 ms per step once config 5's mesh fills the screen).  DET 0: float table + float global atomics; DET 1: the
// bound pass of the deterministic variant (partials[block] = sum of |contributions|); DET 2: 64-bit fixed point with the
// scale derived from that bound (gndc is then the int64 accumulator array).  A fragment that finds no slot within
// kSoftProbe steps (K layers can bring more faces than the table holds) adds to global memory directly.
constexpr int kSoftFaceSlots = 512;

template <int DET>
__global__ __launch_bounds__(256) void raster_k_bwd_kernel(const float *__restrict__ gbary, const float *__restrict__ gzb,
                                                           const float *__restrict__ gdist, const int32_t *__restrict__ p2f,
                                                           const float *__restrict__ ndc, const int32_t *__restrict__ faces,
                                                           int B, int V, int S, int K, int clip, int persp, int tiles_x,
                                                           float *__restrict__ gndc, const int32_t *__restrict__ frag_slot,
                                                           float z_clip, const st3d_det::DetHeader *__restrict__ det,
                                                           float *__restrict__ partials) {
    typedef typename std::conditional<DET == 2, unsigned long long, float>::type acc_t;
    __shared__ int s_key[kSoftFaceSlots];
    __shared__ acc_t s_acc[DET == 1 ? 1 : kSoftFaceSlots][9];
    __shared__ float s4[4];
    const int tid = threadIdx.x;
    if (DET != 1) {
        for (int e = tid; e < kSoftFaceSlots; e += 256) s_key[e] = -1;
        for (int e = tid; e < kSoftFaceSlots * 9; e += 256) (&s_acc[0][0])[e] = (acc_t)0;
        __syncthreads();
    }
    const double dscale = DET == 2 ? det->scale : 1.0;
    const size_t HW = (size_t)S * S;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int yi = ty * 16 + (tid >> 4), xi = tx * 16 + (tid & 15);
    const float *vb = ndc + (size_t)b * V * 3;
    float bound = 0.f;
    if (yi < S && xi < S) {
        const size_t pix = (size_t)b * HW + (size_t)yi * S + xi;
        for (int k = 0; k < K; ++k) {
            const size_t i = pix * K + k;
            const int f = p2f[i];
            if (f < 0) continue;
            float c9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            raster_k_frag_grad(gbary, gzb, gdist, vb, faces, S, clip, persp, frag_slot, z_clip, i, f, xi, yi, c9);
            if (DET == 1) {
#pragma unroll
                for (int c = 0; c < 9; ++c) bound += fabsf(c9[c]);
                continue;
            }
            int slot = (int)(((unsigned)f * 2654435761u) >> 23) & (kSoftFaceSlots - 1);
            bool found = false;
            for (int tries = 0; tries < kSoftProbe; ++tries) {
                const int prev = atomicCAS(&s_key[slot], -1, f);
                if (prev == -1 || prev == f) { found = true; break; }
                slot = (slot + 1) & (kSoftFaceSlots - 1);
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const size_t o = (size_t)b * V * 3 + 3 * (size_t)faces[3 * f + c / 3] + (c % 3);
                if (DET == 2) {
                    const unsigned long long q = (unsigned long long)st3d_det::det_quantise(c9[c], dscale);
                    if (found) atomicAdd(reinterpret_cast<unsigned long long *>(&s_acc[slot][c]), q);
                    else atomicAdd(reinterpret_cast<unsigned long long *>(gndc) + o, q);
                } else {
                    if (found) atomicAdd(reinterpret_cast<float *>(&s_acc[slot][c]), c9[c]);
                    else atomicAdd(gndc + o, c9[c]);
                }
            }
        }
    }
    if (DET == 1) {
        const float t = st3d_det::det_block_sum(bound, s4);
        if (tid == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
        return;
    }
    __syncthreads();
    for (int e = tid; e < kSoftFaceSlots * 9; e += 256) {
        const int slot = e / 9, c = e - slot * 9;
        const int fk = s_key[slot];
        if (fk < 0) continue;
        const acc_t v = s_acc[slot][c];
        if (v == (acc_t)0) continue;
        const size_t o = (size_t)b * V * 3 + 3 * (size_t)faces[3 * fk + c / 3] + (c % 3);
        if (DET == 2) atomicAdd(reinterpret_cast<unsigned long long *>(gndc) + o, (unsigned long long)v);
        else atomicAdd(gndc + o, (float)v);
    }
}

template <int K>
void launch_raster_k(const float4 *rec, int B, int F, int S, float blur, int clip, int cull, int persp, int rpf, int32_t *p2f,
                     float *zbuf, float *bary, float *dists, int32_t *slots, hipStream_t s) {
    const int tiles = st3d::cdiv(S, TILE);
    raster_k_kernel<K><<<dim3(tiles, tiles, B), 256, 0, s>>>(rec, F * rpf, S, blur, clip, cull, persp, rpf, p2f, zbuf, bary, dists,
                                                             slots);
}

}  // namespace

extern "C" size_t st3d_clip_records_bytes(int B, int F) { return (size_t)B * (size_t)F * 2 * 3 * sizeof(float4); }

extern "C" int st3d_face_setup_clip(const float *verts_ndc, const int32_t *faces, int B, int V, int F, float z_clip,
                                    int perspective_correct, void *face_records, size_t records_bytes, st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts_ndc && faces && face_records);
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && z_clip > 0.f && records_bytes >= st3d_clip_records_bytes(B, F));
    ST3D_CHECK_ARG(((uintptr_t)face_records & 15) == 0);
    face_setup_clip_kernel<<<st3d::cdiv((long)B * F, 256), 256, 0, st3d::as_stream(stream)>>>(
        verts_ndc, faces, B, V, F, z_clip, perspective_correct, reinterpret_cast<float4 *>(face_records));
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_raster_soft_fwd(const float *face_records, int B, int F, int S, int K, float blur_radius, int clip_bary,
                                    int cull_backfaces, int perspective_correct, int records_per_face, int32_t *frag_slot,
                                    int32_t *pix_to_face, float *zbuf, float *bary, float *dists, st3d_stream_t stream) {
    ST3D_CHECK_ARG(face_records && pix_to_face && zbuf && bary && dists);
    ST3D_CHECK_ARG(B > 0 && F > 0 && S > 0 && K >= 1 && K <= 8 && blur_radius >= 0.f);
    ST3D_CHECK_ARG(records_per_face == 1 || records_per_face == 2);
    ST3D_CHECK_ARG(((uintptr_t)face_records & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    const float4 *rec = reinterpret_cast<const float4 *>(face_records);
    switch (K) {
        case 1: launch_raster_k<1>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 2: launch_raster_k<2>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 3: launch_raster_k<3>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 4: launch_raster_k<4>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 5: launch_raster_k<5>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 6: launch_raster_k<6>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        case 7: launch_raster_k<7>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
        default: launch_raster_k<8>(rec, B, F, S, blur_radius, clip_bary, cull_backfaces, perspective_correct, records_per_face, pix_to_face, zbuf, bary, dists, frag_slot, s); break;
    }
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_shade_soft_fwd(const int32_t *pix_to_face, const float *bary, const float *zbuf, const float *dists,
                                   const float *verts_uvs, const int32_t *faces_uvs, const float *texture, int B, int S, int T,
                                   int K, float sigma, float gamma, const float *background, float *rgb, float *alpha,
                                   st3d_stream_t stream) {
    ST3D_CHECK_ARG(pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture && background && rgb && alpha);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && K >= 1 && sigma > 0.f && gamma > 0.f);
    SoftArgs a{pix_to_face, bary, zbuf, dists, verts_uvs, faces_uvs, texture, B, S, T, K, sigma, gamma, background[0],
               background[1], background[2]};
    const size_t n = (size_t)B * S * S;
    soft_shade_kernel<0><<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(a, rgb, alpha, nullptr, nullptr, nullptr,
                                                                                        nullptr, nullptr, 0, nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_shade_soft_bwd(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                                   const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                                   int B, int S, int T, int K, float sigma, float gamma, const float *background,
                                   float *grad_texture, float *grad_bary, float *grad_zbuf, float *grad_dists,
                                   st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_rgb && pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture && background);
    ST3D_CHECK_ARG(grad_texture || grad_bary || grad_zbuf || grad_dists);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && K >= 1 && sigma > 0.f && gamma > 0.f);
    SoftArgs a{pix_to_face, bary, zbuf, dists, verts_uvs, faces_uvs, texture, B, S, T, K, sigma, gamma, background[0],
               background[1], background[2]};
    const int tiles = (S + 15) / 16;
    soft_shade_kernel<1><<<dim3(tiles * tiles, B), 256, 0, st3d::as_stream(stream)>>>(a, nullptr, nullptr, grad_rgb, grad_texture,
                                                                                      grad_bary, grad_zbuf, grad_dists, tiles, nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

namespace { constexpr int kSoftDetPartials = 1024; }

extern "C" size_t st3d_shade_soft_bwd_det_workspace_bytes(int T) {
    return st3d_det::workspace_bytes((size_t)T * T * 3, kSoftDetPartials);
}

// st3d_shade_soft_bwd with a bitwise reproducible texture gradient (64-bit fixed-point accumulation, det.h; the bound of
// every partial sum is sum |grad_rgb|: blend and bilinear weights are <= 1).  grad_texture is ACCUMULATED into.
extern "C" int st3d_shade_soft_bwd_det(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                                       const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                                       int B, int S, int T, int K, float sigma, float gamma, const float *background,
                                       float *grad_texture, float *grad_bary, float *grad_zbuf, float *grad_dists,
                                       void *workspace, size_t workspace_bytes, st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_rgb && pix_to_face && bary && zbuf && dists && verts_uvs && faces_uvs && texture && background);
    ST3D_CHECK_ARG(grad_texture && workspace);
    ST3D_CHECK_ARG(B > 0 && S > 0 && T > 1 && K >= 1 && sigma > 0.f && gamma > 0.f);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_shade_soft_bwd_det_workspace_bytes(T) && ((uintptr_t)workspace & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    SoftArgs a{pix_to_face, bary, zbuf, dists, verts_uvs, faces_uvs, texture, B, S, T, K, sigma, gamma, background[0],
               background[1], background[2]};
    auto *hdr = reinterpret_cast<st3d_det::DetHeader *>(workspace);
    float *partials = st3d_det::partials_of(workspace);
    long long *acc = st3d_det::accum_of(workspace, kSoftDetPartials);
    const size_t npx = (size_t)B * 3 * S * S, nacc = (size_t)T * T * 3;
    st3d_det::det_abs_sum_kernel<<<kSoftDetPartials, 256, 0, s>>>(grad_rgb, npx, partials);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_scale_kernel<<<1, 256, 0, s>>>(partials, kSoftDetPartials, hdr);
    ST3D_LAUNCH_CHECK();
    ST3D_HIP(hipMemsetAsync(acc, 0, nacc * sizeof(long long), s));
    const int tiles = (S + 15) / 16;
    soft_shade_kernel<1, 1><<<dim3(tiles * tiles, B), 256, 0, s>>>(a, nullptr, nullptr, grad_rgb, reinterpret_cast<float *>(acc),
                                                                   grad_bary, grad_zbuf, grad_dists, tiles, hdr);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_convert_kernel<<<st3d::cdiv((long)nacc, 256), 256, 0, s>>>(acc, nacc, hdr, 1, grad_texture);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_raster_soft_bwd(const float *grad_bary, const float *grad_zbuf, const float *grad_dists,
                                    const int32_t *pix_to_face, const float *verts_ndc, const int32_t *faces, int B, int V,
                                    int F, int S, int K, int clip_bary, int perspective_correct, const int32_t *frag_slot,
                                    float z_clip, float *grad_verts_ndc, st3d_stream_t stream) {
    ST3D_CHECK_ARG(pix_to_face && verts_ndc && faces && grad_verts_ndc && (grad_bary || grad_zbuf || grad_dists));
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0 && K >= 1);
    hipStream_t s = st3d::as_stream(stream);
    ST3D_HIP(hipMemsetAsync(grad_verts_ndc, 0, (size_t)B * V * 3 * sizeof(float), s));
    const int tiles = (S + 15) / 16;
    raster_k_bwd_kernel<0><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, grad_zbuf, grad_dists, pix_to_face, verts_ndc, faces, B,
                                                                  V, S, K, clip_bary, perspective_correct, tiles, grad_verts_ndc,
                                                                  frag_slot, z_clip, nullptr, nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" size_t st3d_raster_soft_bwd_det_workspace_bytes(int B, int V, int S) {
    const size_t tiles = (size_t)((S + 15) / 16);
    return st3d_det::workspace_bytes((size_t)B * V * 3, tiles * tiles * B);
}

// st3d_raster_soft_bwd with a bitwise reproducible result (fixed-point accumulation, det.h): a first pass bounds the
// partial sums, the second accumulates.
extern "C" int st3d_raster_soft_bwd_det(const float *grad_bary, const float *grad_zbuf, const float *grad_dists,
                                        const int32_t *pix_to_face, const float *verts_ndc, const int32_t *faces, int B, int V,
                                        int F, int S, int K, int clip_bary, int perspective_correct, const int32_t *frag_slot,
                                        float z_clip, float *grad_verts_ndc, void *workspace, size_t workspace_bytes,
                                        st3d_stream_t stream) {
    ST3D_CHECK_ARG(pix_to_face && verts_ndc && faces && grad_verts_ndc && workspace && (grad_bary || grad_zbuf || grad_dists));
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0 && K >= 1);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_raster_soft_bwd_det_workspace_bytes(B, V, S) && ((uintptr_t)workspace & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    const int tiles = (S + 15) / 16;
    const size_t np = (size_t)tiles * tiles * B, nacc = (size_t)B * V * 3;
    auto *hdr = reinterpret_cast<st3d_det::DetHeader *>(workspace);
    float *partials = st3d_det::partials_of(workspace);
    long long *acc = st3d_det::accum_of(workspace, np);
    raster_k_bwd_kernel<1><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, grad_zbuf, grad_dists, pix_to_face, verts_ndc, faces, B,
                                                                  V, S, K, clip_bary, perspective_correct, tiles, nullptr, frag_slot,
                                                                  z_clip, nullptr, partials);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_scale_kernel<<<1, 256, 0, s>>>(partials, (int)np, hdr);
    ST3D_LAUNCH_CHECK();
    ST3D_HIP(hipMemsetAsync(acc, 0, nacc * sizeof(long long), s));
    raster_k_bwd_kernel<2><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, grad_zbuf, grad_dists, pix_to_face, verts_ndc, faces, B,
                                                                  V, S, K, clip_bary, perspective_correct, tiles,
                                                                  reinterpret_cast<float *>(acc), frag_slot, z_clip, hdr, nullptr);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_convert_kernel<<<st3d::cdiv((long)nacc, 256), 256, 0, s>>>(acc, nacc, hdr, 0, grad_verts_ndc);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
