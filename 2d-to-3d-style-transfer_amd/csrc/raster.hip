// raster.hip -- hard rasteriser for gfx950 (K = 1 face per pixel, blur_radius = 0,
// perspective-correct barycentrics): the device work behind utils.py:69
// `renderer(meshes_world=..., cameras=camera)` -> PyTorch3D MeshRasterizer, configured at
// first_approach.py:107 / second_approach.py:101.
//
// Layout / kernels (all B views of a batch in one launch each):
//   project_verts_kernel   (B*V threads)   world -> (x_ndc, y_ndc, z_view)
//   face_setup_kernel      (B*F threads)   gathers the 3 projected vertices of every face into a
//                                          48-byte record (3 x float4) and packs the 16-pixel tiles its
//                                          bounding box can touch into one 32-bit word
//   raster_tile_kernel     one 256-thread workgroup per 16x16 pixel tile: per super-round of 2048 faces every
//                          wave sweeps the packed tile ranges of its own 512 faces (4 B per face, not the
//                          record) and compacts the hits in face order (wave ballots) without a barrier; then
//                          only the hits' records are fetched into LDS and every lane (= pixel) walks the list
//                          (same-address broadcast reads, no bank conflicts).
// Measured at config 2 (8 views, 512^2, cow): 0.18 ms (0.19 before the sweep was decoupled from the record fetch,
// round 2), of which ~0.05 ms is the sweep (mesh pushed off screen: every tile reads every face's word, 192 MB of L2
// reads per 8-view batch); the rest is the per-pixel walk, bounded by the densest tiles (up to 330 one-pixel faces per
// tile, 50 on average over the 32 % non-empty tiles).  Variants that did NOT help and were dropped
// (tools/shade_bench.py): 1024 faces per pass (40 KB of LDS), prefetching records, bbox/area staged in
// LDS with a sign early-out before the six divisions, one 8x8 quadrant per wave with wave-level culling.
// Algorithmic bytes per view = F*52 + S*S*24 written.  Built with -ffp-contract=off so the arithmetic
// is the same operation sequence as oracle/raster_ref.c (bit-comparable).
#include <type_traits>

#include <stdlib.h>

#include "common.h"
#include "det.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr int TILE = 16;       // 16x16 pixels per workgroup
constexpr int LIST_CAP = 512;  // face records staged in LDS per walk
constexpr int SUPER = 2048;    // faces swept per super-round (512 per wave)

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float point_line_dist2(float px, float py, float ax, float ay, float bx, float by) {
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

__global__ void project_verts_kernel(const float *__restrict__ verts, int V, const float *__restrict__ R,
                                     const float *__restrict__ T, int B, float s, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V, v = i - b * V;
    const float *r = R + 9 * b, *t = T + 3 * b;
    const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
    const float xv = x * r[0] + y * r[3] + z * r[6] + t[0];
    const float yv = x * r[1] + y * r[4] + z * r[7] + t[1];
    const float zv = x * r[2] + y * r[5] + z * r[8] + t[2];
    out[3 * (size_t)i + 0] = (s * xv) / zv;
    out[3 * (size_t)i + 1] = (s * yv) / zv;
    out[3 * (size_t)i + 2] = zv;
}

// record: [x0 y0 z0 x1][y1 z1 x2 y2][z2 valid 0 0]
// words (optional, one per face, view stride Fp = F rounded up to 2): the 16-pixel tiles the face's bounding box can
// touch, packed tx0 | tx1<<8 | ty0<<16 | ty1<<24 -- a conservative superset (one pixel of slack each side; the exact
// bbox test is repeated per pixel).  The tile kernel sweeps these 4 bytes per face instead of the 48-byte record
// (every tile reads every face: 8192 tiles x 5856 faces x 48 B = 2.3 GB of L2 reads per 8-view batch before, 0.19 GB now).
constexpr unsigned kEmptyRange = 1u;     // tx0 = 1 > tx1 = 0

__device__ __forceinline__ void pixel_span(float cmin, float cmax, int S, int &lo, int &hi) {
    // pixel index i has NDC centre 1 - (2i+1)/S  <=>  i = ((1 - c) * S - 1) / 2
    const float a = fminf(fmaxf(((1.0f - cmax) * (float)S - 1.0f) * 0.5f, -4.0f), (float)S + 4.0f);
    const float b = fminf(fmaxf(((1.0f - cmin) * (float)S - 1.0f) * 0.5f, -4.0f), (float)S + 4.0f);
    lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
}

// near_flag (optional): set when a face that will be rasterised has a vertex nearer than z_clip -- PyTorch3D would clip or
// drop it (z_clip_value = znear / 2); this specialised path does not clip, so the host turns the flag into an error that
// points at the general kernels (soft.hip), which do.
__global__ void face_setup_kernel(const float *__restrict__ ndc, const int32_t *__restrict__ faces, int B, int V, int F,
                                  float4 *__restrict__ rec, int S, int Fp, unsigned *__restrict__ words, float z_clip,
                                  int32_t *__restrict__ near_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = words ? Fp : F;
    if (i >= B * stride) return;
    const int b = i / stride, f = i - b * stride;
    if (f >= F) { words[i] = kEmptyRange; return; }
    const float *vb = ndc + (size_t)b * V * 3;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float x0 = vb[3 * i0], y0 = vb[3 * i0 + 1], z0 = vb[3 * i0 + 2];
    const float x1 = vb[3 * i1], y1 = vb[3 * i1 + 1], z1 = vb[3 * i1 + 2];
    const float x2 = vb[3 * i2], y2 = vb[3 * i2 + 1], z2 = vb[3 * i2 + 2];
    const float zmax = fmaxf(z0, fmaxf(z1, z2));
    const float area = edge_fn(x2, y2, x0, y0, x1, y1);
    const bool valid = !(zmax < kEps) && !(area <= kEps && area >= -kEps);
    if (near_flag && valid && fminf(z0, fminf(z1, z2)) < z_clip) atomicOr(near_flag, 1);
    const size_t o = (size_t)b * F + f;
    rec[3 * o + 0] = make_float4(x0, y0, z0, x1);
    rec[3 * o + 1] = make_float4(y1, z1, x2, y2);
    rec[3 * o + 2] = make_float4(z2, valid ? 1.f : 0.f, 0.f, 0.f);
    if (words) {
        unsigned w = kEmptyRange;
        if (valid) {
            int xlo, xhi, ylo, yhi;
            pixel_span(fminf(x0, fminf(x1, x2)), fmaxf(x0, fmaxf(x1, x2)), S, xlo, xhi);
            pixel_span(fminf(y0, fminf(y1, y2)), fmaxf(y0, fmaxf(y1, y2)), S, ylo, yhi);
            if (xhi >= 0 && yhi >= 0 && xlo <= S - 1 && ylo <= S - 1) {
                xlo = max(xlo, 0); ylo = max(ylo, 0); xhi = min(xhi, S - 1); yhi = min(yhi, S - 1);
                w = (unsigned)(xlo / TILE) | ((unsigned)(xhi / TILE) << 8) | ((unsigned)(ylo / TILE) << 16) |
                    ((unsigned)(yhi / TILE) << 24);
            }
        }
        words[i] = w;
    }
}

struct Best {
    int f;
    float z, b0, b1, b2;
};

__device__ __forceinline__ void eval_face(int f, float x0, float y0, float z0, float x1, float y1, float z1, float x2,
                                          float y2, float z2, float xf, float yf, Best &best) {
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) return;
    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
    const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
    const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
    const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
    const float t0 = w0 * z1 * z2;
    const float t1 = z0 * w1 * z2;
    const float t2 = z0 * z1 * w2;
    const float den = fmaxf(t0 + t1 + t2, kEps);
    const float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
    const float pz = b0 * z0 + b1 * z1 + b2 * z2;
    if (pz < 0.f) return;
    if (!((b0 > 0.f) && (b1 > 0.f) && (b2 > 0.f))) return;  // blur_radius == 0: only inside pixels survive
    if (best.f < 0 || pz < best.z) {                          // list is in face order: ties keep the smaller index
        best.f = f; best.z = pz; best.b0 = b0; best.b1 = b1; best.b2 = b2;
    }
}

// Coarse level (round 2): bins of BIN x BIN tiles.  raster_bin_kernel sweeps every face's packed tile range ONCE per bin and
// leaves, per (view, bin), the faces that touch it, in face order; a tile then sweeps its bin's list (typically a few
// hundred entries) instead of all F words -- the all-faces sweep is 192 MB of L2 reads per 8-view batch for the cow and
// 24 GB (4 ms) for a 94 k-face mesh at 16 x 1024^2.  A bin whose list would exceed its capacity is marked (-1) and its tiles
// sweep all faces as before; results are bit-identical either way (same candidates after the exact test, same order).
constexpr int BIN = 4;

__global__ __launch_bounds__(256) void raster_bin_kernel(const unsigned *__restrict__ words, int Fp, int nbx, int cap,
                                                         int *__restrict__ bin_count, int *__restrict__ bin_list) {
    __shared__ int s_wcnt[4];
    __shared__ int s_list[4][512];
    const int b = blockIdx.y, bin = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned tx0 = (unsigned)(bin % nbx) * BIN, ty0 = (unsigned)(bin / nbx) * BIN;
    const uint2 *wb = reinterpret_cast<const uint2 *>(words + (size_t)b * Fp);
    int *list = bin_list + ((size_t)b * gridDim.x + bin) * cap;
    int running = 0;
    for (int base = 0; base < Fp; base += SUPER) {
        int cnt = 0;
        uint2 w4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int f0 = base + 512 * wave + 128 * it + 2 * lane;
            w4[it] = (f0 < Fp) ? wb[f0 >> 1] : make_uint2(kEmptyRange, kEmptyRange);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int f0 = base + 512 * wave + 128 * it + 2 * lane;
            const unsigned wv[2] = {w4[it].x, w4[it].y};
            bool hit[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned w = wv[j];
                // kEmptyRange (tx0 = 1 > tx1 = 0: culled faces AND the lanes past the last face) would pass a RANGE overlap
                // test with the first bin -- it is excluded by name, and so is anything past the mesh
                hit[j] = f0 + j < Fp && w != kEmptyRange && (w & 255u) <= tx0 + BIN - 1 && tx0 <= ((w >> 8) & 255u) &&
                         ((w >> 16) & 255u) <= ty0 + BIN - 1 && ty0 <= (w >> 24);
            }
            const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int pos = cnt + __popcll(m0 & lt) + __popcll(m1 & lt);
            if (hit[0]) s_list[wave][pos] = f0;
            if (hit[1]) s_list[wave][pos + (hit[0] ? 1 : 0)] = f0 + 1;
            cnt += __popcll(m0) + __popcll(m1);
        }
        if (lane == 0) s_wcnt[wave] = cnt;
        __syncthreads();
        const int o1 = s_wcnt[0], o2 = o1 + s_wcnt[1], o3 = o2 + s_wcnt[2], total = o3 + s_wcnt[3];
        if (running + total > cap - 2) {            // (workgroup-uniform) does not fit: the bin's tiles sweep all faces
            if (tid == 0) bin_count[(size_t)b * gridDim.x + bin] = -1;
            return;
        }
        for (int i = tid; i < total; i += 256)
            list[running + i] = i < o1 ? s_list[0][i] : (i < o2 ? s_list[1][i - o1] : (i < o3 ? s_list[2][i - o2] : s_list[3][i - o3]));
        running += total;
        __syncthreads();
    }
    if (tid == 0) {
        bin_count[(size_t)b * gridDim.x + bin] = running;
        list[running] = -1; list[running + 1] = -1;         // readers fetch entries in pairs
    }
}

__global__ __launch_bounds__(256) void raster_tile_kernel(const float4 *__restrict__ rec, const unsigned *__restrict__ words,
                                                          int F, int Fp, int S, int32_t *__restrict__ pix_to_face,
                                                          float *__restrict__ zbuf, float *__restrict__ bary,
                                                          float *__restrict__ dists, const int *__restrict__ bin_count,
                                                          const int *__restrict__ bin_list, int nbx, int cap) {
    __shared__ float s_face[LIST_CAP][9];
    __shared__ int s_fidx[LIST_CAP];
    __shared__ int s_wcnt[4];
    __shared__ int s_list[4][512];       // per wave: the faces of its slice of the super-round that touch this tile

    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int px = blockIdx.x * TILE + (tid & (TILE - 1));
    const int py = blockIdx.y * TILE + (tid >> 4);
    const bool in_img = px < S && py < S;
    const float xf = pix_to_ndc(S - 1 - px, S);
    const float yf = pix_to_ndc(S - 1 - py, S);
    const float4 *rb = rec + (size_t)b * F * 3;
    const uint2 *wb = reinterpret_cast<const uint2 *>(words + (size_t)b * Fp);
    const unsigned bx = blockIdx.x, by = blockIdx.y;
    Best best;
    best.f = -1; best.z = 0.f; best.b0 = best.b1 = best.b2 = 0.f;

    // Two decoupled phases per super-round of 2048 faces (round 2; before, every 512-face pass had its own barrier ->
    // record fetch -> barrier -> walk -> barrier chain, twelve times over for the cow):
    //  1. sweep: every wave sweeps ITS 512 consecutive faces' packed tile ranges (4 x 8-byte loads per lane, all in flight
    //     together) and compacts the hits, in face order, into its own index list -- no barrier, no record traffic;
    //  2. one barrier; the four lists concatenated are the tile's faces in face order (ties in depth keep the smaller
    //     index): their records are fetched into LDS in one go (chunks of LIST_CAP) and every lane (= pixel) walks them.
    // candidates: the bin's list when there is one (entries in face order), otherwise all faces
    int ncand = Fp;
    const int2 *cl = nullptr;
    if (bin_count) {
        const int bin = ((int)by / BIN) * nbx + (int)bx / BIN, nbins = nbx * ((int)(gridDim.y + BIN - 1) / BIN);
        const int c = bin_count[(size_t)b * nbins + bin];
        if (c >= 0) {
            ncand = c;
            cl = reinterpret_cast<const int2 *>(bin_list + ((size_t)b * nbins + bin) * cap);
        }
    }
    const unsigned *wsv = words + (size_t)b * Fp;
    for (int base = 0; base < ncand; base += SUPER) {
        int cnt = 0;
        uint2 w4[4];
        int2 f4[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e0 = base + 512 * wave + 128 * it + 2 * lane;
            if (cl) {
                f4[it] = (e0 < ncand) ? cl[e0 >> 1] : make_int2(-1, -1);       // (the list ends with a pair of -1)
            } else {
                f4[it] = make_int2(e0, e0 + 1);
                w4[it] = (e0 < Fp) ? wb[e0 >> 1] : make_uint2(kEmptyRange, kEmptyRange);
            }
        }
        if (cl) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                w4[it].x = (unsigned)f4[it].x < (unsigned)Fp ? wsv[f4[it].x] : kEmptyRange;      // (-1 terminators and anything else out of range)
                w4[it].y = (unsigned)f4[it].y < (unsigned)Fp ? wsv[f4[it].y] : kEmptyRange;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const unsigned wv[2] = {w4[it].x, w4[it].y};
            bool hit[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned w = wv[j];
                hit[j] = (w & 255u) <= bx && bx <= ((w >> 8) & 255u) && ((w >> 16) & 255u) <= by && by <= (w >> 24);
            }
            const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int pos = cnt + __popcll(m0 & lt) + __popcll(m1 & lt);
            if (hit[0]) s_list[wave][pos] = f4[it].x;
            if (hit[1]) s_list[wave][pos + (hit[0] ? 1 : 0)] = f4[it].y;
            cnt += __popcll(m0) + __popcll(m1);
        }
        if (lane == 0) s_wcnt[wave] = cnt;
        __syncthreads();
        const int o1 = s_wcnt[0], o2 = o1 + s_wcnt[1], o3 = o2 + s_wcnt[2], total = o3 + s_wcnt[3];
        for (int cb = 0; cb < total; cb += LIST_CAP) {
            const int n = min(LIST_CAP, total - cb);
#pragma unroll
            for (int h = 0; h < LIST_CAP / 256; ++h) {
                const int slot = tid + 256 * h, i = cb + slot;
                if (slot < n) {
                    const int f = i < o1 ? s_list[0][i] : (i < o2 ? s_list[1][i - o1] : (i < o3 ? s_list[2][i - o2] : s_list[3][i - o3]));
                    const float4 r0 = rb[3 * (size_t)f], r1 = rb[3 * (size_t)f + 1], r2 = rb[3 * (size_t)f + 2];
                    s_fidx[slot] = f;
                    s_face[slot][0] = r0.x; s_face[slot][1] = r0.y; s_face[slot][2] = r0.z;
                    s_face[slot][3] = r0.w; s_face[slot][4] = r1.x; s_face[slot][5] = r1.y;
                    s_face[slot][6] = r1.z; s_face[slot][7] = r1.w; s_face[slot][8] = r2.x;
                }
            }
            __syncthreads();
            for (int i = 0; i < n; ++i) {
                eval_face(s_fidx[i], s_face[i][0], s_face[i][1], s_face[i][2], s_face[i][3], s_face[i][4], s_face[i][5],
                          s_face[i][6], s_face[i][7], s_face[i][8], xf, yf, best);
            }
            __syncthreads();          // the record list (and, after the last chunk, the index lists) are rewritten next
        }
        if (total == 0) __syncthreads();    // (workgroup-uniform) s_wcnt is rewritten by the next super-round
    }
    if (!in_img) return;
    const size_t p = ((size_t)b * S + py) * S + px;
    if (best.f >= 0) {
        const float4 r0 = rb[3 * (size_t)best.f], r1 = rb[3 * (size_t)best.f + 1];
        const float d01 = point_line_dist2(xf, yf, r0.x, r0.y, r0.w, r1.x);
        const float d12 = point_line_dist2(xf, yf, r0.w, r1.x, r1.z, r1.w);
        const float d20 = point_line_dist2(xf, yf, r1.z, r1.w, r0.x, r0.y);
        const float d = fminf(d01, fminf(d12, d20));
        pix_to_face[p] = best.f; zbuf[p] = best.z; dists[p] = -d;
        bary[3 * p] = best.b0; bary[3 * p + 1] = best.b1; bary[3 * p + 2] = best.b2;
    } else {
        pix_to_face[p] = -1; zbuf[p] = -1.f; dists[p] = -1.f;
        bary[3 * p] = -1.f; bary[3 * p + 1] = -1.f; bary[3 * p + 2] = -1.f;
    }
}

}  // namespace

extern "C" int st3d_project_verts(const float *verts, int V, const float *R, const float *T, int B,
                                  float inv_tan_half_fov, float *verts_ndc, st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts && R && T && verts_ndc);
    ST3D_CHECK_ARG(V > 0 && B > 0);
    const int n = B * V;
    project_verts_kernel<<<st3d::cdiv(n, 256), 256, 0, st3d::as_stream(stream)>>>(verts, V, R, T, B, inv_tan_half_fov,
                                                                                  verts_ndc);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

// face records (48 B per face and view) + packed tile ranges (4 B, view stride rounded up to 2 faces)
extern "C" size_t st3d_raster_workspace_bytes(int B, int F) {
    return (size_t)B * (size_t)F * 3 * sizeof(float4) + (size_t)B * (size_t)((F + 1) & ~1) * sizeof(unsigned);
}

// + the coarse bins of the binned rasteriser (per view: counts and lists of `cap` face indices per BIN x BIN-tile bin)
static int raster_bin_cap(int F) {
    const int Fp = (F + 1) & ~1;
    return (Fp < 8192 ? Fp : 8192) + 2;
}
extern "C" size_t st3d_raster_workspace_bytes_binned(int B, int F, int S) {
    const int nb = st3d::cdiv(st3d::cdiv(S, TILE), BIN);
    size_t base = (st3d_raster_workspace_bytes(B, F) + 15) & ~(size_t)15;
    // [counts: B*nb*nb ints, padded to an even number][lists: cap (even) ints per bin] -- the lists are read as int2
    return base + ((((size_t)B * nb * nb + 1) & ~(size_t)1) + (size_t)B * nb * nb * raster_bin_cap(F)) * sizeof(int);
}

extern "C" int st3d_face_setup(const float *verts_ndc, const int32_t *faces, int B, int V, int F, void *face_records,
                               size_t records_bytes, st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts_ndc && faces && face_records);
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && records_bytes >= st3d_raster_workspace_bytes(B, F));
    ST3D_CHECK_ARG(((uintptr_t)face_records & 15) == 0);
    face_setup_kernel<<<st3d::cdiv((long)B * F, 256), 256, 0, st3d::as_stream(stream)>>>(
        verts_ndc, faces, B, V, F, reinterpret_cast<float4 *>(face_records), 0, F, nullptr, 0.f, nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_raster_fwd(const float *verts_ndc, const int32_t *faces, int B, int V, int F, int S, void *workspace,
                               size_t workspace_bytes, int32_t *pix_to_face, float *zbuf, float *bary, float *dists,
                               float z_clip, int32_t *near_flag, st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts_ndc && faces && workspace && pix_to_face && zbuf && bary && dists);
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0 && S <= 4096);       // 8-bit tile indices in the packed ranges
    ST3D_CHECK_ARG(workspace_bytes >= st3d_raster_workspace_bytes(B, F));
    ST3D_CHECK_ARG(((uintptr_t)workspace & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    float4 *rec = reinterpret_cast<float4 *>(workspace);
    const int Fp = (F + 1) & ~1;
    unsigned *words = reinterpret_cast<unsigned *>(rec + (size_t)B * F * 3);
    face_setup_kernel<<<st3d::cdiv((long)B * Fp, 256), 256, 0, s>>>(verts_ndc, faces, B, V, F, rec, S, Fp, words, z_clip, near_flag);
    ST3D_LAUNCH_CHECK();
    const int tiles = st3d::cdiv(S, TILE);
    // coarse bins when the caller's workspace has room for them (st3d_raster_workspace_bytes_binned) and the mesh is big
    // enough for the all-faces sweep to matter; ST3D_RASTER_BINS=0 keeps the flat sweep (A/B runs)
    static const bool allow_bins = [] { const char *e = getenv("ST3D_RASTER_BINS"); return !(e && e[0] == '0'); }();
    const int nb = st3d::cdiv(tiles, BIN), cap = raster_bin_cap(F);
    if (allow_bins && Fp > 2048 && workspace_bytes >= st3d_raster_workspace_bytes_binned(B, F, S)) {
        const size_t base = (st3d_raster_workspace_bytes(B, F) + 15) & ~(size_t)15;
        int *bin_count = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + base);
        int *bin_list = bin_count + (((size_t)B * nb * nb + 1) & ~(size_t)1);      // 8-byte aligned: read as int2 (cap is even)
        raster_bin_kernel<<<dim3(nb * nb, B), 256, 0, s>>>(words, Fp, nb, cap, bin_count, bin_list);
        ST3D_LAUNCH_CHECK();
        raster_tile_kernel<<<dim3(tiles, tiles, B), 256, 0, s>>>(rec, words, F, Fp, S, pix_to_face, zbuf, bary, dists, bin_count,
                                                                bin_list, nb, cap);
    } else {
        raster_tile_kernel<<<dim3(tiles, tiles, B), 256, 0, s>>>(rec, words, F, Fp, S, pix_to_face, zbuf, bary, dists, nullptr,
                                                                nullptr, 0, 0);
    }
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

// ------------------------------------------------------------------------------------------
// Vertex path: backward of the perspective-correct barycentrics w.r.t. the
// projected vertices and of the projection w.r.t. the world vertices.  One thread per pixel,
// 9 float atomics per covered pixel into (B,V,3); the projection backward gathers over views
// per vertex (no atomics).  Same derivation as oracle/raster_ref.c:ref_raster_bwd.
namespace {

// One workgroup per 16x16-pixel tile.  A tile sees a handful of faces, so the nine per-pixel contributions are first
// summed per face in LDS (open-addressing table keyed by the face index, ds_add_f32) and only one set of nine global
// atomics per (tile, face) goes out -- ~20x fewer L2 atomics than one set per pixel, all of them contended.
constexpr int kBwdSlots = 512;          // > 256 pixels: the probe always terminates

// DET 0: float LDS table + float global atomics.  DET 1: bound pass of the deterministic variant (partials[block] = sum of
// the absolute contributions of the block's pixels).  DET 2: the same binning in 64-bit fixed point with the scale derived
// from that bound (det.h): `gndc` is then the int64 accumulator array.
template <int DET>
__global__ __launch_bounds__(256) void raster_bwd_kernel(const float *__restrict__ gbary, const int32_t *__restrict__ p2f,
                                                         const float *__restrict__ ndc, const int32_t *__restrict__ faces,
                                                         int B, int V, int S, int tiles_x, float *__restrict__ gndc,
                                                         const st3d_det::DetHeader *__restrict__ det, float *__restrict__ partials) {
    typedef typename std::conditional<DET == 2, unsigned long long, float>::type acc_t;
    __shared__ int s_key[kBwdSlots];
    __shared__ acc_t s_acc[DET == 1 ? 1 : kBwdSlots][9];
    __shared__ float s4[4];
    const int tid = threadIdx.x;
    if (DET != 1) {
        for (int e = tid; e < kBwdSlots; e += 256) s_key[e] = -1;
        for (int e = tid; e < kBwdSlots * 9; e += 256) (&s_acc[0][0])[e] = (acc_t)0;
        __syncthreads();
    }
    const double dscale = DET == 2 ? det->scale : 1.0;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int yi = ty * 16 + (tid >> 4), xi = tx * 16 + (tid & 15);
    const size_t HW = (size_t)S * S;
    const float *vb = ndc + (size_t)b * V * 3;
    int f = -1;
    size_t i = 0;
    float bound = 0.f;
    if (yi < S && xi < S) { i = (size_t)b * HW + (size_t)yi * S + xi; f = p2f[i]; }
    if (f >= 0) {
        const float px = pix_to_ndc(S - 1 - xi, S), py = pix_to_ndc(S - 1 - yi, S);
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const float x0 = vb[3 * i0], y0 = vb[3 * i0 + 1], z0 = vb[3 * i0 + 2];
        const float x1 = vb[3 * i1], y1 = vb[3 * i1 + 1], z1 = vb[3 * i1 + 2];
        const float x2 = vb[3 * i2], y2 = vb[3 * i2 + 1], z2 = vb[3 * i2 + 2];
        const float A = edge_fn(x2, y2, x0, y0, x1, y1) + kEps;
        const float w0 = edge_fn(px, py, x1, y1, x2, y2) / A;
        const float w1 = edge_fn(px, py, x2, y2, x0, y0) / A;
        const float w2 = edge_fn(px, py, x0, y0, x1, y1) / A;
        const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
        const float den = t0 + t1 + t2;
        if (den > kEps) {
            const float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
            const float g0 = gbary[3 * i], g1 = gbary[3 * i + 1], g2 = gbary[3 * i + 2];
            const float gs = g0 * b0 + g1 * b1 + g2 * b2;
            const float dt0 = (g0 - gs) / den, dt1 = (g1 - gs) / den, dt2 = (g2 - gs) / den;
            const float dw0 = dt0 * z1 * z2, dw1 = dt1 * z0 * z2, dw2 = dt2 * z0 * z1;
            const float dz0 = dt1 * w1 * z2 + dt2 * z1 * w2;
            const float dz1 = dt0 * w0 * z2 + dt2 * z0 * w2;
            const float dz2 = dt0 * w0 * z1 + dt1 * z0 * w1;
            const float de0 = dw0 / A, de1 = dw1 / A, de2 = dw2 / A;
            const float dA = -(dw0 * w0 + dw1 * w1 + dw2 * w2) / A;
            float c9[9];
            c9[0] = de1 * -(py - y2) + de2 * (py - y1) + dA * (y2 - y1);
            c9[1] = de1 * (px - x2) + de2 * (x1 - px) + dA * (x1 - x2);
            c9[2] = dz0;
            c9[3] = de0 * (py - y2) + de2 * -(py - y0) + dA * -(y2 - y0);
            c9[4] = de0 * (x2 - px) + de2 * (px - x0) + dA * (x2 - x0);
            c9[5] = dz1;
            c9[6] = de0 * -(py - y1) + de1 * (py - y0) + dA * (y1 - y0);
            c9[7] = de0 * (px - x1) + de1 * (x0 - px) + dA * -(x1 - x0);
            c9[8] = dz2;
            if (DET == 1) {
#pragma unroll
                for (int c = 0; c < 9; ++c) bound += fabsf(c9[c]);
            } else {
                int slot = (int)(((unsigned)f * 2654435761u) >> 23) & (kBwdSlots - 1);
                for (;;) {
                    const int prev = atomicCAS(&s_key[slot], -1, f);
                    if (prev == -1 || prev == f) break;
                    slot = (slot + 1) & (kBwdSlots - 1);
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    if (DET == 2) atomicAdd(reinterpret_cast<unsigned long long *>(&s_acc[slot][c]),
                                            (unsigned long long)st3d_det::det_quantise(c9[c], dscale));
                    else atomicAdd(reinterpret_cast<float *>(&s_acc[slot][c]), c9[c]);
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
    for (int e = tid; e < kBwdSlots * 9; e += 256) {
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

__global__ void project_verts_bwd_kernel(const float *__restrict__ verts, int V, const float *__restrict__ R,
                                         const float *__restrict__ T, int B, float s, const float *__restrict__ gndc,
                                         int accumulate, float *__restrict__ gverts) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *r = R + 9 * b, *t = T + 3 * b;
        const float xv = x * r[0] + y * r[3] + z * r[6] + t[0];
        const float yv = x * r[1] + y * r[4] + z * r[7] + t[1];
        const float zv = x * r[2] + y * r[5] + z * r[8] + t[2];
        const float *g = gndc + ((size_t)b * V + v) * 3;
        const float dxv = s * g[0] / zv, dyv = s * g[1] / zv;
        const float dzv = g[2] - s * (g[0] * xv + g[1] * yv) / (zv * zv);
        ax += dxv * r[0] + dyv * r[1] + dzv * r[2];
        ay += dxv * r[3] + dyv * r[4] + dzv * r[5];
        az += dxv * r[6] + dyv * r[7] + dzv * r[8];
    }
    if (accumulate) { ax += gverts[3 * v]; ay += gverts[3 * v + 1]; az += gverts[3 * v + 2]; }
    gverts[3 * v] = ax; gverts[3 * v + 1] = ay; gverts[3 * v + 2] = az;
}

}  // namespace

extern "C" int st3d_raster_bwd(const float *grad_bary, const int32_t *pix_to_face, const float *verts_ndc,
                               const int32_t *faces, int B, int V, int F, int S, float *grad_verts_ndc,
                               st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_bary && pix_to_face && verts_ndc && faces && grad_verts_ndc);
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    hipStream_t s = st3d::as_stream(stream);
    ST3D_HIP(hipMemsetAsync(grad_verts_ndc, 0, (size_t)B * V * 3 * sizeof(float), s));
    const int tiles = (S + 15) / 16;
    raster_bwd_kernel<0><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, pix_to_face, verts_ndc, faces, B, V, S, tiles,
                                                                grad_verts_ndc, nullptr, nullptr);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" size_t st3d_raster_bwd_det_workspace_bytes(int B, int V, int S) {
    const size_t tiles = (size_t)((S + 15) / 16);
    return st3d_det::workspace_bytes((size_t)B * V * 3, tiles * tiles * B);
}

// st3d_raster_bwd with a bitwise reproducible result (fixed-point accumulation, det.h): a first pass bounds the partial
// sums, the second accumulates; about twice the time of the float-atomic version.
extern "C" int st3d_raster_bwd_det(const float *grad_bary, const int32_t *pix_to_face, const float *verts_ndc,
                                   const int32_t *faces, int B, int V, int F, int S, float *grad_verts_ndc, void *workspace,
                                   size_t workspace_bytes, st3d_stream_t stream) {
    ST3D_CHECK_ARG(grad_bary && pix_to_face && verts_ndc && faces && grad_verts_ndc && workspace);
    ST3D_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_raster_bwd_det_workspace_bytes(B, V, S) && ((uintptr_t)workspace & 15) == 0);
    hipStream_t s = st3d::as_stream(stream);
    const int tiles = (S + 15) / 16;
    const size_t np = (size_t)tiles * tiles * B, nacc = (size_t)B * V * 3;
    auto *hdr = reinterpret_cast<st3d_det::DetHeader *>(workspace);
    float *partials = st3d_det::partials_of(workspace);
    long long *acc = st3d_det::accum_of(workspace, np);
    raster_bwd_kernel<1><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, pix_to_face, verts_ndc, faces, B, V, S, tiles, nullptr,
                                                                nullptr, partials);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_scale_kernel<<<1, 256, 0, s>>>(partials, (int)np, hdr);
    ST3D_LAUNCH_CHECK();
    ST3D_HIP(hipMemsetAsync(acc, 0, nacc * sizeof(long long), s));
    raster_bwd_kernel<2><<<dim3(tiles * tiles, B), 256, 0, s>>>(grad_bary, pix_to_face, verts_ndc, faces, B, V, S, tiles,
                                                                reinterpret_cast<float *>(acc), hdr, nullptr);
    ST3D_LAUNCH_CHECK();
    st3d_det::det_convert_kernel<<<st3d::cdiv((long)nacc, 256), 256, 0, s>>>(acc, nacc, hdr, 0, grad_verts_ndc);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_project_verts_bwd(const float *verts, int V, const float *R, const float *T, int B,
                                      float inv_tan_half_fov, const float *grad_verts_ndc, int accumulate, float *grad_verts,
                                      st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts && R && T && grad_verts_ndc && grad_verts);
    ST3D_CHECK_ARG(V > 0 && B > 0);
    project_verts_bwd_kernel<<<st3d::cdiv(V, 256), 256, 0, st3d::as_stream(stream)>>>(verts, V, R, T, B, inv_tan_half_fov,
                                                                                      grad_verts_ndc, accumulate, grad_verts);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
