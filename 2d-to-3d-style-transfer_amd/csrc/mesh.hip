// mesh.hip -- view-independent mesh regularisers of optimization_target 'mesh'/'both'
// (reference losses.py:84-87,93-96,112-115,121-124: F.mse_loss(verts, target_verts) +
// pytorch3d.loss mesh_edge_loss + mesh_laplacian_smoothing('uniform') + mesh_normal_consistency),
// forward and gradient for ONE mesh (kernel K15).  Topology is static: the host
// builds the unique edge list, the CSR vertex adjacency and the list of face pairs once.
// O(V + E + P) work, a few tens of KB: latency-bound; losses are reduced in fixed order
// (per-workgroup partials, then one workgroup) so they are bitwise reproducible; so are the
// gradients: every term's gradient is a per-vertex GATHER over static CSR lists in a fixed order
// (round 3: the edge term walks the vertex adjacency, the normal term the vertex -> (pair, role)
// list over per-pair gradients staged in scratch) -- no float atomics anywhere.
#include "common.h"

namespace {

constexpr int NPART = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ void block_partial(float v, float *partials) {
    __shared__ float s[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float *p, int i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ void st3(float *p, int i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// mse(verts, target): mean over V*3; grad = 2 w (v - t) / (3V)
__global__ __launch_bounds__(256) void verts_mse_kernel(const float *__restrict__ v, const float *__restrict__ t, int n,
                                                        float gcoef, float *__restrict__ g, float *__restrict__ partials) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float d = v[i] - t[i];
        acc += d * d;
        if (gcoef != 0.f) g[i] += gcoef * d;     // one thread per element: no race
    }
    block_partial(acc, partials);
}

// mean over edges of |v0 - v1|^2 (target length 0)
__global__ __launch_bounds__(256) void edge_kernel(const float *__restrict__ v, const int32_t *__restrict__ edges, int E,
                                                   float *__restrict__ partials) {
    float acc = 0.f;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int a = edges[2 * e], b = edges[2 * e + 1];
        const V3 d = sub(ld3(v, a), ld3(v, b));
        acc += dot(d, d);
    }
    block_partial(acc, partials);
}

// edge gradient as a gather: every unique edge (a, b) is listed once in a's and once in b's adjacency row, and
// d|a-b|^2/da = 2 (a - b), so grad_k += gcoef * sum_{j in N(k)} (v_k - v_j) in the row's (ascending j) order
__global__ __launch_bounds__(256) void edge_bwd_kernel(const float *__restrict__ v, const int32_t *__restrict__ off,
                                                       const int32_t *__restrict__ nbr, int V, float gcoef,
                                                       float *__restrict__ g) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= V) return;
    const V3 vk = ld3(v, k);
    V3 s = {0.f, 0.f, 0.f};
    for (int q = off[k]; q < off[k + 1]; ++q) s = add(s, sub(vk, ld3(v, nbr[q])));
    g[3 * k] += gcoef * s.x; g[3 * k + 1] += gcoef * s.y; g[3 * k + 2] += gcoef * s.z;
}

// y_i = mean_{j in N(i)} v_j - v_i ; loss += |y_i| ; u_i = y_i / |y_i| (0 at |y_i| = 0)
__global__ __launch_bounds__(256) void laplacian_fwd_kernel(const float *__restrict__ v, const int32_t *__restrict__ off,
                                                            const int32_t *__restrict__ nbr, int V, float *__restrict__ u,
                                                            float *__restrict__ partials) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        const int b = off[i], e = off[i + 1];
        V3 y = {0.f, 0.f, 0.f};
        if (e > b) {
            V3 s = {0.f, 0.f, 0.f};
            for (int k = b; k < e; ++k) s = add(s, ld3(v, nbr[k]));
            y = sub(mul(s, 1.0f / (float)(e - b)), ld3(v, i));
        }
        const float n = sqrtf(dot(y, y));
        acc += n;
        const V3 ui = n > 0.f ? mul(y, 1.0f / n) : V3{0.f, 0.f, 0.f};
        u[3 * i] = ui.x; u[3 * i + 1] = ui.y; u[3 * i + 2] = ui.z;
    }
    block_partial(acc, partials);
}

// grad_k += c * ( sum_{i in N(k)} u_i / deg_i - u_k )      (adjacency is symmetric)
__global__ __launch_bounds__(256) void laplacian_bwd_kernel(const float *__restrict__ u, const int32_t *__restrict__ off,
                                                            const int32_t *__restrict__ nbr, int V, float c,
                                                            float *__restrict__ g) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= V) return;
    V3 s = {0.f, 0.f, 0.f};
    for (int q = off[k]; q < off[k + 1]; ++q) {
        const int i = nbr[q];
        const int deg = off[i + 1] - off[i];
        s = add(s, mul(ld3(u, i), 1.0f / (float)deg));
    }
    if (off[k + 1] > off[k]) s = sub(s, ld3(u, k));
    g[3 * k] += c * s.x; g[3 * k + 1] += c * s.y; g[3 * k + 2] += c * s.z;
}

// pairs (v0, v1, a, b): n0 = (v1-v0) x (a-v0), n1 = -(v1-v0) x (b-v0); loss = 1 - cos(n0, n1)
__global__ __launch_bounds__(256) void normal_kernel(const float *__restrict__ v, const int32_t *__restrict__ pairs, int P,
                                                     float gcoef, float *__restrict__ pg, float *__restrict__ partials) {
    float acc = 0.f;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int i0 = pairs[4 * p], i1 = pairs[4 * p + 1], ia = pairs[4 * p + 2], ib = pairs[4 * p + 3];
        const V3 v0 = ld3(v, i0);
        const V3 e = sub(ld3(v, i1), v0), pa = sub(ld3(v, ia), v0), pb = sub(ld3(v, ib), v0);
        const V3 n0 = cross(e, pa), n1 = mul(cross(e, pb), -1.0f);
        const float eps = 1e-8f;
        const float l0 = fmaxf(sqrtf(dot(n0, n0)), eps), l1 = fmaxf(sqrtf(dot(n1, n1)), eps);
        const float c = dot(n0, n1) / (l0 * l1);
        acc += 1.0f - c;
        if (gcoef != 0.f) {
            // d(1-c)/dn0 = -(n1/(l0 l1) - c n0/l0^2), same for n1
            const V3 g0 = mul(sub(mul(n1, 1.0f / (l0 * l1)), mul(n0, c / (l0 * l0))), -gcoef);
            const V3 g1 = mul(sub(mul(n0, 1.0f / (l0 * l1)), mul(n1, c / (l1 * l1))), -gcoef);
            // n0 = e x pa: de = pa x g0, dpa = g0 x e ; n1 = -(e x pb): de -= pb x g1, dpb = -(g1 x e)
            const V3 de = sub(cross(pa, g0), cross(pb, g1));
            const V3 dpa = cross(g0, e);
            const V3 dpb = mul(cross(g1, e), -1.0f);
            // staged per (pair, role) -- role = column of the pair row (v0, v1, a, b); normal_gather_kernel sums them per vertex
            st3(pg, 4 * p + 0, mul(add(add(de, dpa), dpb), -1.0f));
            st3(pg, 4 * p + 1, de);
            st3(pg, 4 * p + 2, dpa);
            st3(pg, 4 * p + 3, dpb);
        }
    }
    block_partial(acc, partials);
}

// grad_k += sum over the (pair, role) entries of vertex k, in list order (static, built by the host: ascending pair * 4 + role)
__global__ __launch_bounds__(256) void normal_gather_kernel(const float *__restrict__ pg, const int32_t *__restrict__ off,
                                                            const int32_t *__restrict__ ref, int V, float *__restrict__ g) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= V) return;
    V3 s = {0.f, 0.f, 0.f};
    for (int q = off[k]; q < off[k + 1]; ++q) s = add(s, ld3(pg, ref[q]));
    g[3 * k] += s.x; g[3 * k + 1] += s.y; g[3 * k + 2] += s.z;
}

// loss_out = {sum_k w_k * term_k, mse, edge, laplacian, normal}
__global__ __launch_bounds__(256) void mesh_finish_kernel(const float *__restrict__ partials, int n0, int n1, int n2, int n3,
                                                          float s0, float s1, float s2, float s3, float w0, float w1,
                                                          float w2, float w3, float *__restrict__ out) {
    __shared__ double red[256];
    const int cnt[4] = {n0, n1, n2, n3};
    const float sc[4] = {s0, s1, s2, s3};
    const float w[4] = {w0, w1, w2, w3};
    double total = 0.0;
    for (int k = 0; k < 4; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < cnt[k]; i += 256) a += (double)partials[k * NPART + i];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const double term = red[0] * (double)sc[k];
        if (threadIdx.x == 0) out[1 + k] = (float)term;
        total += (double)w[k] * term;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}

inline int grid_for(int n) {
    int b = (n + 255) / 256;
    return b < 1 ? 1 : (b > NPART ? NPART : b);
}

}  // namespace

extern "C" size_t st3d_mesh_reg_scratch_floats(int V, int P) { return (size_t)3 * V + (size_t)12 * P; }

extern "C" int st3d_mesh_reg(const float *verts, const float *target_verts, int V, const int32_t *edges, int E,
                             const int32_t *nbr_off, const int32_t *nbr_idx, const int32_t *pairs, int P,
                             const int32_t *pair_off, const int32_t *pair_ref,
                             const float *weights, float *scratch, float *partials, float *loss_out, float *grad_verts,
                             st3d_stream_t stream) {
    ST3D_CHECK_ARG(verts && target_verts && edges && nbr_off && nbr_idx && weights && scratch && partials && loss_out);
    ST3D_CHECK_ARG(V > 0 && E > 0 && P >= 0 && (P == 0 || (pairs && pair_off && pair_ref)));
    hipStream_t s = st3d::as_stream(stream);
    const float wv = weights[0], we = weights[1], wl = weights[2], wn = weights[3];
    const bool wg = grad_verts != nullptr;
    const int g0 = grid_for(3 * V), g1 = grid_for(E), g2 = grid_for(V), g3 = P > 0 ? grid_for(P) : 0;
    verts_mse_kernel<<<g0, 256, 0, s>>>(verts, target_verts, 3 * V, wg ? 2.0f * wv / (3.0f * V) : 0.f, grad_verts, partials);
    ST3D_LAUNCH_CHECK();
    edge_kernel<<<g1, 256, 0, s>>>(verts, edges, E, partials + NPART);
    ST3D_LAUNCH_CHECK();
    if (wg && we != 0.f) {
        edge_bwd_kernel<<<st3d::cdiv(V, 256), 256, 0, s>>>(verts, nbr_off, nbr_idx, V, 2.0f * we / (float)E, grad_verts);
        ST3D_LAUNCH_CHECK();
    }
    laplacian_fwd_kernel<<<g2, 256, 0, s>>>(verts, nbr_off, nbr_idx, V, scratch, partials + 2 * NPART);
    ST3D_LAUNCH_CHECK();
    if (wg && wl != 0.f) {
        laplacian_bwd_kernel<<<st3d::cdiv(V, 256), 256, 0, s>>>(scratch, nbr_off, nbr_idx, V, wl / (float)V, grad_verts);
        ST3D_LAUNCH_CHECK();
    }
    if (P > 0) {
        float *pg = scratch + (size_t)3 * V;        // (P, 4, 3) per-pair gradients
        const bool ng = wg && wn != 0.f;
        normal_kernel<<<g3, 256, 0, s>>>(verts, pairs, P, ng ? wn / (float)P : 0.f, pg, partials + 3 * NPART);
        ST3D_LAUNCH_CHECK();
        if (ng) {
            normal_gather_kernel<<<st3d::cdiv(V, 256), 256, 0, s>>>(pg, pair_off, pair_ref, V, grad_verts);
            ST3D_LAUNCH_CHECK();
        }
    }
    mesh_finish_kernel<<<1, 256, 0, s>>>(partials, g0, g1, g2, g3, 1.0f / (3.0f * V), 1.0f / (float)E, 1.0f / (float)V,
                                        P > 0 ? 1.0f / (float)P : 0.f, wv, we, wl, wn, loss_out);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
