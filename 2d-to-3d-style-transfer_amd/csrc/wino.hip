// wino.hip -- conv3x3(pad 1) forward / input-gradient as Winograd F(2x2,3x3) on the gfx950 fp32
// matrix pipe: 16 multiplies per 2x2 output tile instead of 36 (2.25x fewer MFMA flops than the
// direct implicit GEMM of conv.hip), still exact-fp32 products and fp32 accumulation.
// Used for every VGG layer with Cin, Cout multiples of 64 (all but conv1_1), both directions.
//
//   Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A        per (cout, 2x2 tile)
//   U[xi][cin][cout] = G g G^T   -- precomputed once (weights are frozen, utils.py:50-51)
//   V[xi][cin][tile] = B^T d B   -- computed on the fly from the haloed input patch in LDS
//   M[xi] = U[xi]^T V[xi]        -- 16 independent GEMMs (M = cout, N = tile, K = cin) on
//                                   v_mfma_f32_32x32x2_f32
//
// Mapping: one 512-thread workgroup = 64 cout x 64 tiles (8 rows x 32 columns of pixels).
// 8 waves = 4 (row a of the 4x4 Winograd domain: xi = 4a..4a+3) x 2 (32-cout halves); every wave
// covers all 64 tiles: 8 MFMA tiles = 128 accumulator VGPRs, two waves per SIMD (they hide each
// other's LDS-operand latency and barrier bubbles).  The output transform A^T M A runs along b in
// registers and along a through ONE [4][64][64] LDS exchange per output column; after it a lane
// owns a whole 2x2 output tile, which IS the MaxPool2d(2,2) window, so pooling (+argmax) fuses
// into the epilogue for free and the stores are one cout row (64 tiles) per wave instruction.
// K loop: chunks of 4 input channels; per chunk U [16][4][64] (16 KB) and the patch
// [4][10][34] are staged global->registers->LDS two chunks ahead (triple-buffered), the
// input transform of chunk c+1 (patch -> V, one (tile, channel, row-half) per thread,
// ds_read_b64 / ds_write_b32 conflict-free) runs beside the MFMAs of chunk c (double-buffered
// V): one barrier per chunk.  The ReLU gate / 2x2 max-unpool of the backward pass are fused into
// the patch load exactly as in conv.hip.  The cout-tile index is the fastest grid dimension so
// the workgroups of one XCD (dispatch is round-robin over the 8 XCDs) stream the same U slice
// out of that XCD's L2.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 4;                 // input channels per chunk
constexpr int BCO = 64;               // cout per workgroup
constexpr int TROWS = 8, TCOLS = 32;  // output pixels per workgroup
constexpr int NT = 512;               // threads per workgroup (8 waves, 2 per SIMD)
constexpr int PR = TROWS + 2, PC = TCOLS + 2, PS = PR * PC;   // haloed patch per channel
constexpr int P_ELEMS = KC * PS;                               // 1360
constexpr int P_PER_T = (P_ELEMS + NT - 1) / NT;               // 3
constexpr int P_PAD = P_PER_T * NT;                            // 1536
constexpr int U_ELEMS = 16 * KC * BCO;                         // 4096 floats = 1024 float4 = 2 per thread
constexpr int V_ELEMS = 16 * KC * 64;                          // 4096
constexpr int SMEM_FLOATS = 3 * U_ELEMS + 3 * P_PAD + 2 * V_ELEMS;   // 25088 floats = 98 KB
static_assert(SMEM_FLOATS >= 4 * 64 * 64, "the epilogue exchange needs [4][64][64] floats");

struct WinoArgs {
    const float *x;       // MODE 0/1: (N,Cin,H,W); MODE 2: pooled-resolution gradient (N,Cin,H/2,W/2)
    const float *aux;     // MODE 1: saved post-ReLU activation; MODE 2: pooled values
    const uint8_t *idx;   // MODE 2: pool argmax
    const float *U;       // [16][Cin][Cout]
    const float *bias;    // (Cout) or nullptr
    float *y;             // (N,Cout,H,W) or nullptr (EPI 1 may skip the full-resolution store)
    float *yp;            // EPI 1: pooled output (N,Cout,H/2,W/2)
    uint8_t *yidx;        // EPI 1: argmax
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_y, n_ct;
};

// Wave roles (8 waves): a = wave & 3 is the row of the 4x4 Winograd domain the wave accumulates
// (xi = 4a .. 4a+3), mh = wave >> 2 the 32-cout half; every wave covers all 64 tiles (2 MFMA
// n-tiles).  8 MFMA tiles = 128 accumulator VGPRs per wave -> two waves per SIMD, which hides the
// LDS-operand latency and the barrier bubbles of each other.  Per k-step a wave reads 4 A + 8 B
// dwords for 8 MFMAs.  The output transform runs along b inside the wave (registers) and along a
// across the four a-waves through one [4][64][64] LDS exchange per output column j.
template <int MODE, int EPI>
__global__ __launch_bounds__(NT, 2) void wino_kernel(const WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *sU = smem;                          // [3][U_ELEMS]
    float *sP = smem + 3 * U_ELEMS;            // [3][P_PAD]
    float *sV = sP + 3 * P_PAD;                // [2][V_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wa = wave & 3, mh = wave >> 2;

    // grid: x = cout tile (fastest) + n_ct * (pixel tile + tiles * image)
    int bid = blockIdx.x;
    const int ct = bid % a.n_ct; bid /= a.n_ct;
    const int tile_x = bid % a.tiles_x; bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int x0 = tile_x * TCOLS, y0 = tile_y * TROWS, co0 = ct * BCO;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    const size_t in_plane = (MODE == 2) ? (size_t)Hp * Wp : HW;

    // ---- per-thread patch coordinates
    int xoff[P_PER_T];
    unsigned xvalid = 0;
    unsigned xpos[MODE == 2 ? P_PER_T : 1];
#pragma unroll
    for (int i = 0; i < P_PER_T; ++i) {
        const int e = tid + i * NT;
        const int ci = e / PS, rem = e - ci * PS;
        const int r = rem / PC, cc = rem - r * PC;
        const int gy = y0 + r - 1, gx = x0 + cc - 1;
        const bool ok = (e < P_ELEMS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (ok) xvalid |= 1u << i;
        if (MODE == 2) {
            xoff[i] = ok ? (int)(ci * in_plane + (size_t)(gy >> 1) * Wp + (gx >> 1)) : 0;
            xpos[i] = ((gy & 1) << 1) | (gx & 1);
        } else {
            xoff[i] = ok ? (int)(ci * in_plane + (size_t)gy * W + gx) : 0;
        }
    }
    const float *xin = a.x + (size_t)n * a.Cin * in_plane;
    const float *auxin = (MODE != 0) ? a.aux + (size_t)n * a.Cin * in_plane : nullptr;
    const uint8_t *idxin = (MODE == 2) ? a.idx + (size_t)n * a.Cin * in_plane : nullptr;

    float xv[P_PER_T];
    float xa[MODE != 0 ? P_PER_T : 1];
    unsigned char xi[MODE == 2 ? P_PER_T : 1];
    f32x4 uv[2];

    auto gload = [&](int c) __attribute__((always_inline)) {
        const int ci0 = c * KC;
        const size_t cbase = (size_t)ci0 * in_plane;
#pragma unroll
        for (int i = 0; i < P_PER_T; ++i) {
            const size_t o = ((xvalid >> i) & 1u) ? cbase + xoff[i] : 0;
            xv[i] = xin[o];
            if (MODE != 0) xa[i] = auxin[o];
            if (MODE == 2) xi[i] = idxin[o];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e4 = tid + i * NT, row = e4 >> 4, c4 = e4 & 15;
            const int xi_ = row >> 2, k = row & 3;
            uv[i] = *reinterpret_cast<const f32x4 *>(a.U + ((size_t)(xi_ * a.Cin + ci0 + k) * a.Cout + co0 + c4 * 4));
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < P_PER_T; ++i) {
            bool ok = (xvalid >> i) & 1u;
            if (MODE != 0) ok = ok && (xa[i] > 0.f);
            if (MODE == 2) ok = ok && (xi[i] == xpos[i]);
            sP[buf * P_PAD + tid + i * NT] = ok ? xv[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(&sU[buf * U_ELEMS + (tid + i * NT) * 4]) = uv[i];
    };
    // input transform V = B^T d B: thread = (channel k = wave & 3, rows i in {2*rh, 2*rh+1} with
    // rh = wave >> 2, tile t = lane)
    const int tk = wave & 3, rh = wave >> 2;
    const int tty = lane >> 4, ttx = lane & 15;
    auto transform = [&](int pbuf, int vbuf) __attribute__((always_inline)) {
        // rows d[rh], d[rh+1], d[rh+2] of the 4x4 patch (rh = 0: d0,d1,d2; rh = 1: d1,d2,d3)
        const float *p = &sP[pbuf * P_PAD + tk * PS + (2 * tty + rh) * PC + 2 * ttx];
        float d[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f32x2 lo = *reinterpret_cast<const f32x2 *>(p + r * PC);
            const f32x2 hi = *reinterpret_cast<const f32x2 *>(p + r * PC + 2);
            d[r][0] = lo[0]; d[r][1] = lo[1]; d[r][2] = hi[0]; d[r][3] = hi[1];
        }
        float t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // rh = 0: t0 = d0 - d2, t1 = d1 + d2 ; rh = 1: t2 = d2 - d1, t3 = d1 - d3 (d1,d2,d3 = d[0],d[1],d[2])
            t[0][j] = rh == 0 ? d[0][j] - d[2][j] : d[1][j] - d[0][j];
            t[1][j] = rh == 0 ? d[1][j] + d[2][j] : d[0][j] - d[2][j];
        }
        float *v = &sV[vbuf * V_ELEMS + tk * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 2 * rh + i;
            v[((row * 4 + 0) * KC) * 64] = t[i][0] - t[i][2];
            v[((row * 4 + 1) * KC) * 64] = t[i][1] + t[i][2];
            v[((row * 4 + 2) * KC) * 64] = t[i][2] - t[i][1];
            v[((row * 4 + 3) * KC) * 64] = t[i][1] - t[i][3];
        }
    };

    f32x16 acc[4][2];      // [b][n]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][nn][r] = 0.f;

    const int nchunks = a.Cin / KC;
    gload(0);
    lstore(0);
    __syncthreads();
    if (nchunks > 1) gload(1);
    transform(0, 0);
    if (nchunks > 1) lstore(1);
    __syncthreads();

    int ub = 0, vb = 0;       // U/patch buffer of chunk c (mod 3), V buffer of chunk c (mod 2)
    for (int c = 0; c < nchunks; ++c) {
        const int ub1 = (ub == 2) ? 0 : ub + 1, ub2 = (ub1 == 2) ? 0 : ub1 + 1;
        if (c + 2 < nchunks) gload(c + 2);
        const float *pu = &sU[ub * U_ELEMS + ((wa * 4) * KC + lhi) * 64 + mh * 32 + l31];
        const float *pv = &sV[vb * V_ELEMS + ((wa * 4) * KC + lhi) * 64 + l31];
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float av = pu[(q * KC + ks * 2) * 64];
                const float b0 = pv[(q * KC + ks * 2) * 64];
                const float b1 = pv[(q * KC + ks * 2) * 64 + 32];
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[q][1], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) transform(ub1, vb ^ 1);
        if (c + 2 < nchunks) lstore(ub2);
        __syncthreads();
        ub = ub1; vb ^= 1;
    }

    // ---- epilogue: Y = A^T M A.  Along b in registers (z_j), along a through LDS.
    //   z_0 = m_a0 + m_a1 + m_a2 ; z_1 = m_a1 - m_a2 - m_a3      (this wave's row a)
    //   y_0j = z_j(a=0) + z_j(a=1) + z_j(a=2) ; y_1j = z_j(a=1) - z_j(a=2) - z_j(a=3)
    float *ex = smem;      // [4 a][64 co][64 tiles]; the last main-loop barrier has been passed
    float yv[8][2][2];     // per thread: 8 (co, tile) elements x 2x2 outputs
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1) __syncthreads();           // pass-0 reads done before the buffer is rewritten
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = j == 0 ? (acc[0][nn][r] + acc[1][nn][r] + acc[2][nn][r])
                                       : (acc[1][nn][r] - acc[2][nn][r] - acc[3][nn][r]);
                const int co = mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                ex[(wa * 64 + co) * 64 + nn * 32 + l31] = z;
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = e * 8 + wave;
            const float z0 = ex[(0 * 64 + co) * 64 + lane], z1 = ex[(1 * 64 + co) * 64 + lane];
            const float z2 = ex[(2 * 64 + co) * 64 + lane], z3 = ex[(3 * 64 + co) * 64 + lane];
            yv[e][0][j] = z0 + z1 + z2;
            yv[e][1][j] = z1 - z2 - z3;
        }
    }
    const int oy = y0 + 2 * (lane >> 4), ox = x0 + 2 * (lane & 15);
    const bool inb = oy < H && ox < W;     // H, W even: the whole 2x2 tile is inside or outside
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = co0 + e * 8 + wave;
        const float bsum = a.bias ? a.bias[co] : 0.f;
        float y00 = yv[e][0][0] + bsum, y01 = yv[e][0][1] + bsum, y10 = yv[e][1][0] + bsum, y11 = yv[e][1][1] + bsum;
        if (a.relu) {
            y00 = y00 > 0.f ? y00 : 0.f; y01 = y01 > 0.f ? y01 : 0.f;
            y10 = y10 > 0.f ? y10 : 0.f; y11 = y11 > 0.f ? y11 : 0.f;
        }
        if (!inb) continue;
        if (a.y) {
            float *dst = a.y + ((size_t)n * a.Cout + co) * HW + (size_t)oy * W + ox;
            f32x2 r0, r1;
            r0[0] = y00; r0[1] = y01; r1[0] = y10; r1[1] = y11;
            *reinterpret_cast<f32x2 *>(dst) = r0;
            *reinterpret_cast<f32x2 *>(dst + W) = r1;
        }
        if (EPI == 1) {     // MaxPool2d(2,2): first maximum in row-major window order (ATen)
            float best = y00; int bi = 0;
            if (y01 > best || y01 != y01) { best = y01; bi = 1; }
            if (y10 > best || y10 != y10) { best = y10; bi = 2; }
            if (y11 > best || y11 != y11) { best = y11; bi = 3; }
            const size_t po = ((size_t)n * a.Cout + co) * (size_t)Hp * Wp + (size_t)(oy >> 1) * Wp + (ox >> 1);
            a.yp[po] = best;
            if (a.yidx) a.yidx[po] = (uint8_t)bi;
        }
    }
}

// w (Cout,Cin,3,3) -> U_fwd [16][Cin][Cout] = G g G^T and U_dgrad [16][Cout][Cin] = G g' G^T with
// g'[ky][kx] = w[co][ci][2-ky][2-kx] (transposed convolution); computed in fp64, stored fp32.
__global__ void wino_pack_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ uf,
                                 float *__restrict__ ud) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * Cin) return;
    const int co = i / Cin, ci = i % Cin;
    const float *g = w + i * 9;
    for (int dir = 0; dir < 2; ++dir) {
        float *out = dir == 0 ? uf : ud;
        if (!out) continue;
        double gg[3][3];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) gg[ky][kx] = dir == 0 ? g[ky * 3 + kx] : g[(2 - ky) * 3 + (2 - kx)];
        double t[4][3];
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = gg[0][kx];
            t[1][kx] = 0.5 * (gg[0][kx] + gg[1][kx] + gg[2][kx]);
            t[2][kx] = 0.5 * (gg[0][kx] - gg[1][kx] + gg[2][kx]);
            t[3][kx] = gg[2][kx];
        }
        for (int aa = 0; aa < 4; ++aa) {
            const double u0 = t[aa][0], u1 = 0.5 * (t[aa][0] + t[aa][1] + t[aa][2]),
                         u2 = 0.5 * (t[aa][0] - t[aa][1] + t[aa][2]), u3 = t[aa][2];
            const double u[4] = {u0, u1, u2, u3};
            for (int bb = 0; bb < 4; ++bb) {
                const int xi = aa * 4 + bb;
                const size_t o = dir == 0 ? ((size_t)xi * Cin + ci) * Cout + co : ((size_t)xi * Cout + co) * Cin + ci;
                out[o] = (float)u[bb];
            }
        }
    }
}

template <int MODE>
int launch_wino(WinoArgs a, hipStream_t s) {
    a.tiles_x = st3d::cdiv(a.W, TCOLS);
    a.tiles_y = st3d::cdiv(a.H, TROWS);
    a.n_ct = a.Cout / BCO;
    const long blocks = (long)a.n_ct * a.tiles_x * a.tiles_y * a.N;
    if (a.yp) wino_kernel<MODE, 1><<<(unsigned)blocks, NT, 0, s>>>(a);
    else wino_kernel<MODE, 0><<<(unsigned)blocks, NT, 0, s>>>(a);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

bool shape_ok(int Cin, int Cout, int H, int W) {
    return Cin >= 64 && (Cin % 4) == 0 && (Cout % 64) == 0 && (H % 2) == 0 && (W % 2) == 0;
}

}  // namespace

extern "C" int st3d_wino_supported(int Cin, int Cout, int H, int W) { return shape_ok(Cin, Cout, H, W) ? 1 : 0; }

extern "C" size_t st3d_wino_packed_floats(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

extern "C" int st3d_wino_pack(const float *w, int Cout, int Cin, float *u_fwd, float *u_dgrad, st3d_stream_t stream) {
    ST3D_CHECK_ARG(w && (u_fwd || u_dgrad));
    ST3D_CHECK_ARG(Cout > 0 && Cin > 0);
    wino_pack_kernel<<<st3d::cdiv((long)Cout * Cin, 256), 256, 0, st3d::as_stream(stream)>>>(w, Cout, Cin, u_fwd, u_dgrad);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_wino_fwd(const float *x, const float *u_fwd, const float *bias, float *y, float *y_pooled,
                             uint8_t *pool_idx, int N, int Cin, int Cout, int H, int W, int relu, st3d_stream_t stream) {
    ST3D_CHECK_ARG(x && u_fwd && (y || y_pooled));
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cin, Cout, H, W));
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)u_fwd & 15) == 0);
    WinoArgs a{x, nullptr, nullptr, u_fwd, bias, y, y_pooled, pool_idx, N, Cin, Cout, H, W, relu, 0, 0, 0};
    return launch_wino<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_wino_dgrad(const float *gy, const float *act, const float *u_dgrad, float *gx, int N, int Cin, int Cout,
                               int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cout, Cin, H, W));
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0);
    WinoArgs a{gy, act, nullptr, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0};
    return act ? launch_wino<1>(a, st3d::as_stream(stream)) : launch_wino<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_wino_dgrad_unpool(const float *gy_pooled, const uint8_t *pool_idx, const float *pooled,
                                      const float *u_dgrad, float *gx, int N, int Cin, int Cout, int H, int W,
                                      st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy_pooled && pool_idx && pooled && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cout, Cin, H, W));
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0);
    WinoArgs a{gy_pooled, pooled, pool_idx, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0};
    return launch_wino<2>(a, st3d::as_stream(stream));
}
