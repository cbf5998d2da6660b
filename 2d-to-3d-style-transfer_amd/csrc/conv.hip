// conv.hip -- VGG-19 conv3x3(pad 1)+bias+ReLU forward and input-gradient ("dgrad") as an
// implicit GEMM on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32
// multiply-accumulate, 64 FLOP/clk/SIMD), plus MaxPool2d(2,2).
// Replaces the cuDNN/MIOpen calls behind `x = layer(x)` in style_transfer.py:23-24 for the
// modules of utils.py:49 (torchvision vgg19().features) and their autograd backward
// (second_approach.py:188).  Weights are frozen (utils.py:50-51): there is no wgrad.
//
// Mapping (one 256-thread workgroup = 4 waves, one wave per SIMD):
//   GEMM M = output channels, N = 32 consecutive pixels of one image row, K = (tap, cin).
//   Per MFMA: A[i][k] = W[tap][cin0+k][co0+i], B[k][j] = X[cin0+k][y+ky-1][x0+j+kx-1].
//   Workgroup tile = BM output channels x TH rows x 32 columns; each wave owns MT x 4 MFMA
//   tiles (MT*32 channels x 4 rows), i.e. up to 128 accumulator VGPRs.
//   K loop: chunks of KC=4 input channels; per chunk the weights [9][4][BM] and the haloed
//   input patch [4][TH+2][34] are staged in LDS (double-buffered; the next chunk's global
//   loads are issued before the 9*2*MT*4 MFMAs of the current one and written to the other
//   buffer after them: one barrier per chunk).  LDS reads are ds_read_b32 with the 32 lanes of
//   a half-wave on 32 consecutive words (conflict-free for both operands).
//   Weights are pre-packed once ([tap][cin][cout], zero-padded) so the A operand is
//   lane-contiguous; dgrad is the same kernel on the 180-degree-rotated, channel-transposed
//   pack, with the ReLU gate (and the 2x2 max-unpool) fused into the input-patch load.
// MFMA-bound: 2*9*Cin*Cout flops per output pixel; operands re-read from L2/LDS.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 4;    // input channels per K chunk
constexpr int TW = 32;   // pixel columns per tile (= MFMA N)
constexpr int XC = TW + 2;

struct ConvArgs {
    const float *x;      // MODE 0/1: (N,Cin,H,W); MODE 2: pooled-resolution gradient (N,Cin,H/2,W/2)
    const float *aux;    // MODE 1: saved post-ReLU activation (N,Cin,H,W); MODE 2: pooled values
    const uint8_t *idx;  // MODE 2: pool argmax
    const float *w;      // packed [9][CinP][CoutP]
    const float *bias;   // (Cout) or nullptr
    float *y;            // (N,Cout,H,W)
    int N, Cin, Cout, H, W, CinP, CoutP, relu, tiles_x;
};

template <int WAVES_M, int MT, int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const ConvArgs a) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int NT = 4;
    constexpr int TH = WAVES_N * NT;
    constexpr int XR = TH + 2;
    constexpr int XS = XR * XC;
    constexpr int X_ELEMS = KC * XS;
    constexpr int W_ELEMS = 9 * KC * BM;
    constexpr int X_PER_T = (X_ELEMS + 255) / 256;
    constexpr int W4_PER_T = (W_ELEMS / 4 + 255) / 256;
    // LDS regions are padded to a whole number of per-thread items so staging needs no
    // per-item predicate (a predicated float4 register array ends up in scratch)
    constexpr int W_PAD = W4_PER_T * 256 * 4;
    constexpr int X_PAD = X_PER_T * 256;
    constexpr int STAGE = W_PAD + X_PAD;

    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tile_x = blockIdx.x % a.tiles_x, tile_y = blockIdx.x / a.tiles_x;
    const int x0 = tile_x * TW, y0 = tile_y * TH;
    const int co0 = blockIdx.y * BM;
    const int n = blockIdx.z;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // ---- per-thread input-patch coordinates (constant over the K loop)
    int xoff[X_PER_T];
    unsigned xvalid = 0;
    unsigned xpos[MODE == 2 ? X_PER_T : 1];
    const int Hp = H >> 1, Wp = W >> 1;
    const size_t in_plane = (MODE == 2) ? (size_t)Hp * Wp : HW;
#pragma unroll
    for (int i = 0; i < X_PER_T; ++i) {
        const int e = tid + i * 256;
        const int ci = e / XS, rem = e - ci * XS;
        const int r = rem / XC, cc = rem - r * XC;
        const int gy = y0 + r - 1, gx = x0 + cc - 1;
        const bool ok = (e < X_ELEMS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (ok) xvalid |= 1u << i;
        if (MODE == 2) {
            // H, W odd tails are not pooled (floor): such pixels get no gradient
            const bool pooled = ok && (gy >> 1) < Hp && (gx >> 1) < Wp;
            if (!pooled) xvalid &= ~(1u << i);
            xoff[i] = pooled ? (int)(ci * in_plane + (size_t)(gy >> 1) * Wp + (gx >> 1)) : 0;
            xpos[i] = ((gy & 1) << 1) | (gx & 1);
        } else {
            xoff[i] = ok ? (int)(ci * in_plane + (size_t)gy * W + gx) : 0;
        }
    }
    const float *xin = a.x + (size_t)n * a.Cin * in_plane;
    const float *auxin = (MODE != 0) ? a.aux + (size_t)n * a.Cin * in_plane : nullptr;
    const uint8_t *idxin = (MODE == 2) ? a.idx + (size_t)n * a.Cin * in_plane : nullptr;

    // raw loaded values; the gates are applied when the chunk is written to LDS, AFTER the
    // MFMAs of the previous chunk, so no s_waitcnt on these loads sits in front of the MFMAs
    float xv[X_PER_T];
    float xa[MODE != 0 ? X_PER_T : 1];
    unsigned char xi[MODE == 2 ? X_PER_T : 1];
    f32x4 wv[W4_PER_T];

    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int ci0 = c * KC;
        const size_t cbase = (size_t)ci0 * in_plane;
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {
            const int e = tid + i * 256;
            const int ci = e / XS;
            // always-valid address (a branch per load would serialise the loads)
            const bool ok = ((xvalid >> i) & 1u) && (ci0 + ci) < a.Cin;
            const size_t o = ok ? cbase + xoff[i] : 0;
            xv[i] = xin[o];
            if (MODE != 0) xa[i] = auxin[o];
            if (MODE == 2) xi[i] = idxin[o];
        }
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) {
            const int e4 = min(tid + i * 256, W_ELEMS / 4 - 1);   // tail threads re-load the last item
            const int row = e4 / (BM / 4), c4 = e4 - row * (BM / 4);
            const int tap = row / KC, k = row - tap * KC;
            wv[i] = *reinterpret_cast<const f32x4 *>(a.w + ((size_t)(tap * a.CinP + ci0 + k) * a.CoutP + co0 + c4 * 4));
        }
    };
    auto store_chunk = [&](int c, int buf) __attribute__((always_inline)) {
        const int ci0 = c * KC;
        float *Ws = smem + buf * STAGE;
        float *Xs = Ws + W_PAD;
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {
            const int e = tid + i * 256;
            const int ci = e / XS;
            bool ok = ((xvalid >> i) & 1u) && (ci0 + ci) < a.Cin;
            if (MODE != 0) ok = ok && (xa[i] > 0.f);                 // ReLU gate
            if (MODE == 2) ok = ok && (xi[i] == xpos[i]);            // max-unpool gate
            Xs[e] = ok ? xv[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) *reinterpret_cast<f32x4 *>(Ws + (tid + i * 256) * 4) = wv[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    const int nchunks = a.CinP / KC;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const float *Ws = smem + buf * STAGE;
        const float *Xs = Ws + W_PAD;
        const float *wa = Ws + lhi * BM + wm * (MT * 32) + l31;
        const float *xb = Xs + lhi * XS + (wn * NT) * XC + l31;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) {
                float av[MT], bv[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = wa[(tap * KC + ks * 2) * BM + m * 32];
#pragma unroll
                for (int q = 0; q < NT; ++q) bv[q] = xb[(ks * 2) * XS + (q + ky) * XC + kx];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m][q], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) store_chunk(c + 1, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + ReLU, NCHW store (each store instruction = two 128-B row segments)
    const int px = x0 + l31;
    float *yout = a.y + (size_t)n * a.Cout * HW;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * (MT * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co >= a.Cout) continue;
            const float bsum = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int py = y0 + wn * NT + q;
                if (py < H && px < W) {
                    float v = acc[m][q][r] + bsum;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    yout[(size_t)co * HW + (size_t)py * W + px] = v;
                }
            }
        }
    }
}

// w (Cout,Cin,3,3) -> fwd pack [9][CinP][CoutP] (CinP = ceil4(Cin), CoutP = ceil128(Cout))
//                   and dgrad pack [9][CoutP4][CinP128] with tap' = 8 - tap
__global__ void pack_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ wf, float *__restrict__ wd) {
    const int CinP = (Cin + 3) & ~3, CoutP = (Cout + 127) & ~127;
    const int CoutP4 = (Cout + 3) & ~3, CinP128 = (Cin + 127) & ~127;
    const size_t nf = (size_t)9 * CinP * CoutP, nd = (size_t)9 * CoutP4 * CinP128;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wf && i < nf) {
        const int co = i % CoutP, ci = (i / CoutP) % CinP, tap = i / ((size_t)CoutP * CinP);
        wf[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    }
    if (wd && i < nd) {
        const int ci = i % CinP128, co = (i / CinP128) % CoutP4, tap = i / ((size_t)CinP128 * CoutP4);
        wd[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;
    }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ y, float *__restrict__ p,
                                                      uint8_t *__restrict__ idx, size_t planes, int H, int W) {
    const int Hp = H >> 1, Wp = W >> 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Hp * Wp) return;
    const int xo = i % Wp, yo = (i / Wp) % Hp;
    const size_t pl = i / ((size_t)Wp * Hp);
    const float *src = y + pl * H * W + (size_t)(2 * yo) * W + 2 * xo;
    const float2 r0 = *reinterpret_cast<const float2 *>(src);
    const float2 r1 = *reinterpret_cast<const float2 *>(src + W);
    // ATen max_pool2d scans the window row-major and keeps the first maximum (NaN propagates)
    float best = r0.x; int bi = 0;
    if (r0.y > best || r0.y != r0.y) { best = r0.y; bi = 1; }
    if (r1.x > best || r1.x != r1.x) { best = r1.x; bi = 2; }
    if (r1.y > best || r1.y != r1.y) { best = r1.y; bi = 3; }
    p[i] = best;
    if (idx) idx[i] = (uint8_t)bi;
}

// Input gradient of a conv with <= 4 INPUT channels (conv1_1: 64 -> 3 at full resolution).  With
// M = 3 the implicit GEMM would waste 29/32 of every MFMA, and the layer is HBM-bound anyway
// (reads the 64-channel gradient + its ReLU gate: 2 x 64 x H x W x 4 B, writes 3 x H x W x 4 B),
// so it runs on the vector ALU: one thread per pixel, 3 accumulators, the gated gradient tile
// [8][10][34] staged in LDS with 16-byte loads, weights read as scalars (wave-uniform).
// wd is the dgrad pack [tap'][CoutP4][CinP128] of pack_kernel.
constexpr int SG_KC = 8, SG_TH = 8, SG_PCP = 40;     // LDS row pitch 40: columns x0-4 .. x0+35
template <int CI>
__global__ __launch_bounds__(256) void dgrad_small_kernel(const float *__restrict__ gy, const float *__restrict__ act,
                                                          const float *__restrict__ wd, float *__restrict__ gx, int Cout,
                                                          int CoutP4, int H, int W, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float tile[SG_KC][SG_TH + 2][SG_PCP];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
    const int x0 = tx * 32, y0 = ty * SG_TH;
    const int px = tid & 31, py = tid >> 5;
    const size_t HW = (size_t)H * W;
    const float *gb = gy + (size_t)n * Cout * HW;
    const float *ab = act ? act + (size_t)n * Cout * HW : nullptr;
    // staging items: 8 channels x 10 rows x 10 float4 (columns x0-4+4l .. +3); W % 4 == 0 so an item is
    // entirely inside or outside the image
    constexpr int ITEMS = SG_KC * (SG_TH + 2) * 10;
    float acc[CI];
#pragma unroll
    for (int i = 0; i < CI; ++i) acc[i] = 0.f;
    for (int c0 = 0; c0 < Cout; c0 += SG_KC) {
        __syncthreads();
        for (int e = tid; e < ITEMS; e += 256) {
            const int ci = e / 100, rem = e - ci * 100, r = rem / 10, l = rem - r * 10;
            const int yy = y0 + r - 1, xx = x0 - 4 + 4 * l;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W && c0 + ci < Cout) {
                const size_t o = (size_t)(c0 + ci) * HW + (size_t)yy * W + xx;
                v = *reinterpret_cast<const float4 *>(gb + o);
                if (ab) {
                    const float4 m = *reinterpret_cast<const float4 *>(ab + o);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                    v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
            }
            *reinterpret_cast<float4 *>(&tile[ci][r][4 * l]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < SG_KC; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const float v = tile[c][py + ky][px + kx + 3];        // column x0-1+px+kx sits at LDS column +3
                const float *wr = wd + ((size_t)tap * CoutP4 + (c0 + c)) * 128;   // wave-uniform: scalar loads
#pragma unroll
                for (int i = 0; i < CI; ++i) acc[i] += v * wr[i];
            }
        }
    }
    const int ox = x0 + px, oy = y0 + py;
    if (ox < W && oy < H) {
#pragma unroll
        for (int i = 0; i < CI; ++i) gx[((size_t)n * CI + i) * HW + (size_t)oy * W + ox] = acc[i];
    }
}

template <int MODE>
int launch_conv(const ConvArgs &a0, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_x = st3d::cdiv(a.W, TW);
    if (a.Cout > 64) {
        constexpr int TH = 8, BM = 128;
        dim3 grid(a.tiles_x * st3d::cdiv(a.H, TH), st3d::cdiv(a.Cout, BM), a.N);
        conv3x3_kernel<2, 2, MODE><<<grid, 256, 0, s>>>(a);
    } else if (a.Cout > 32) {
        constexpr int TH = 16, BM = 64;
        dim3 grid(a.tiles_x * st3d::cdiv(a.H, TH), st3d::cdiv(a.Cout, BM), a.N);
        conv3x3_kernel<1, 2, MODE><<<grid, 256, 0, s>>>(a);
    } else {
        constexpr int TH = 16, BM = 32;
        dim3 grid(a.tiles_x * st3d::cdiv(a.H, TH), st3d::cdiv(a.Cout, BM), a.N);
        conv3x3_kernel<1, 1, MODE><<<grid, 256, 0, s>>>(a);
    }
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" size_t st3d_conv3x3_packed_floats(int Cout, int Cin) {
    const size_t f = (size_t)9 * ceil_to(Cin, 4) * ceil_to(Cout, 128);
    const size_t d = (size_t)9 * ceil_to(Cout, 4) * ceil_to(Cin, 128);
    return f > d ? f : d;
}

extern "C" int st3d_conv3x3_pack(const float *w, int Cout, int Cin, float *w_fwd, float *w_dgrad, st3d_stream_t stream) {
    ST3D_CHECK_ARG(w && (w_fwd || w_dgrad));
    ST3D_CHECK_ARG(Cout > 0 && Cin > 0);
    const size_t n = st3d_conv3x3_packed_floats(Cout, Cin);
    pack_kernel<<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(w, Cout, Cin, w_fwd, w_dgrad);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_conv3x3_fwd(const float *x, const float *w_fwd_packed, const float *bias, float *y, int N, int Cin,
                                int Cout, int H, int W, int relu, st3d_stream_t stream) {
    ST3D_CHECK_ARG(x && w_fwd_packed && y);
    ST3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)w_fwd_packed & 15) == 0);
    ConvArgs a{x, nullptr, nullptr, w_fwd_packed, bias, y, N, Cin, Cout, H, W, ceil_to(Cin, 4), ceil_to(Cout, 128), relu, 0};
    return launch_conv<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_conv3x3_dgrad(const float *gy, const float *act, const float *w_dgrad_packed, float *gx, int N, int Cin,
                                  int Cout, int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && w_dgrad_packed && gx);
    ST3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)w_dgrad_packed & 15) == 0);
    if (Cin == 3 && (W % 4) == 0 && (Cout % SG_KC) == 0 && ceil_to(Cin, 128) == 128) {     // conv1_1: HBM-bound VALU kernel
        const int tiles_x = st3d::cdiv(W, 32);
        dgrad_small_kernel<3><<<dim3(tiles_x * st3d::cdiv(H, SG_TH), N), 256, 0, st3d::as_stream(stream)>>>(
            gy, act, w_dgrad_packed, gx, Cout, ceil_to(Cout, 4), H, W, tiles_x);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
    // the transposed convolution reads Cout channels and writes Cin channels
    ConvArgs a{gy, act, nullptr, w_dgrad_packed, nullptr, gx, N, Cout, Cin, H, W, ceil_to(Cout, 4), ceil_to(Cin, 128), 0, 0};
    return act ? launch_conv<1>(a, st3d::as_stream(stream)) : launch_conv<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_conv3x3_dgrad_unpool(const float *gy_pooled, const uint8_t *pool_idx, const float *pooled,
                                         const float *w_dgrad_packed, float *gx, int N, int Cin, int Cout, int H, int W,
                                         st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy_pooled && pool_idx && pooled && w_dgrad_packed && gx);
    ST3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 1 && W > 1);
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)w_dgrad_packed & 15) == 0);
    ConvArgs a{gy_pooled, pooled, pool_idx, w_dgrad_packed, nullptr, gx, N, Cout, Cin, H, W, ceil_to(Cout, 4),
               ceil_to(Cin, 128), 0, 0};
    return launch_conv<2>(a, st3d::as_stream(stream));
}

extern "C" int st3d_maxpool2x2_fwd(const float *y, float *p, uint8_t *idx, int N, int C, int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(y && p);
    ST3D_CHECK_ARG(N > 0 && C > 0 && H > 1 && W > 1 && (W % 2) == 0);
    const size_t n = (size_t)N * C * (H / 2) * (W / 2);
    maxpool_kernel<<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(y, p, idx, (size_t)N * C, H, W);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
