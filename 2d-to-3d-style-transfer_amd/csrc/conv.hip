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
    float2 r0, r1;
    if ((W & 1) == 0) {         // rows stay 8-byte aligned
        r0 = *reinterpret_cast<const float2 *>(src);
        r1 = *reinterpret_cast<const float2 *>(src + W);
    } else {                    // odd width (floor pooling drops the last column, like MaxPool2d)
        r0 = make_float2(src[0], src[1]);
        r1 = make_float2(src[W], src[W + 1]);
    }
    // ATen max_pool2d scans the window row-major and keeps the first maximum (NaN propagates)
    float best = r0.x; int bi = 0;
    if (r0.y > best || r0.y != r0.y) { best = r0.y; bi = 1; }
    if (r1.x > best || r1.x != r1.x) { best = r1.x; bi = 2; }
    if (r1.y > best || r1.y != r1.y) { best = r1.y; bi = 3; }
    p[i] = best;
    if (idx) idx[i] = (uint8_t)bi;
}

// Input gradient of a conv with 3 INPUT channels (conv1_1: 64 -> 3 at full resolution).  With M = 3 the implicit
// GEMM would waste 29/32 of every MFMA, and the layer is HBM-bound anyway (reads the 64-channel gradient + its ReLU
// gate: 2 x 64 x H x W x 4 B, writes 3 x H x W x 4 B), so it runs on the vector ALU.  A 256-thread workgroup owns an
// 8 x 128-pixel tile, every thread 4 consecutive pixels x 3 channels = 12 accumulators: per input channel it reads its
// 3 x 6 gated-gradient window with 9 LDS instructions and the 27 weights with 7 broadcast reads for 108 FMAs.  The
// gated gradient of the NEXT 4 channels is fetched into registers (16-byte loads) while the current 4 are consumed.
// wd is the dgrad pack [tap'][CoutP4][CinP128] of pack_kernel; it is re-laid [cout][tap'][3 (+1 pad)] in LDS once.
constexpr int SG_KC = 4, SG_TH = 8, SG_TW = 128, SG_PCP = 140;     // LDS row: columns x0-4 .. x0+131 (+ pad)
template <int CI>
__global__ __launch_bounds__(256) void dgrad_small_kernel(const float *__restrict__ gy, const float *__restrict__ act,
                                                          const float *__restrict__ wd, float *__restrict__ gx, int Cout,
                                                          int CoutP4, int H, int W, int tiles_x) {
    static_assert(CI == 3, "conv1_1 only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*tile)[SG_TH + 2][SG_PCP] = reinterpret_cast<float(*)[SG_TH + 2][SG_PCP]>(smem);      // [SG_KC]
    float *wl = smem + SG_KC * (SG_TH + 2) * SG_PCP;                                              // [Cout][28]
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
    const int x0 = tx * SG_TW, y0 = ty * SG_TH;
    const int px = tid & 31, py = tid >> 5;
    const size_t HW = (size_t)H * W;
    const float *gb = gy + (size_t)n * Cout * HW;
    const float *ab = act ? act + (size_t)n * Cout * HW : nullptr;
    for (int e = tid; e < Cout * 28; e += 256) {
        const int c = e / 28, r = e - c * 28, tap = r / 3, i = r - tap * 3;
        wl[e] = (r < 27) ? wd[((size_t)tap * CoutP4 + c) * 128 + i] : 0.f;
    }
    // staging items: SG_KC channels x 10 rows x 34 float4 (columns x0-4+4l .. +3); W % 4 == 0 so an item is entirely
    // inside or outside the image
    constexpr int ROW4 = (SG_TW + 8) / 4;
    constexpr int ITEMS = SG_KC * (SG_TH + 2) * ROW4;
    constexpr int NIT = (ITEMS + 255) / 256;
    float acc[4][CI];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < CI; ++i) acc[p][i] = 0.f;
    f32x4 rg[NIT], ra[NIT];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int e = tid + 256 * j;
            const int ci = e / ((SG_TH + 2) * ROW4), rem = e - ci * ((SG_TH + 2) * ROW4), r = rem / ROW4, l = rem - r * ROW4;
            const int yy = y0 + r - 1, xx = x0 - 4 + 4 * l;
            const bool ok = e < ITEMS && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const size_t o = ok ? (size_t)(c0 + ci) * HW + (size_t)yy * W + xx : 0;     // always a valid address
            rg[j] = *reinterpret_cast<const f32x4 *>(gb + o);
            ra[j] = ab ? *reinterpret_cast<const f32x4 *>(ab + o) : f32x4{1.f, 1.f, 1.f, 1.f};
            if (!ok) ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cout; c0 += SG_KC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int e = tid + 256 * j;
            if (e < ITEMS) {
                const int ci = e / ((SG_TH + 2) * ROW4), rem = e - ci * ((SG_TH + 2) * ROW4), r = rem / ROW4, l = rem - r * ROW4;
                f32x4 v;
                v.x = ra[j].x > 0.f ? rg[j].x : 0.f; v.y = ra[j].y > 0.f ? rg[j].y : 0.f;
                v.z = ra[j].z > 0.f ? rg[j].z : 0.f; v.w = ra[j].w > 0.f ? rg[j].w : 0.f;
                *reinterpret_cast<f32x4 *>(&tile[ci][r][4 * l]) = v;
            }
        }
        __syncthreads();
        if (c0 + SG_KC < Cout) fetch(c0 + SG_KC);
#pragma unroll
        for (int c = 0; c < SG_KC; ++c) {
            float w[28];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(wl + (c0 + c) * 28 + 4 * q);      // same address in every lane
                w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // window columns x-1 .. x+4 of the thread's 4 pixels sit at LDS columns 4px+3 .. 4px+8
                const float *row = &tile[c][py + ky][4 * px];
                const float l = row[3];
                const f32x4 m = *reinterpret_cast<const f32x4 *>(row + 4);
                const float r = row[8];
                const float v[6] = {l, m.x, m.y, m.z, m.w, r};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int i = 0; i < CI; ++i) acc[p][i] += v[p + kx] * w[(ky * 3 + kx) * 3 + i];
            }
        }
    }
    const int oy = y0 + py, ox = x0 + 4 * px;
    if (oy < H && ox < W) {                     // W % 4 == 0: the 4 pixels are inside together
#pragma unroll
        for (int i = 0; i < CI; ++i)
            *reinterpret_cast<f32x4 *>(gx + ((size_t)n * CI + i) * HW + (size_t)oy * W + ox) =
                f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
    }
}

// Forward of a conv with 3 INPUT channels (conv1_1: 3 -> 64 at full resolution): K = 27 cannot feed the matrix pipe
// (the direct MFMA kernel pads K to 36 and spends its time in prologue/epilogue), and the layer only has to stream
// its 64-channel output, so it runs on the vector ALU too.  Same geometry as dgrad_small_kernel: 8 x 128-pixel tile,
// 4 consecutive pixels per thread; the 3-channel haloed patch is staged once, the thread keeps its 3 x 3 x 6 window
// in registers and walks the output channels 16 at a time (64 accumulators): per weight row one 16-float broadcast
// read for 64 FMAs; bias + ReLU fused; 16-byte stores.  wf is the forward pack [tap][CinP4][CoutP128].
constexpr int SF_CG = 16;
__global__ __launch_bounds__(256, 2) void conv_small_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wf,
                                                             const float *__restrict__ bias, float *__restrict__ y, int Cout,
                                                             int CoutP, int H, int W, int tiles_x, int relu) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*tile)[SG_TH + 2][SG_PCP] = reinterpret_cast<float(*)[SG_TH + 2][SG_PCP]>(smem);      // [3]
    float *wl = smem + 3 * (SG_TH + 2) * SG_PCP;                                                 // [27][Cout], k = (ci*3+ky)*3+kx
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
    const int x0 = tx * SG_TW, y0 = ty * SG_TH;
    const int px = tid & 31, py = tid >> 5;
    const size_t HW = (size_t)H * W;
    const float *xb = x + (size_t)n * 3 * HW;
    for (int e = tid; e < 27 * Cout; e += 256) {
        const int k = e / Cout, co = e - k * Cout, ci = k / 9, tap = k - ci * 9;
        wl[e] = wf[((size_t)tap * 4 + ci) * CoutP + co];
    }
    constexpr int ROW4 = (SG_TW + 8) / 4;
    constexpr int ITEMS = 3 * (SG_TH + 2) * ROW4;
    for (int e = tid; e < ITEMS; e += 256) {
        const int ci = e / ((SG_TH + 2) * ROW4), rem = e - ci * ((SG_TH + 2) * ROW4), r = rem / ROW4, l = rem - r * ROW4;
        const int yy = y0 + r - 1, xx = x0 - 4 + 4 * l;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f32x4 *>(xb + (size_t)ci * HW + (size_t)yy * W + xx);
        *reinterpret_cast<f32x4 *>(&tile[ci][r][4 * l]) = v;
    }
    __syncthreads();
    float win[9][6];        // [row = ci*3+ky][column x-1 .. x+4]
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float *row = &tile[ci][py + ky][4 * px];
            const f32x4 m = *reinterpret_cast<const f32x4 *>(row + 4);
            float *w6 = win[ci * 3 + ky];
            w6[0] = row[3]; w6[1] = m.x; w6[2] = m.y; w6[3] = m.z; w6[4] = m.w; w6[5] = row[8];
        }
    const int oy = y0 + py, ox = x0 + 4 * px;
    const bool inside = oy < H && ox < W;
    for (int cg = 0; cg < Cout; cg += SF_CG) {
        float acc[SF_CG][4];
#pragma unroll
        for (int j = 0; j < SF_CG; ++j) {
            const float bv = bias ? bias[cg + j] : 0.f;
            acc[j][0] = bv; acc[j][1] = bv; acc[j][2] = bv; acc[j][3] = bv;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {           // r = ci*3 + ky
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float *wrow = wl + (r * 3 + kx) * Cout + cg;          // same address in every lane
#pragma unroll
                for (int q = 0; q < SF_CG / 4; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(wrow + 4 * q);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        acc[4 * q][p] += win[r][p + kx] * wv.x;
                        acc[4 * q + 1][p] += win[r][p + kx] * wv.y;
                        acc[4 * q + 2][p] += win[r][p + kx] * wv.z;
                        acc[4 * q + 3][p] += win[r][p + kx] * wv.w;
                    }
                }
            }
            // pin the schedule per window row: without this the compiler reads all 27 weight rows up front and
            // spills them (the accumulators pass through an empty asm, so this row's FMAs end here and the next
            // row's LDS reads start after it)
#pragma unroll
            for (int j = 0; j < SF_CG; ++j)
                asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]), "+v"(acc[j][2]), "+v"(acc[j][3]) : : "memory");
        }
        if (inside) {
#pragma unroll
            for (int j = 0; j < SF_CG; ++j) {
                f32x4 o = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                *reinterpret_cast<f32x4 *>(y + ((size_t)n * Cout + cg + j) * HW + (size_t)oy * W + ox) = o;
            }
        }
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
    if (Cin == 3 && (W % 4) == 0 && (Cout % SF_CG) == 0 && Cout <= 128) {          // conv1_1: VALU kernel
        const int tiles_x = st3d::cdiv(W, SG_TW);
        const size_t lds = ((size_t)3 * (SG_TH + 2) * SG_PCP + (size_t)27 * Cout) * sizeof(float);
        conv_small_fwd_kernel<<<dim3(tiles_x * st3d::cdiv(H, SG_TH), N), 256, lds, st3d::as_stream(stream)>>>(
            x, w_fwd_packed, bias, y, Cout, ceil_to(Cout, 128), H, W, tiles_x, relu);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
    ConvArgs a{x, nullptr, nullptr, w_fwd_packed, bias, y, N, Cin, Cout, H, W, ceil_to(Cin, 4), ceil_to(Cout, 128), relu, 0};
    return launch_conv<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_conv3x3_dgrad(const float *gy, const float *act, const float *w_dgrad_packed, float *gx, int N, int Cin,
                                  int Cout, int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && w_dgrad_packed && gx);
    ST3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
    ST3D_CHECK_ARG((size_t)Cin * H * W < (1u << 31) && (size_t)Cout * H * W < (1u << 31));
    ST3D_CHECK_ARG(((uintptr_t)w_dgrad_packed & 15) == 0);
    if (Cin == 3 && (W % 4) == 0 && (Cout % SG_KC) == 0 && Cout <= 256 && ceil_to(Cin, 128) == 128) {     // conv1_1: VALU kernel
        const int tiles_x = st3d::cdiv(W, SG_TW);
        const size_t lds = ((size_t)SG_KC * (SG_TH + 2) * SG_PCP + (size_t)Cout * 28) * sizeof(float);
        dgrad_small_kernel<3><<<dim3(tiles_x * st3d::cdiv(H, SG_TH), N), 256, lds, st3d::as_stream(stream)>>>(
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
    ST3D_CHECK_ARG(N > 0 && C > 0 && H > 1 && W > 1);
    const size_t n = (size_t)N * C * (H / 2) * (W / 2);
    maxpool_kernel<<<st3d::cdiv((long)n, 256), 256, 0, st3d::as_stream(stream)>>>(y, p, idx, (size_t)N * C, H, W);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
