// wino43.hip -- conv3x3(pad 1) forward / input-gradient as Winograd F(4x4,3x3) on the gfx950 fp32 matrix pipe
// (round 3): 36 multiplies per 4x4 output tile = 2.25 per output pixel, against 4 for the F(2x2,3x3) of wino.hip and 9
// for the direct convolution.  Still fp32 products and fp32 accumulation; the larger transform constants cost about one
// decimal digit (measured <= 1.3e-5 of the output scale at K = 512 against an fp64 convolution; F(2x2,3x3): 1e-6), inside
// the 3e-5 the Winograd kernels are tested to.  Used for the deep layers (Cin >= 128 at W % 64 == 0), where the K loop
// dominates; the short-K layers stay on wino4_kernel.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A       per (cout, 4x4 tile), 6x6 Winograd domain xi = (a, b)
//
// Mapping.  One 768-thread workgroup (12 waves, three per SIMD, one workgroup per CU) = 64 cout x 16 tiles (4 rows x 64
// columns of pixels).  Wave w = (a = w >> 1, bh = w & 1) accumulates row a of the domain for b in {3 bh .. 3 bh + 2} and
// all 64 couts on v_mfma_f32_16x16x4_f32 (M = 16 cout, N = 16 tiles, K = 4 input channels): 3 b x 4 cout blocks = 12
// accumulator tiles = 48 VGPRs, and every B operand feeds FOUR MFMAs.
//   * B operands V = B^T d B.  The ROW half of the transform (T_a = sum_r B^T[a][r] d[r], 13 VALU per column for all six
//     a) is done ONCE per patch column by the staging threads on its way into LDS -- the ring holds T[ci][a][column], not
//     the raw patch -- so a wave reads only ITS row a (6 floats per (tile, channel)) and does the column half for its
//     three b (6-7 VALU).  That is the same ~1 vector-ALU instruction per 64 MFMA cycles in the waves' loop as
//     wino4_kernel has, for 2.25 instead of 4 MFMA-multiplies per output.
//   * A operands U = G g G^T (fp64 -> fp32, packed per lane: 12 contiguous floats per 4-channel k-step) come straight from
//     L2 through a buffer descriptor, one k-step ahead; they never touch LDS.
//   * Staging: 16 channels per stage, 3-deep ring, one barrier per stage.  A half-item = (channel, column pair): six row
//     loads -> gate (MODE) -> row transform -> six 8-byte LDS stores.  576 half-items per stage on 768 threads: every
//     wave stages the same amount every stage, interleaved with its MFMAs -- with one workgroup per CU a wave that works
//     alone behind its MFMAs (staging in turns was tried first) holds up all twelve at the barrier.
//   * Epilogue: along b in registers (partial over the wave's three b), then per output column j one LDS exchange
//     [6 a][64 co][16 tiles][2 bh]; a reader owns whole 4x4 output tiles (two per thread of the first 8 waves), so the
//     2x2 max-pool (+argmax), ReLU, the producer-side gates of the backward chain and 16-byte row stores all happen in
//     registers, as in wino4_kernel.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NT6 = 768;                 // 12 waves
constexpr int T6_ROWS = 4, T6_COLS = 64; // output pixels per workgroup: one row of 16 4x4 tiles
constexpr int KS6 = 16;                  // input channels per stage (four k-steps of 4): one barrier and one staging turn per 48 MFMAs
constexpr int NK6 = KS6 / 4;
constexpr int PITCH6 = 76;               // 18 strips of 4 floats (patch columns x0 - 4 .. x0 + 67) stored from index 1: a tile's six
                                         // columns 4 tx + 3 .. + 8 then start 16-byte aligned (one 16-byte + one 8-byte LDS read)
constexpr int STRIPS6 = 18;
constexpr int ITEMS6 = KS6 * STRIPS6;    // 288 staging items per stage: 4.5 waves -> the two groups of six waves take turns
constexpr int GRP6 = 384;                // threads per staging group
constexpr int TSTAGE6 = KS6 * 6 * PITCH6;        // floats per ring stage: [ci 8][a 6][72]
constexpr int EX6 = 6 * 64 * 16 * 2;             // exchange floats per output column: [a 6][co 64][tile 16][bh 2]
constexpr int SMEM6 = EX6 > 3 * TSTAGE6 ? EX6 : 3 * TSTAGE6;      // 81 KB (dynamic shared memory: above the 64 KB static limit)

struct Wino43Args {
    const float *x;       // MODE 0: (N,Cin,H,W); MODE 3: pooled-resolution gradient (N,Cin,H/2,W/2), already gated
    const uint8_t *idx;   // MODE 3: pool argmax
    const float *U;       // packed [ct][kstep][wave 12][lane 64][12]  (wino43_pack_kernel)
    const float *bias;    // (Cout) or nullptr
    float *y;             // (N,Cout,H,W) or nullptr (EPI 1 may skip the full-resolution store)
    float *yp;            // EPI 1: pooled output (N,Cout,H/2,W/2)
    uint8_t *yidx;        // EPI 1: argmax
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_y, n_ct;
    const float *gate;    // (N,Cout,H,W) or nullptr: outputs are zeroed where gate <= 0 (the consumer's ReLU gate, see wino.hip)
    const float *addt;    // with gate: outputs become gate > 0 ? y + addc * (gate - addt) : 0
    float addc;
};

// T_a = sum_r B^T[a][r] d[r] for a = 0..5 (and, applied to t0..t5, the column half V_b = sum_c B^T[b][c] t[c])
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float &t0, float &t1, float &t2,
                                    float &t3, float &t4, float &t5) {
    const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
    t1 = p + q; t2 = p - q;
    t0 = 4.f * d0 + (d4 - 5.f * d2);
    const float r = d4 - d2, s = 2.f * (d3 - d1);
    t3 = r + s; t4 = r - s;
    t5 = 4.f * d1 + (d5 - 5.f * d3);
}

// MODE 0: plain input.  MODE 3: input = 2x2 max-unpool of a pooled-resolution gradient (routing by the argmax bytes).
// EPI 1: MaxPool2d(2,2) (+argmax) fused into the epilogue.  GATE as in wino4_kernel (0 none, 1 output gate, 2 + content term).
template <int MODE, int EPI, int GATE>
__global__ __launch_bounds__(NT6, 3) void wino43_kernel(const Wino43Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: every role test below is a scalar branch)
    const int wa = wave >> 1, bh = wave & 1;           // row a of the domain, b half
    const int tx = lane & 15, kq = lane >> 4;           // MFMA n index (tile) / k index (channel within the k-step)

    // grid: cout tile fastest, then pixel tiles (x, y), then image
    int bid = blockIdx.x;
    const int ct = bid % a.n_ct; bid /= a.n_ct;
    const int tile_x = bid % a.tiles_x; bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int x0 = tile_x * T6_COLS, y0 = tile_y * T6_ROWS;
    const int co0 = ct * 64;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    constexpr bool UNPOOL = MODE == 3;
    const size_t in_plane = UNPOOL ? (size_t)Hp * Wp : HW;
    const int nstages = a.Cin / KS6;
    const unsigned kOob = 0x80000000u;

    // ---- staging: EVERY thread stages one half-item per stage = (channel ci, strip l, column pair h): rows gy = y0 - 1 + r,
    // columns x0 - 4 + 4 l + 2 h, + 1.  576 half-items on 768 threads: all twelve waves carry the same staging work, so it
    // sits in the same basic block as their MFMAs (interleaved, no tail behind them) and nobody is late at the barrier.
    const int s_ci = tid / 36, s_rem = tid - s_ci * 36;
    const int s_col = 2 * s_rem;                         // = 4 l + 2 h, column offset inside the 72-float strip row
    const bool s_on = tid < ITEMS6 * 2;
    unsigned voff[6];
    unsigned rowbit[UNPOOL ? 6 : 1];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int gy = y0 - 1 + r, gx0 = x0 - 4 + s_col;
        const bool ok = s_on && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
        if (UNPOOL) {       // one pooled element (and its argmax byte) covers this row's two columns
            voff[r] = ok ? (unsigned)((s_ci * in_plane + (size_t)(gy >> 1) * Wp + (gx0 >> 1)) * 4) : kOob;
            rowbit[r] = (unsigned)(gy & 1) << 1;
        } else {
            voff[r] = ok ? (unsigned)((s_ci * in_plane + (size_t)gy * W + gx0) * 4) : kOob;
        }
    }
    const int loff = s_on ? (s_ci * 6 * PITCH6 + s_col + 1) : 0;      // + a * PITCH6 per transformed row (odd index: two 4-byte stores in one ds_write2)
    const unsigned img_bytes = (unsigned)((size_t)a.Cin * in_plane * 4);
    const unsigned stage_bytes = (unsigned)(KS6 * in_plane * 4);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x + (size_t)n * a.Cin * in_plane), 0, img_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ridx = rx;
    if (UNPOOL)
        ridx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.idx + (size_t)n * a.Cin * in_plane), 0, img_bytes / 4, 0x00020000);
    const int nksteps = a.Cin / 4;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.U + (size_t)ct * nksteps * (12 * 64 * 12)), 0, (unsigned)((size_t)nksteps * 12 * 64 * 12 * 4), 0x00020000);
    const unsigned uvoff = (unsigned)((wave * 64 + lane) * 48);

    struct Staged { f32x2 v[UNPOOL ? 1 : 6]; float g[UNPOOL ? 6 : 1]; unsigned i[UNPOOL ? 6 : 1]; };
    auto gload = [&](int st, Staged &x) __attribute__((always_inline)) {
        const unsigned so = (unsigned)st * stage_bytes;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (UNPOOL) {
                // y0 % 4 == 0: patch rows (1, 2) and (3, 4) are the two halves of ONE pooled row each -- four loads, not six
                if (r == 2 || r == 4) { x.g[r] = x.g[r - 1]; x.i[r] = x.i[r - 1]; continue; }
                x.g[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff[r], so, 0));
                x.i[r] = __builtin_amdgcn_raw_buffer_load_b8(ridx, voff[r] == kOob ? kOob : voff[r] / 4, so / 4, 0);
            } else {
                x.v[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, voff[r], so, 0));
            }
        }
    };
    // gate (MODE 3: route the pooled gradient to its argmax position), row-transform the two columns, store the six T rows
    auto lstore = [&](int buf, const Staged &x) __attribute__((always_inline)) {
        f32x2 t[6];
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            float d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (UNPOOL) d[r] = ((x.i[r] & 0xffu) == (rowbit[r] | (unsigned)j2)) ? x.g[r] : 0.f;
                else d[r] = x.v[r][j2];
            }
            float t0, t1, t2, t3, t4, t5;
            bt6(d[0], d[1], d[2], d[3], d[4], d[5], t0, t1, t2, t3, t4, t5);
            t[0][j2] = t0; t[1][j2] = t1; t[2][j2] = t2; t[3][j2] = t3; t[4][j2] = t4; t[5][j2] = t5;
        }
        float *dst = &smem[buf * TSTAGE6 + loff];
        if (s_on) {
#pragma unroll
            for (int aa = 0; aa < 6; ++aa) { dst[aa * PITCH6] = t[aa][0]; dst[aa * PITCH6 + 1] = t[aa][1]; }
        }
    };

    // ---- B operands: this lane's row-a values of (tile tx, channel 4 kk + kq): columns 4 tx + 3 .. 4 tx + 8 of the strip row
    const int tbase = (kq * 6 + wa) * PITCH6 + 4 * tx;
    struct Trow { f32x4 m; f32x2 n; };       // t0..t3, t4..t5
    auto tread = [&](int buf, int kk, Trow &o) __attribute__((always_inline)) {
        const float *p = &smem[buf * TSTAGE6 + kk * (4 * 6 * PITCH6) + tbase];
        o.m = *reinterpret_cast<const f32x4 *>(p + 4);         // columns 4 tx + 3 .. + 6 of the strip row (stored from index 1)
        o.n = *reinterpret_cast<const f32x2 *>(p + 8);
    };
    auto vcompute = [&](const Trow &o, float v[3]) __attribute__((always_inline)) {
        const float t0 = o.m[0], t1 = o.m[1], t2 = o.m[2], t3 = o.m[3], t4 = o.n[0], t5 = o.n[1];
        if (bh == 0) {
            const float p = t4 - 4.f * t2, q = t3 - 4.f * t1;
            v[0] = 4.f * t0 + (t4 - 5.f * t2);
            v[1] = p + q;
            v[2] = p - q;
        } else {
            const float r = t4 - t2, s = 2.f * (t3 - t1);
            v[0] = r + s;
            v[1] = r - s;
            v[2] = 4.f * t1 + (t5 - 5.f * t3);
        }
    };
    struct Uop { f32x4 q[3]; };               // [b 3] -> 4 floats (cout block)
    auto uload = [&](int kstep, Uop &u) __attribute__((always_inline)) {
#ifdef ST3D_W43_DIAG_U       // diagnostic build: every k-step reads the SAME (cache-resident) filter operands -- wrong results, timing only
        const unsigned so = (unsigned)(min(kstep, nksteps - 1) & 0) * (unsigned)(12 * 64 * 12 * 4);
#else
        const unsigned so = (unsigned)min(kstep, nksteps - 1) * (unsigned)(12 * 64 * 12 * 4);
#endif
#pragma unroll
        for (int b = 0; b < 3; ++b) u.q[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff + 16 * b, so, 0));
    };

    f32x4 acc[3][4];       // [b][cout block]
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 and 1 (both requests in flight together)
    {
        Staged x0s, x1s;
        gload(0, x0s);
        gload(nstages > 1 ? 1 : 0, x1s);
        lstore(0, x0s);
        lstore(1, x1s);
    }
    // filter operands THREE k-steps ahead (k-step kk of a stage uses set kk and requests set (kk + 3) & 3): F(4x4,3x3)'s U is
    // 2.25x the F(2x2,3x3) one (4.7 MB per cout tile at 512 x 512: it streams through the XCD's L2 from the memory-side
    // cache), and with one workgroup per CU a late operand stalls the whole CU
    Uop u4[4];
    uload(0, u4[0]);
    uload(1, u4[1]);
    uload(2, u4[2]);
    __syncthreads();
    Trow trow;
    float vcur[3], vnext[3];
    tread(0, 0, trow);
    vcompute(trow, vcur);

    // One stage = 16 input channels = four k-steps of 12 MFMAs, one barrier.  The loop is unrolled by 3 stages: the ring
    // position P = c % 3 is a compile-time constant (immediate LDS offsets); k-step kk uses filter-operand set kk and requests
    // the set of three k-steps ahead.  Staging of stage c + 2: requested at the start of the stage, transformed and stored
    // under the MFMAs of its last k-step (three k-steps = ~3.5 k cycles in flight; every wave does the same, see above).
#define W6_MFMAS(u, v)                                                                                            \
    _Pragma("unroll") for (int b = 0; b < 3; ++b)                                                                 \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                          \
            acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.q[b][cb], v[b], acc[b][cb], 0, 0, 0);
    Staged xs;
    auto stage = [&](int c, auto Pc) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value, pb = P, pb1 = (P + 1) % 3, pb2 = (P + 2) % 3;
#pragma unroll
        for (int kk = 0; kk < NK6; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            // the next k-step's row (the last one prefetches k-step 0 of the NEXT stage: staged a barrier ago)
            if (kk + 1 < NK6) tread(pb, kk + 1, trow); else tread(pb1, 0, trow);
            // (B operands ping-pong between vcur / vnext by k-step parity: NK6 is even, so every stage starts on vcur)
            if (kk & 1) { W6_MFMAS(u4[kk], vnext) } else { W6_MFMAS(u4[kk], vcur) }
            if (kk == 0) gload(min(c + 2, nstages - 1), xs);
            uload(NK6 * c + kk + 3, u4[(kk + 3) & 3]);
            if (kk & 1) vcompute(trow, vcur); else vcompute(trow, vnext);
            if (kk == NK6 - 1) lstore(pb2, xs);
            // ST3D_W43_SCHED (compile time): how the k-step's non-MFMA instructions are placed among its 12 MFMAs.
            //   0: one MFMA, then its share of the others (the wino4_kernel pattern)   1: groups of 4 MFMAs
            //   2: the 12 MFMAs as one burst, everything else behind it               3: compiler's own order
#ifndef ST3D_W43_SCHED
#define ST3D_W43_SCHED 3
#endif
            constexpr int MG = ST3D_W43_SCHED == 0 ? 1 : (ST3D_W43_SCHED == 1 ? 4 : 12);
            if (ST3D_W43_SCHED != 3) {
#pragma unroll
                for (int i_ = 0; i_ < 12 / MG; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MG, 0);
                    if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x120, MG, 0);       // the stage's loads + LDS reads
                    else if (kk == NK6 - 1) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 3 * MG, 0);            // gate + row transform + column transform
                        __builtin_amdgcn_sched_group_barrier(0x320, MG, 0);                // LDS reads / writes, filter loads
                    } else __builtin_amdgcn_sched_group_barrier(0x126, MG, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    int c = 0;
    for (; c + 3 <= nstages; c += 3) {
        stage(c, std::integral_constant<int, 0>{});
        stage(c + 1, std::integral_constant<int, 1>{});
        stage(c + 2, std::integral_constant<int, 2>{});
    }
    if (c < nstages) stage(c, std::integral_constant<int, 0>{});
    if (c + 1 < nstages) stage(c + 1, std::integral_constant<int, 1>{});
#undef W6_MFMAS

    // ---- epilogue.  Along b (this wave's three b): z_j = sum_b A^T[j][b] m_b, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    //   bh 0 (b 0 1 2): z0 = m0 + s, z1 = z3 = d, z2 = s            with s = m1 + m2, d = m1 - m2
    //   bh 1 (b 3 4 5): z0 = s, z1 = 2 d, z2 = 4 s, z3 = 8 d + m5   with s = m3 + m4, d = m3 - m4
    // then per output column j one exchange [a][co][tile][bh]; readers sum the two halves and apply A^T along a.
    float *ex = smem;
    constexpr int NIT = 2;                         // 4x4 output tiles per reader thread (waves 0..7: 512 threads x 2 = 64 co x 16 tiles)
    const bool reader = tid < 512;
    float yt[NIT][4][4];                           // [item][row i][col j]
    // the readers' bias and the consumer's ReLU gate (GATE >= 1) are requested NOW: they arrive under the four exchange
    // passes instead of in front of the stores (one workgroup per CU: an exposed HBM round trip idles the whole CU)
    const unsigned out_bytes = (unsigned)((size_t)a.Cout * HW * 4);
    unsigned vo_it[NIT];
    float bias_it[NIT];
    f32x4 gq[GATE >= 1 ? NIT : 1][4];
    if (reader) {
        __amdgpu_buffer_rsrc_t rg = rx;
        if (GATE >= 1) rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gate + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = tid + 512 * it, col = q >> 4, tl = q & 15;
            const int ox = x0 + 4 * tl;
            const bool inb = y0 < H && ox < W;           // H % 4 == 0, W % 64 == 0: always (kept for the descriptor sentinel)
            vo_it[it] = inb ? (unsigned)((((size_t)co0 + col) * HW + (size_t)y0 * W + ox) * 4) : kOob;
            bias_it[it] = a.bias ? a.bias[co0 + col] : 0.f;
            if (GATE >= 1 && it == 0) {       // (the second item's gate is requested behind the passes: 16 more live registers spill)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    gq[it][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo_it[it], (unsigned)(i * W * 4), 0));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][cb][r], m1 = acc[1][cb][r], m2 = acc[2][cb][r];
                float z;
                if (bh == 0) z = j == 0 ? m0 + (m1 + m2) : (j == 2 ? m1 + m2 : m1 - m2);
                else z = j == 0 ? m0 + m1 : (j == 1 ? 2.f * (m0 - m1) : (j == 2 ? 4.f * (m0 + m1) : 8.f * (m0 - m1) + m2));
                const int co = cb * 16 + 4 * kq + r;
                ex[((wa * 64 + co) * 16 + tx) * 2 + bh] = z;
            }
        __syncthreads();
        if (reader) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + 512 * it, co = q >> 4, tl = q & 15;
                float zs[6];
#pragma unroll
                for (int aa = 0; aa < 6; ++aa) {
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(&ex[((aa * 64 + co) * 16 + tl) * 2]);
                    zs[aa] = v[0] + v[1];
                }
                const float s12 = zs[1] + zs[2], d12 = zs[1] - zs[2], s34 = zs[3] + zs[4], d34 = zs[3] - zs[4];
                yt[it][0][j] = zs[0] + s12 + s34;
                yt[it][1][j] = d12 + 2.f * d34;
                yt[it][2][j] = s12 + 4.f * s34;
                yt[it][3][j] = d12 + 8.f * d34 + zs[5];
            }
        }
        __syncthreads();
    }
    if (!reader) return;

    __amdgpu_buffer_rsrc_t ry = rx, ryp = rx, ryi = rx, rt = rx;
    if (GATE >= 1) {
        const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gate + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
#pragma unroll
        for (int it = 1; it < NIT; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gq[it][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg2, vo_it[it], (unsigned)(i * W * 4), 0));
    }
    if (a.y) ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.Cout * HW, 0, out_bytes, 0x00020000);
    if (GATE == 2) rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.addt + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
    const size_t HpWp = (size_t)Hp * Wp;
    if (EPI == 1) {
        ryp = __builtin_amdgcn_make_buffer_rsrc(a.yp + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp * 4), 0x00020000);
        if (a.yidx) ryi = __builtin_amdgcn_make_buffer_rsrc(a.yidx + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp), 0x00020000);
    }
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + 512 * it, col = q >> 4, tl = q & 15;
        const int oy = y0, ox = x0 + 4 * tl;
        const unsigned vo = vo_it[it];
        const bool inb = vo != kOob;
        const float bsum = bias_it[it];
        f32x4 rowv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = yt[it][i][j] + bsum;
                if (GATE == 0) v = __builtin_fmaxf(v, relu_floor);
                rowv[i][j] = v;
            }
            if (GATE >= 1) {
                const f32x4 g = gq[it][i];
                if (GATE == 2) {
                    const f32x4 tg = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, vo, (unsigned)(i * W * 4), 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // unfused multiply / add (the empty asm keeps the product out of an fma): bitwise what
                        // st3d_axpy_diff -- built without contraction -- adds
                        float m = a.addc * (g[j] - tg[j]);
                        asm volatile("" : "+v"(m));
                        rowv[i][j] = rowv[i][j] + m;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) rowv[i][j] = g[j] > 0.f ? rowv[i][j] : 0.f;
            }
            // (row offset in the VECTOR offset, scalar offset 0: with an SGPR scalar offset the compiler assumes the 16-byte
            //  store has read its data registers at issue and reuses them for the next row at once -- on gfx950 the last
            //  lanes of the store then picked up the next row's values now and then: rowv[1][0] came out as rowv[2][2])
            if (a.y) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rowv[i]), ry, inb ? vo + (unsigned)(i * W * 4) : kOob, 0, 0);
        }
        if (EPI == 1) {     // MaxPool2d(2,2): first maximum in row-major window order (ATen); four windows per 4x4 tile
            const unsigned vp = inb ? (unsigned)((((size_t)co0 + col) * HpWp + (size_t)(oy >> 1) * Wp + (ox >> 1)) * 4) : kOob;
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                f32x2 best; unsigned bidx = 0;
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const float w00 = rowv[2 * pi][2 * pj], w01 = rowv[2 * pi][2 * pj + 1];
                    const float w10 = rowv[2 * pi + 1][2 * pj], w11 = rowv[2 * pi + 1][2 * pj + 1];
                    float bv = w00; int bi = 0;
                    if (w01 > bv || w01 != w01) { bv = w01; bi = 1; }
                    if (w10 > bv || w10 != w10) { bv = w10; bi = 2; }
                    if (w11 > bv || w11 != w11) { bv = w11; bi = 3; }
                    best[pj] = bv; bidx |= (unsigned)(bi << (8 * pj));
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, best), ryp, vp, (unsigned)(pi * Wp * 4), 0);
                if (a.yidx) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bidx, ryi, vp == kOob ? kOob : vp / 4, (unsigned)(pi * Wp), 0);
            }
        }
    }
}

// w (Cout,Cin,3,3) -> U = G g G^T (6x6, fp64 -> fp32), forward and transposed (180-degree rotated filter, channel roles
// swapped), in the A-operand order of wino43_kernel for GEMM (M = out channel m, K = in channel k):
//   [ct = m/64][kstep = k/4][wave = 2 a + bh][lane = (m%16) + 16 (k%4)][bi*4 + cb]   with b = 3 bh + bi, cb = (m%64)/16
__device__ __forceinline__ size_t upack43_index(int m, int k, int aa, int bb, int K) {
    const int ct = m >> 6, col = m & 63, cb = col >> 4, m16 = col & 15;
    const int kstep = k >> 2, kq = k & 3;
    const int wave = 2 * aa + (bb >= 3 ? 1 : 0), bi = bb % 3;
    const int lane = m16 + 16 * kq;
    return ((((size_t)ct * (K >> 2) + kstep) * 12 + wave) * 64 + lane) * 12 + bi * 4 + cb;
}

__global__ void wino43_pack_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ uf, float *__restrict__ ud) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cout * Cin) return;
    const int co = i / Cin, ci = i % Cin;
    const float *g = w + i * 9;
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    for (int dir = 0; dir < 2; ++dir) {
        float *out = dir == 0 ? uf : ud;
        if (!out) continue;
        double gg[3][3];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) gg[ky][kx] = dir == 0 ? g[ky * 3 + kx] : g[(2 - ky) * 3 + (2 - kx)];
        double t[6][3];
        for (int aa = 0; aa < 6; ++aa)
            for (int kx = 0; kx < 3; ++kx) t[aa][kx] = G[aa][0] * gg[0][kx] + G[aa][1] * gg[1][kx] + G[aa][2] * gg[2][kx];
        for (int aa = 0; aa < 6; ++aa)
            for (int bb = 0; bb < 6; ++bb) {
                const double u = t[aa][0] * G[bb][0] + t[aa][1] * G[bb][1] + t[aa][2] * G[bb][2];
                const size_t o = dir == 0 ? upack43_index(co, ci, aa, bb, Cin) : upack43_index(ci, co, aa, bb, Cout);
                out[o] = (float)u;
            }
    }
}

bool shape_ok43(int Cin, int Cout, int H, int W) {
    return Cin >= 2 * KS6 && (Cin % KS6) == 0 && (Cout % 64) == 0 && (H % 4) == 0 && (W % 64) == 0 && H > 0 && W > 0 &&
           (unsigned long long)Cin * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31) &&
           (unsigned long long)Cout * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31);
}

template <int MODE>
int launch_wino43(Wino43Args a, hipStream_t s) {
    a.tiles_x = a.W / T6_COLS;
    a.tiles_y = a.H / T6_ROWS;
    a.n_ct = a.Cout / 64;
    const long blocks = (long)a.n_ct * a.tiles_x * a.tiles_y * a.N;
    constexpr size_t kSmem = (size_t)SMEM6 * sizeof(float);       // 81 KB of dynamic LDS: opt in once per instantiation
    auto go = [&](auto kernel) -> int {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem);
        if (attr != hipSuccess) { st3d::set_error("wino43: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr)); return ST3D_E_HIP; }
        kernel<<<(unsigned)blocks, NT6, kSmem, s>>>(a);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    };
    if (a.yp) {
        if (MODE != 0 || a.gate) { st3d::set_error("wino43: the fused pool belongs to the plain forward"); return ST3D_E_INVALID; }
        return go(wino43_kernel<0, 1, 0>);
    }
    if (a.gate && a.addt) {
        if (MODE != 0) { st3d::set_error("wino43: the content-target term rides on ungated input (MODE 0) only"); return ST3D_E_INVALID; }
        return go(wino43_kernel<0, 0, 2>);
    }
    if (a.gate) return go(wino43_kernel<MODE, 0, 1>);
    return go(wino43_kernel<MODE, 0, 0>);
}

}  // namespace

extern "C" int st3d_wino43_supported(int Cin, int Cout, int H, int W) { return shape_ok43(Cin, Cout, H, W) ? 1 : 0; }

extern "C" size_t st3d_wino43_packed_floats(int Cout, int Cin) { return (size_t)36 * Cout * Cin; }

extern "C" int st3d_wino43_pack(const float *w, int Cout, int Cin, float *u_fwd, float *u_dgrad, st3d_stream_t stream) {
    ST3D_CHECK_ARG(w && (u_fwd || u_dgrad));
    ST3D_CHECK_ARG(Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0);
    wino43_pack_kernel<<<st3d::cdiv((long)Cout * Cin, 256), 256, 0, st3d::as_stream(stream)>>>(w, Cout, Cin, u_fwd, u_dgrad);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

extern "C" int st3d_wino43_fwd(const float *x, const float *u_fwd, const float *bias, float *y, float *y_pooled,
                               uint8_t *pool_idx, int N, int Cin, int Cout, int H, int W, int relu, st3d_stream_t stream) {
    ST3D_CHECK_ARG(x && u_fwd && (y || y_pooled));
    ST3D_CHECK_ARG(N > 0 && shape_ok43(Cin, Cout, H, W));
    ST3D_CHECK_ARG(((uintptr_t)u_fwd & 15) == 0);
    Wino43Args a{x, nullptr, u_fwd, bias, y, y_pooled, pool_idx, N, Cin, Cout, H, W, relu, 0, 0, 0, nullptr, nullptr, 0.f};
    return launch_wino43<0>(a, st3d::as_stream(stream));
}

// gy: gradient w.r.t. the conv's output, ALREADY gated by its producer (or, with pool_idx, the pooled-resolution gradient
// of the pool behind the conv, gated at pooled resolution); out_gate / add_target / add_coef as st3d_wino_dgrad_chain.
extern "C" int st3d_wino43_dgrad_chain(const float *gy, const uint8_t *pool_idx, const float *u_dgrad, const float *out_gate,
                                       const float *add_target, float add_coef, float *gx, int N, int Cin, int Cout, int H,
                                       int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok43(Cout, Cin, H, W));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0 && ((uintptr_t)out_gate & 15) == 0 && ((uintptr_t)add_target & 15) == 0);
    ST3D_CHECK_ARG(!add_target || out_gate);
    Wino43Args a{gy, pool_idx, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0, out_gate, add_target, add_coef};
    hipStream_t s = st3d::as_stream(stream);
    return pool_idx ? launch_wino43<3>(a, s) : launch_wino43<0>(a, s);
}
