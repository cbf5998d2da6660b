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
// Mapping (wino4_kernel, the one Winograd kernel of the shipped library): one 256-thread workgroup = 64 cout x 32 tiles
// (4 rows x 32 columns of pixels), two workgroups per CU.  Wave w accumulates row a = w of the 4x4 Winograd domain
// (xi = 4a..4a+3) for both 32-cout halves: 8 MFMA tiles = 128 accumulator VGPRs.  A operands (U) never touch LDS -- the
// pack stores every lane's operands of a 4-channel sub-chunk contiguously, fetched two sub-chunks ahead through a buffer
// descriptor; B operands (V) are computed by each lane from the haloed patch, staged 8 channels per barrier through a
// 3-deep LDS ring.  The output transform runs along b in registers and along a through one LDS exchange; after it a lane
// owns 2x2 output tiles, which ARE the MaxPool2d(2,2) windows, so pooling (+argmax) fuses into the epilogue, as do the
// ReLU gate / max-unpool of the backward pass (into the patch staging or, producer-side, into the output store).
// Details at the kernel.  (The round-1 kernel -- 512 threads, 64 x 64 tiles -- lives in lab/wino8.inc, lab builds only.)
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 4;                 // input channels per MFMA sub-chunk (2 k-steps of 2)
constexpr int BCO = 64;               // cout per workgroup
constexpr int PCP = 48;                          // LDS row pitch: 2 rows apart = 96 words = 32 banks (mod 64),
                                                 // so the two tile rows of a half-wave never share a bank

struct WinoArgs {
    const float *x;       // MODE 0/1: (N,Cin,H,W); MODE 2: pooled-resolution gradient (N,Cin,H/2,W/2)
    const float *aux;     // MODE 1: saved post-ReLU activation; MODE 2: pooled values
    const uint8_t *idx;   // MODE 2: pool argmax
    const float *U;       // packed [ct][chunk][wave 8][lane 64][ks 2][q 4]  (see wino_pack_kernel)
    const float *bias;    // (Cout) or nullptr
    float *y;             // (N,Cout,H,W) or nullptr (EPI 1 may skip the full-resolution store)
    float *yp;            // EPI 1: pooled output (N,Cout,H/2,W/2)
    uint8_t *yidx;        // EPI 1: argmax
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_y, n_ct;
    int tiles_per_xcd;    // wino4_kernel: pixel tiles owned by one XCD
    unsigned long long *dbg;   // diagnostic builds only (DBG != 0): per-wave phase cycle sums
    const float *gate;    // wino4_kernel: (N,Cout,H,W) or nullptr; outputs are zeroed where gate <= 0 (the consumer's ReLU gate,
                          // applied by the producer so that the consumer streams ONE operand: see st3d_wino_dgrad_chain)
    const float *addt;    // with gate: outputs become gate > 0 ? y + addc * (gate - addt) : 0 -- the content-loss gradient
    float addc;           // addc * (activation - target) (losses.py:24-28) lands in the same store; nullptr = none
};

#ifdef ST3D_LAB
#include "lab/wino8.inc"      // the retired 8-wave kernel: lab builds only (A/B runs, phase stamps)
#endif

// ------------------------------------------------------------------------------------------------------------------
// wino4_kernel: the same algorithm on a HALF-SIZE workgroup -- 256 threads = 4 waves, 64 cout x 32 tiles (4 x 32
// pixels) -- so that TWO workgroups share a CU (2 waves per SIMD as before, 32 KB of LDS each).  They are independent:
// one's prologue (two dependent HBM round trips) and epilogue (LDS exchange + stores) run under the other's MFMA loop,
// which the single 512-thread workgroup per CU of wino_kernel cannot do (28 % of a tile's time on the 8-stage Cin = 64
// layers).  Wave w accumulates row a = w of the 4x4 Winograd domain for BOTH 32-cout halves over ONE 32-tile MFMA
// n-tile: 2 (halves) x 4 (b) = 8 MFMA tiles = 128 accumulator VGPRs.  Every B operand (V = B^T d B, computed by the lane
// from the staged patch) now feeds two MFMAs instead of one, halving the VALU + LDS-read work per MFMA; the price is
// twice the A-operand (U) loads per MFMA (4 dwordx4 per 16 MFMAs) and a 1.59x instead of 1.33x halo on the patch.
// Same pack layout (pack-waves a and a + 4 are this wave's two halves), same gates, same epilogue algebra.
constexpr int T4_ROWS = 4, T4_COLS = 32;
constexpr int NT4 = 256;
constexpr int PR4 = T4_ROWS + 2;                  // 6 patch rows
constexpr int PS4 = PR4 * PCP;                    // 288 floats per channel
#ifndef ST3D_WINO_KS
#define ST3D_WINO_KS 8
#endif
constexpr int KS4 = ST3D_WINO_KS;                 // input channels staged per barrier: 8 (default) or 16 = 2 or 4 MFMA sub-chunks
                                                  // (16 measured: one barrier per 64 MFMAs, but the prologue doubles -- conv1_2
                                                  //  +11 %, conv4_x +1.5 %, 256 VGPRs with a spill: not taken; 4 -- one sub-chunk per
                                                  //  barrier, operand parity alternating per stage -- was also measured: +4 % overall)
constexpr int NSUB4 = KS4 / KC;
constexpr int P4_STAGE = KS4 * PS4 + 8;           // floats per stage
constexpr int EX4_FLOATS = 2 * 4 * 64 * 32;       // [2 j][4 a][64 co][32 tiles]
constexpr int SMEM4_FLOATS = EX4_FLOATS > 3 * P4_STAGE ? EX4_FLOATS : 3 * P4_STAGE;      // 64 KB: two workgroups per CU use 128 of its 160 KB

// GATE: 0 = plain outputs, 1 = outputs zeroed where a.gate <= 0, 2 = a.addc * (a.gate - a.addt) added first (EPI 0 only)
// MH: 32-cout halves per workgroup.  2 (what ships): 64 cout, 128 accumulators, two workgroups per CU.  1 (round 3, the
// short-K experiment, lab builds only): 32 cout, 64 accumulators (149-157 VGPRs) and a 32 KB exchange, so THREE workgroups
// fit a CU -- two can be inside their K loops while the third is between tiles -- at the price of one MFMA per B operand
// instead of two.  Measured 7-12 % slower on every VGG shape (see launch_wino4).
template <int MODE, int EPI, int DBG = 0, int GATE = 0, int MH = 2>
__global__ __launch_bounds__(NT4, MH == 1 ? 3 : 2) void wino4_kernel(const WinoArgs a) {
    constexpr int BCOH = 32 * MH;              // cout per workgroup
    constexpr int EXF = 2 * 4 * BCOH * 32;     // exchange floats [2 j][4 a][BCOH co][32 tiles]
    constexpr int SMEMF = EXF > 3 * P4_STAGE ? EXF : 3 * P4_STAGE;
    __shared__ __attribute__((aligned(16))) float smem[SMEMF];
    float *sP = smem;                          // [3][KS4][PR4][PCP]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wa = tid >> 6;  // wave = row a of the Winograd domain
    const int l31 = lane & 31, lhi = lane >> 5;

    // grid (XCD-aware, see launch_wino4): blocks are dealt round-robin over the 8 XCDs, so bid & 7 names the XCD.  Each
    // XCD owns a contiguous run of pixel tiles and walks (cout tile fastest, then pixel tile): the cout tiles of one pixel
    // tile -- which read the same patch -- and neighbouring pixel tiles -- which share halo rows -- hit the same L2.
    int ct, pt;
    if (a.tiles_per_xcd > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        ct = j % a.n_ct;
        pt = xcd * a.tiles_per_xcd + j / a.n_ct;
        if (j / a.n_ct >= a.tiles_per_xcd || pt >= a.tiles_x * a.tiles_y * a.N) return;  // padding blocks (whole workgroup)
    } else {                                   // ST3D_WINO_MAP=rr (A/B runs): cout tile fastest, pixel tiles dealt round-robin
        ct = blockIdx.x % a.n_ct;
        pt = blockIdx.x / a.n_ct;
    }
    const int tile_x = pt % a.tiles_x; pt /= a.tiles_x;
    const int tile_y = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    int x0 = tile_x * T4_COLS, y0 = tile_y * T4_ROWS;
    const int co0 = ct * BCOH;
    if (DBG == 2) { x0 = 32; y0 = 8; }       // diagnostic: every workgroup reads the same (cache-resident) patch
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    constexpr bool UNPOOL = MODE == 2 || MODE == 3;     // MODE 3: the pooled gradient arrives already gated (pooled value > 0)
    const size_t in_plane = UNPOOL ? (size_t)Hp * Wp : HW;
    const int nstages = a.Cin / KS4;

    // staging: 8 channels x 6 rows x 10 sixteen-byte items = 480 items per stage, two per thread
    constexpr int ITEMS = KS4 * PR4 * 10, IPT = KS4 / 4;
    const unsigned kOob = 0x80000000u;
    unsigned voff[IPT];
    int loff[IPT];
    unsigned rowbit[UNPOOL ? IPT : 1];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int e = tid + i * NT4;
        const int ci = e / (PR4 * 10), rem = e - ci * (PR4 * 10);
        const int r = rem / 10, l = rem - r * 10;
        const int gy = (DBG == 9 ? 8 : y0) + r - 1, gx0 = (DBG == 9 ? 32 : x0) - 4 + 4 * l;
        const bool ok = e < ITEMS && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
        if (UNPOOL) {
            voff[i] = ok ? (unsigned)((ci * in_plane + (size_t)(gy >> 1) * Wp + (gx0 >> 1)) * 4) : kOob;
            rowbit[i] = (gy & 1) << 1;
        } else {
            voff[i] = ok ? (unsigned)((ci * in_plane + (size_t)gy * W + gx0) * 4) : kOob;
        }
        // the 32 threads without a second item write theirs (zeros: voff is out of range) into the stage's trailing slack:
        // no branch in lstore, so the stage loop is one basic block the scheduler can interleave
        loff[i] = (e < ITEMS) ? (4 + ci * PS4 + r * PCP + 4 * l - 3) : (KS4 * PS4 + 4);
    }
    const unsigned img_bytes = (unsigned)((size_t)a.Cin * in_plane * 4);
    const unsigned stage_bytes = (unsigned)(KS4 * in_plane * 4);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x + (size_t)n * a.Cin * in_plane), 0, img_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t raux = rx, ridx = rx;
    if (MODE == 1 || MODE == 2)
        raux = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.aux + (size_t)n * a.Cin * in_plane), 0, img_bytes, 0x00020000);
    if (UNPOOL)
        ridx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.idx + (size_t)n * a.Cin * in_plane), 0, img_bytes / 4, 0x00020000);
    const int nsub = a.Cin / KC;
    // (the pack groups 64 couts: a 32-cout workgroup reads half `ct & 1` of group ct >> 1)
    const int ct64 = MH == 2 ? ct : (ct >> 1), uhalf = MH == 2 ? 0 : (ct & 1);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.U + (size_t)ct64 * nsub * 4096), 0, (unsigned)((size_t)nsub * 4096 * 4), 0x00020000);
    const unsigned uvoff0 = (unsigned)(((wa + 4 * uhalf) * 64 + lane) * 32), uvoff1 = (unsigned)(((wa + 4) * 64 + lane) * 32);

    f32x4 xv[IPT];
    f32x4 xa[MODE == 1 ? IPT : 1];
    f32x2 xg[UNPOOL ? IPT : 1], xp[MODE == 2 ? IPT : 1];
    unsigned xi[UNPOOL ? IPT : 1];

    auto gload = [&](int st) __attribute__((always_inline)) {
        const unsigned so = (unsigned)st * stage_bytes;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            if (UNPOOL) {
                xg[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, voff[i], so, 0));
                if (MODE == 2) xp[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(raux, voff[i], so, 0));
                xi[i] = __builtin_amdgcn_raw_buffer_load_b16(ridx, voff[i] == kOob ? kOob : voff[i] / 4, so / 4, 0);
            } else {
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff[i], so, 0));
                if (MODE == 1 && DBG != 6) xa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, voff[i], so, 0));
                if (MODE == 1 && DBG == 6) xa[i] = xv[i];      // diagnostic: no second load stream
            }
        }
    };
    // lgate: the staged items' values after their gate (ReLU / pool routing) -- VALU work; lwrite: into the LDS ring.  Kept
    // apart so the stage loop can place them in different quarters (lstore = both, for the prologue).
    struct Vals { float v[IPT][4]; };
    auto lgate = [&](Vals &o) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v;
                if (MODE == 0) v = xv[i][jj];
                if (MODE == 1) v = (xa[i][jj] > 0.f) ? xv[i][jj] : 0.f;
                if (MODE == 2) {
                    const unsigned ib = (xi[i] >> (8 * (jj >> 1))) & 0xffu;
                    v = (xp[i][jj >> 1] > 0.f && ib == (rowbit[i] | (jj & 1))) ? xg[i][jj >> 1] : 0.f;
                }
                if (MODE == 3) {
                    const unsigned ib = (xi[i] >> (8 * (jj >> 1))) & 0xffu;
                    v = (ib == (rowbit[i] | (jj & 1))) ? xg[i][jj >> 1] : 0.f;
                }
                o.v[i][jj] = v;
            }
        }
    };
    auto lwrite = [&](int buf, const Vals &o) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            float *dst = &sP[buf * P4_STAGE + loff[i]];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dst[jj] = o.v[i][jj];
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        Vals o;
        lgate(o);
        lwrite(buf, o);
    };

    // B operands: row transform of the wave's row a (t = d[r1] + sg * d[r2]), then the column transform per b
    const int r1 = (wa == 0) ? 0 : (wa == 2 ? 2 : 1);
    const int r2 = (wa == 3) ? 3 : (wa == 2 ? 1 : 2);
    const float sg = (wa == 1) ? 1.f : -1.f;
    const int lane_off = lhi * PS4 + (2 * (l31 >> 4)) * PCP + 2 * (l31 & 15);
    const int off1 = lane_off + r1 * PCP, off2 = lane_off + r2 * PCP;
    struct Raw { f32x2 v[2][2][2]; };          // [ks][row 1/2][column pair]
    auto pread = [&](int buf, int sub, Raw &d) __attribute__((always_inline)) {
        const float *p = &sP[buf * P4_STAGE + 4 + sub * KC * PS4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float *q1 = p + off1 + (2 * ks) * PS4;
            const float *q2 = p + off2 + (2 * ks) * PS4;
            d.v[ks][0][0] = *reinterpret_cast<const f32x2 *>(q1);
            d.v[ks][0][1] = *reinterpret_cast<const f32x2 *>(q1 + 2);
            d.v[ks][1][0] = *reinterpret_cast<const f32x2 *>(q2);
            d.v[ks][1][1] = *reinterpret_cast<const f32x2 *>(q2 + 2);
        }
    };
    struct Bop { float v[2][4]; };             // [ks][q = b]
    auto bcompute = [&](const Raw &d, Bop &bv) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float t0 = d.v[ks][0][0][0] + sg * d.v[ks][1][0][0];
            const float t1 = d.v[ks][0][0][1] + sg * d.v[ks][1][0][1];
            const float t2 = d.v[ks][0][1][0] + sg * d.v[ks][1][1][0];
            const float t3 = d.v[ks][0][1][1] + sg * d.v[ks][1][1][1];
            bv.v[ks][0] = t0 - t2;
            bv.v[ks][1] = t1 + t2;
            bv.v[ks][2] = t2 - t1;
            bv.v[ks][3] = t1 - t3;
        }
    };

    // [cout half][b].  Never zeroed: the first eight MFMAs of the tile (stage 0, sub-chunk 0, k-step 0) take the constant 0
    // as their C operand (W4_MFMA0) -- 128 v_mov per lane and tile less, all of them paid in matrix-pipe time (round 3)
    f32x16 acc[MH][4];

    struct Uop { f32x4 h[MH][2]; };            // [cout half][ks] -> 4 floats (q)
    auto uload = [&](int sub, Uop &u) __attribute__((always_inline)) {
        const unsigned so = (unsigned)min(sub, nsub - 1) * 16384u;
        u.h[0][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff0, so, 0));
        u.h[0][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff0 + 16, so, 0));
        if (MH == 2) {
            u.h[MH - 1][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff1, so, 0));
            u.h[MH - 1][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff1 + 16, so, 0));
        }
    };

    gload(0);
    lstore(0);
    gload(nstages > 1 ? 1 : 0);
    lstore(1);
    // ST3D_WINO_UDEPTH (compile time): sub-chunks the filter operands are requested ahead of their MFMAs.  2 (three operand
    // sets, 248-255 VGPRs, the default) against 1 (two sets): forward -1.2 %, conv1_2 forward -2.7 % -- 8 TB/s of L2 reads
    // over all CUs are these operands, and one sub-chunk (16 MFMAs ~ 0.5 us) did not always cover their latency.
#ifndef ST3D_WINO_UDEPTH
#define ST3D_WINO_UDEPTH 2
#endif
#if ST3D_WINO_UDEPTH == 2
    Uop u3[3];                 // filter operands two sub-chunks ahead (three sets)
    uload(0, u3[0]);
    uload(1, u3[1]);
#else
    Uop ua, ub;
    uload(0, ua);
#endif
    __syncthreads();
    if (DBG == 8 || DBG == 9) {       // diagnostics: from here on (8: the loop) the patch loads hit one cache-resident tile / (9) are real again
        const int fx = DBG == 8 ? 32 : x0, fy = DBG == 8 ? 8 : y0;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int e = tid + i * NT4;
            const int ci = e / (PR4 * 10), rem = e - ci * (PR4 * 10);
            const int r = rem / 10, l = rem - r * 10;
            const int gy = fy + r - 1, gx0 = fx - 4 + 4 * l;
            const bool ok = e < ITEMS && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
            voff[i] = ok ? (unsigned)((ci * in_plane + (size_t)gy * W + gx0) * 4) : kOob;
        }
    }
    Raw draw;
    Bop bcur, bnext;
    pread(0, 0, draw);
    bcompute(draw, bcur);

#define W4_MFMA(u, bv, ks)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
        _Pragma("unroll") for (int h_ = 0; h_ < MH; ++h_)                                                       \
            acc[h_][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.h[h_][ks][q], bv.v[ks][q], acc[h_][q], 0, 0, 0); \
    }
#define W4_MFMA0(u, bv)                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
        const f32x16 zero_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           \
        _Pragma("unroll") for (int h_ = 0; h_ < MH; ++h_)                                                       \
            acc[h_][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(u.h[h_][0][q], bv.v[0][q], zero_, 0, 0, 0);        \
    }
    // one stage = 8 input channels = sub-chunks s0 (operands ready in ua / bcur) and s1, as four quarters of 8 MFMAs.  Each
    // quarter's memory / LDS / VALU work (which prepares LATER quarters) is INTERLEAVED with its MFMAs by
    // sched_group_barrier -- one matrix instruction, then one to three of the others -- so the pipe stays fed even while the
    // co-resident workgroup is between tiles and this wave is alone on its SIMD (round 2: 2 % over the ten VGG shapes, 3-4 %
    // on the 8-/16-stage layers, against the same quarters issued as bursts of 8 MFMAs followed by the other work).  The
    // stage loop is one basic block (lstore has no branch) so that the scheduler may do this.
#define W4_ILV(mask, per)                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(mask, per, 0);                           \
    }
    // ST3D_WINO_SCHED (compile-time, tools/wino_sched_ab.sh): 0 = one memory instruction per MFMA in the two memory quarters
    // (default), 1 = two.  Measured and dropped: three per MFMA (-1 %), halves of 16 MFMAs as one region with the VALU work
    // spread over them (-2 %), LDS reads before the global loads (=), s_setprio around the stage (=), the stage's barrier
    // moved behind the first quarter's MFMAs instead of the stage end (legal for the 3-deep ring; -0.5 %).
#ifndef ST3D_WINO_SCHED
#define ST3D_WINO_SCHED 0
#endif
    // The ring position is a compile-time constant of each copy of the stage (the loop is unrolled by the ring depth, 3):
    // every LDS address is the lane's base plus an immediate, no per-stage address arithmetic on the vector ALU.
    auto stage = [&](int c, auto PBc, auto FIRSTc) __attribute__((always_inline)) {
        constexpr int pb = decltype(PBc)::value, pb1 = (pb + 1) % 3, pb2 = (pb + 2) % 3;
        constexpr bool first = decltype(FIRSTc)::value;     // the tile's first stage: its first MFMAs start the accumulators
        Vals staged;
#pragma unroll
        for (int sc = 0; sc < NSUB4; ++sc) {          // sub-chunk sc: its operands are ready in (ua | ub) / (bcur | bnext) by parity
#if ST3D_WINO_UDEPTH == 2
            constexpr int sub0 = (NSUB4 * pb) % 3;         // (the loop is unrolled by 3 stages: sub-chunk index mod 3 is static)
            Uop &ucur = u3[(sub0 + sc) % 3];
            Uop &unext = u3[(sub0 + sc + 2) % 3];
#else
            Uop &ucur = (sc & 1) ? ub : ua;
            Uop &unext = (sc & 1) ? ua : ub;
#endif
            Bop &bc = (sc & 1) ? bnext : bcur;
            Bop &bn = (sc & 1) ? bcur : bnext;
            // memory quarter: the k-steps 0 of the sub-chunk || the patch items of stage c + 2 (first sub-chunk), the next
            // sub-chunk's filter operands and patch reads (the last one reads the NEXT stage's buffer, staged a barrier ago),
            // and in the last sub-chunk the gates of the items that arrived meanwhile
            __builtin_amdgcn_sched_barrier(0);
            if (first && sc == 0) { W4_MFMA0(ucur, bc) } else { W4_MFMA(ucur, bc, 0) }
            if (sc == 0) gload(min(c + 2, nstages - 1));
            uload(NSUB4 * c + sc + ST3D_WINO_UDEPTH, unext);
            if (sc + 1 < NSUB4) pread(pb, sc + 1, draw);
            else pread(pb1, 0, draw);
            if (sc == NSUB4 - 1) lgate(staged);
            _Pragma("unroll") for (int i_ = 0; i_ < 4 * MH; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x120, (ST3D_WINO_SCHED == 0 ? 1 : 2) * (3 - MH), 0);
                if (MODE != 0 && sc == NSUB4 - 1) __builtin_amdgcn_sched_group_barrier(0x002, (KS4 / 2) * (3 - MH), 0);
            }
            // transform quarter: the k-steps 1 || the next sub-chunk's B operands (VALU), the ring writes in the last one
            __builtin_amdgcn_sched_barrier(0);
            W4_MFMA(ucur, bc, 1)
            bcompute(draw, bn);
            if (sc == NSUB4 - 1) lwrite(pb2, staged);
            _Pragma("unroll") for (int i_ = 0; i_ < 4 * MH; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2 * (3 - MH), 0);
                if (sc == NSUB4 - 1) __builtin_amdgcn_sched_group_barrier(0x200, (KS4 / 8) * (3 - MH), 0);       // LDS writes
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    constexpr std::false_type kRest{};
    stage(0, std::integral_constant<int, 0>{}, std::true_type{});
    int c = 1;                                     // ring position of stage c is c % 3: the unrolled loop starts at 1
    for (; c + 3 <= nstages; c += 3) {
        stage(c, std::integral_constant<int, 1>{}, kRest);
        stage(c + 1, std::integral_constant<int, 2>{}, kRest);
        stage(c + 2, std::integral_constant<int, 0>{}, kRest);
    }
    if (c < nstages) stage(c, std::integral_constant<int, 1>{}, kRest);
    if (c + 1 < nstages) stage(c + 1, std::integral_constant<int, 2>{}, kRest);
#undef W4_ILV
#undef W4_MFMA
#undef W4_MFMA0

    if (DBG == 1) {                          // diagnostic: no epilogue (keeps the accumulators live with one store)
        float t = 0.f;
#pragma unroll
        for (int mh = 0; mh < MH; ++mh)
#pragma unroll
            for (int q = 0; q < 4; ++q) t += acc[mh][q][0];
        if (t == 12345.678f && a.y) a.y[0] = t;
        return;
    }
    // ---- epilogue: Y = A^T M A.  Along b in registers (z_0 = m0 + m1 + m2, z_1 = m1 - m2 - m3 of this wave's row a),
    // along a through ONE LDS exchange [2 j][4 a][64 co][32 tiles] (64 KB: both output columns at once, one barrier).
    // Readers own TWO horizontally adjacent 2x2 tiles of one cout = 4 consecutive pixels on two rows, so the exchange is
    // read with ds_read_b64 and the image written with 16-byte stores (8 store instructions per lane instead of 16
    // dwordx2: the tail of a tile is store-issue bound).
    float *ex = smem;
    // this thread's four biases are fetched NOW: a load issued between the stores below would have to wait for every
    // older store to complete (vmcnt counts loads and stores in order), serialising the four output items
    constexpr int NIT = 2 * MH;            // output items per thread: 16 couts each
    float bias4[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) bias4[it] = (a.bias && DBG != 4) ? a.bias[co0 + it * 16 + (tid >> 4)] : 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int mh = 0; mh < MH; ++mh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = jj == 0 ? (acc[mh][0][r] + acc[mh][1][r] + acc[mh][2][r])
                                        : (acc[mh][1][r] - acc[mh][2][r] - acc[mh][3][r]);
                const int co = mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                ex[((jj * 4 + wa) * BCOH + co) * 32 + l31] = z;
            }
    const int pair = tid & 15;                        // tiles 2 * pair, 2 * pair + 1 (same tile row)
    const int oy = y0 + 2 * (pair >> 3), ox = x0 + 4 * (pair & 7);
    const bool inb = oy < H && ox < W;                // H even, W % 4 == 0: the 2 x 4 pixel item is inside or outside
    // Outputs (and the gate / content-target tiles) are addressed through buffer descriptors of image n with 32-bit byte
    // offsets: the lane's offset once, the item's channel step as the scalar offset, out-of-image lanes at the sentinel --
    // no 64-bit address arithmetic and no branch per item (the epilogue's vector-ALU work is paid in the neighbour
    // workgroup's matrix-pipe time).
    const unsigned out_bytes = (unsigned)((size_t)a.Cout * HW * 4);
    const unsigned vo0 = inb ? (unsigned)((((size_t)co0 + (tid >> 4)) * HW + (size_t)oy * W + ox) * 4) : kOob;
    const unsigned vo1 = inb ? vo0 + (unsigned)W * 4u : kOob;
    const unsigned item_bytes = (unsigned)(16 * HW * 4);
    // the consumer's ReLU gate, applied here (a.gate: the tensor this gradient belongs to): fetched now -- the accumulators
    // are dead, and a load issued between the stores below would queue behind them -- and used after the exchange
    f32x4 gq[GATE >= 1 ? NIT : 1][2], tq[GATE == 2 ? NIT : 1][2];
    if (GATE >= 1) {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.gate + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            gq[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo0, it * item_bytes, 0));
            gq[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo1, it * item_bytes, 0));
        }
    }
    if (GATE == 2) {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.addt + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            tq[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, vo0, it * item_bytes, 0));
            tq[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, vo1, it * item_bytes, 0));
        }
    }
    __amdgpu_buffer_rsrc_t ry = rx, ryp = rx, ryi = rx;
    if (a.y) ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.Cout * HW, 0, out_bytes, 0x00020000);
    const size_t HpWp = (size_t)Hp * Wp;
    unsigned vp = kOob;
    if (EPI == 1) {
        ryp = __builtin_amdgcn_make_buffer_rsrc(a.yp + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp * 4), 0x00020000);
        if (a.yidx) ryi = __builtin_amdgcn_make_buffer_rsrc(a.yidx + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp), 0x00020000);
        if (inb) vp = (unsigned)((((size_t)co0 + (tid >> 4)) * HpWp + (size_t)(oy >> 1) * Wp + (ox >> 1)) * 4);
    }
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();      // ReLU as max(v, floor) with a wave-uniform floor
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = it * 16 + (tid >> 4);         // cout within the workgroup's 32 * MH
        f32x2 z[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int aa = 0; aa < 4; ++aa)
                z[jj][aa] = *reinterpret_cast<const f32x2 *>(&ex[((jj * 4 + aa) * BCOH + col) * 32 + 2 * pair]);
        const float bsum = bias4[it];
        // y[i][jj] per tile t: i = 0: z0 + z1 + z2, i = 1: z1 - z2 - z3 (along a)
        float y[2][2][2];      // [tile][row i][col jj]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                float v0 = z[jj][0][t] + z[jj][1][t] + z[jj][2][t] + bsum;
                float v1 = z[jj][1][t] - z[jj][2][t] - z[jj][3][t] + bsum;
                if (GATE == 0) { v0 = __builtin_fmaxf(v0, relu_floor); v1 = __builtin_fmaxf(v1, relu_floor); }     // one v_max each (gated launches are input-gradients: no ReLU)
                y[t][0][jj] = v0; y[t][1][jj] = v1;
            }
        if (DBG == 3 && y[0][0][0] != 12345.678f) continue;       // diagnostic: whole epilogue but no global stores
        if (a.y) {
            f32x4 q0, q1;
            q0[0] = y[0][0][0]; q0[1] = y[0][0][1]; q0[2] = y[1][0][0]; q0[3] = y[1][0][1];
            q1[0] = y[0][1][0]; q1[1] = y[0][1][1]; q1[2] = y[1][1][0]; q1[3] = y[1][1][1];
            if (GATE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // unfused multiply / add: bitwise what st3d_axpy_diff (built without contraction) adds
                    q0[e] = __fadd_rn(q0[e], __fmul_rn(a.addc, __fsub_rn(gq[it][0][e], tq[it][0][e])));
                    q1[e] = __fadd_rn(q1[e], __fmul_rn(a.addc, __fsub_rn(gq[it][1][e], tq[it][1][e])));
                }
            }
            if (GATE >= 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q0[e] = gq[it][0][e] > 0.f ? q0[e] : 0.f;
                    q1[e] = gq[it][1][e] > 0.f ? q1[e] : 0.f;
                }
            }
            // the item's channel step goes into the VECTOR offset and the scalar offset stays 0: with an SGPR scalar offset
            // the compiler assumes a 16-byte store has read its data registers at issue and may reuse them at once; on
            // gfx950 the last lanes of such a store were seen to pick up the next values (round 3, csrc/wino43.hip)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q0), ry, inb ? vo0 + it * item_bytes : kOob, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q1), ry, inb ? vo1 + it * item_bytes : kOob, 0, 0);
        }
        if (EPI == 1) {     // MaxPool2d(2,2): first maximum in row-major window order (ATen); two adjacent windows per lane
            f32x2 best; unsigned bidx = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bv = y[t][0][0]; int bi = 0;
                if (y[t][0][1] > bv || y[t][0][1] != y[t][0][1]) { bv = y[t][0][1]; bi = 1; }
                if (y[t][1][0] > bv || y[t][1][0] != y[t][1][0]) { bv = y[t][1][0]; bi = 2; }
                if (y[t][1][1] > bv || y[t][1][1] != y[t][1][1]) { bv = y[t][1][1]; bi = 3; }
                best[t] = bv; bidx |= (unsigned)(bi << (8 * t));
            }
            const unsigned sp = (unsigned)(it * 16 * HpWp * 4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, best), ryp, vp, sp, 0);
            if (a.yidx) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bidx, ryi, vp == kOob ? kOob : vp / 4, sp / 4, 0);
        }
    }
}

// w (Cout,Cin,3,3) -> Winograd-domain filters U = G g G^T (fp64, stored fp32), forward and
// transposed (g'[ky][kx] = w[co][ci][2-ky][2-kx], channel roles swapped), laid out as the
// MFMA A operands of wino_kernel: for GEMM (M = out channel m, K = in channel k, xi):
//   [ct = m/64][chunk = k/4][wave = (xi>>2) + 4*((m%64)/32)][lane = (m%32) + 32*(k&1)][ks = (k%4)>>1][q = xi&3]
// so a lane's 8 operands of a chunk are 32 contiguous bytes and a wave's are 2 KB.
__device__ __forceinline__ size_t upack_index(int m, int k, int xi, int K) {
    const int ct = m >> 6, col = m & 63, mh = col >> 5, l31 = col & 31;
    const int chunk = k >> 2, kk = k & 3, lhi = kk & 1, ks = kk >> 1;
    const int wave = (xi >> 2) + 4 * mh, lane = l31 + 32 * lhi;
    return ((((size_t)ct * (K >> 2) + chunk) * 8 + wave) * 64 + lane) * 8 + ks * 4 + (xi & 3);
}

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
            const double u[4] = {t[aa][0], 0.5 * (t[aa][0] + t[aa][1] + t[aa][2]), 0.5 * (t[aa][0] - t[aa][1] + t[aa][2]),
                                 t[aa][2]};
            for (int bb = 0; bb < 4; ++bb) {
                const int xi = aa * 4 + bb;
                // forward: M = cout, K = cin ; dgrad: M = cin (the output of the transposed conv), K = cout
                const size_t o = dir == 0 ? upack_index(co, ci, xi, Cin) : upack_index(ci, co, xi, Cout);
                out[o] = (float)u[bb];
            }
        }
    }
}

#ifdef ST3D_LAB
// lab builds: ST3D_WINO_VARIANT=8 selects the retired 8-wave kernel (A/B runs, tools/wino_layers.py)
int wino_variant(const WinoArgs &a) {
    static const int forced = [] { const char *e = getenv("ST3D_WINO_VARIANT"); return e ? atoi(e) : 0; }();
    if (forced == 4 || forced == 8) return forced;
    return 4;
}
#endif

template <int MODE>
int launch_wino4(WinoArgs a, hipStream_t s) {
    a.tiles_x = st3d::cdiv(a.W, T4_COLS);
    a.tiles_y = st3d::cdiv(a.H, T4_ROWS);
#ifdef ST3D_LAB
    // lab builds: ST3D_WINO_MH1_MAXK=k runs layers with at most k input channels on 32-cout workgroups, three per CU
    // (MH = 1).  Measured in round 3 (tools/wino_layers.py) and NOT taken: 7-12 % slower on every VGG shape, conv1_2
    // included -- one MFMA per B operand doubles the vector-ALU work per MFMA, which costs more than the third workgroup hides.
    static const int mh1_maxk = [] { const char *e = getenv("ST3D_WINO_MH1_MAXK"); return e ? atoi(e) : 0; }();
    const bool mh1 = a.Cin <= mh1_maxk;
#else
    constexpr bool mh1 = false;
#endif
    a.n_ct = a.Cout / (mh1 ? 32 : BCO);
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.N;
    // block -> (pixel tile, cout tile): XCD-chunked for the full-resolution inputs (measured 1-6 % faster than dealing pixel
    // tiles round-robin: halo rows and the cout tiles' shared patch hit the XCD's L2), round-robin for the fused-unpool
    // input-gradient, whose pooled-resolution operands are a quarter of the size (measured 2-4 % faster there).
    // ST3D_WINO_MAP=rr / xcd forces one mapping (A/B runs).
    static const int forced = [] { const char *e = getenv("ST3D_WINO_MAP"); return !e ? 0 : (strcmp(e, "rr") == 0 ? 1 : 2); }();
    const bool rr = forced ? forced == 1 : (MODE == 2 || MODE == 3);
    a.tiles_per_xcd = rr ? 0 : (int)((ntiles + 7) / 8);
    const long blocks = rr ? ntiles * a.n_ct : 8L * a.tiles_per_xcd * a.n_ct;   // bid & 7 = XCD, bid >> 3 = (pixel tile in the XCD's run, cout tile)
#ifdef ST3D_WINO_DEBUG
    static const int dbgmode = [] { const char *e = getenv("ST3D_WINO_DBGMODE"); return e ? atoi(e) : 0; }();
    if (dbgmode == 1 && MODE == 0) { wino4_kernel<0, 0, 1><<<(unsigned)blocks, NT4, 0, s>>>(a); return ST3D_OK; }
    if (dbgmode == 6 && MODE == 1) { wino4_kernel<1, 0, 6><<<(unsigned)blocks, NT4, 0, s>>>(a); return ST3D_OK; }
    if ((dbgmode == 3 || dbgmode == 4) && MODE == 0) {
        if (dbgmode == 3) { if (a.yp) wino4_kernel<0, 1, 3><<<(unsigned)blocks, NT4, 0, s>>>(a); else wino4_kernel<0, 0, 3><<<(unsigned)blocks, NT4, 0, s>>>(a); }
        else { if (a.yp) wino4_kernel<0, 1, 4><<<(unsigned)blocks, NT4, 0, s>>>(a); else wino4_kernel<0, 0, 4><<<(unsigned)blocks, NT4, 0, s>>>(a); }
        return ST3D_OK;
    }
    if ((dbgmode == 8 || dbgmode == 9) && MODE == 0) {
        if (dbgmode == 8) { if (a.yp) wino4_kernel<0, 1, 8><<<(unsigned)blocks, NT4, 0, s>>>(a); else wino4_kernel<0, 0, 8><<<(unsigned)blocks, NT4, 0, s>>>(a); }
        else { if (a.yp) wino4_kernel<0, 1, 9><<<(unsigned)blocks, NT4, 0, s>>>(a); else wino4_kernel<0, 0, 9><<<(unsigned)blocks, NT4, 0, s>>>(a); }
        return ST3D_OK;
    }
    if (dbgmode == 2 && MODE == 0) { if (a.yp) wino4_kernel<0, 1, 2><<<(unsigned)blocks, NT4, 0, s>>>(a); else wino4_kernel<0, 0, 2><<<(unsigned)blocks, NT4, 0, s>>>(a); return ST3D_OK; }
#endif
#ifdef ST3D_LAB
    if (mh1) {          // 32-cout workgroups, three per CU
        if (a.yp) wino4_kernel<MODE, 1, 0, 0, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);
        else if (a.gate && a.addt) {
            if (MODE != 0) { st3d::set_error("wino4: the content-target term rides on ungated input (MODE 0) only"); return ST3D_E_INVALID; }
            wino4_kernel<0, 0, 0, 2, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);
        } else if (a.gate) wino4_kernel<MODE, 0, 0, 1, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);
        else wino4_kernel<MODE, 0, 0, 0, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
#endif
    if (a.yp) wino4_kernel<MODE, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);        // (forward: never gated)
    else if (a.gate && a.addt) {
        if (MODE != 0) { st3d::set_error("wino4: the content-target term rides on ungated input (MODE 0) only"); return ST3D_E_INVALID; }
        wino4_kernel<0, 0, 0, 2><<<(unsigned)blocks, NT4, 0, s>>>(a);
    } else if (a.gate) wino4_kernel<MODE, 0, 0, 1><<<(unsigned)blocks, NT4, 0, s>>>(a);
    else wino4_kernel<MODE, 0><<<(unsigned)blocks, NT4, 0, s>>>(a);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

template <int MODE>
int launch_wino(WinoArgs a, hipStream_t s) {
#ifndef ST3D_LAB
    return launch_wino4<MODE>(a, s);
#else
    if ((wino_variant(a) == 4 && !a.dbg) || a.gate) return launch_wino4<MODE>(a, s);      // the output gate exists in wino4 only
    a.tiles_x = st3d::cdiv(a.W, TCOLS);
    a.tiles_y = st3d::cdiv(a.H, TROWS);
    a.n_ct = a.Cout / BCO;
    const long blocks = (long)a.n_ct * a.tiles_x * a.tiles_y * a.N;
#ifdef ST3D_WINO_DEBUG
    if (a.dbg && MODE == 0 && !a.yp) wino_kernel<0, 0, 1><<<(unsigned)blocks, NT, 0, s>>>(a);
    else
#endif
    if (a.yp) wino_kernel<MODE, 1><<<(unsigned)blocks, NT, 0, s>>>(a);
    else wino_kernel<MODE, 0><<<(unsigned)blocks, NT, 0, s>>>(a);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
#endif
}

// The patch loads address one image through a buffer descriptor with 32-bit BYTE offsets (voff, stage_bytes) and
// 0x80000000 as the out-of-image sentinel, so one image of the INPUT operand must stay below 2^31 bytes
// (Cin*H*W < 2^29 floats); above that the callers fall back to the direct kernels of conv.hip (64-bit addressing).
bool shape_ok(int Cin, int Cout, int H, int W) {
    return Cin >= KS4 && (Cin % KS4) == 0 && (Cout % 64) == 0 && (H % 2) == 0 && (W % 4) == 0 && H > 0 && W > 0 &&
           (unsigned long long)Cin * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31) &&
           (unsigned long long)Cout * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31);      // outputs: same addressing
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
    ST3D_CHECK_ARG(((uintptr_t)u_fwd & 15) == 0);
    unsigned long long *dbg = nullptr;
#ifdef ST3D_WINO_DEBUG      // diagnostic builds only (tools/wino_bench.py): device pointer for the s_memtime stamps
    if (const char *e = getenv("ST3D_WINO_STAMP")) dbg = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0));
#endif
    WinoArgs a{x, nullptr, nullptr, u_fwd, bias, y, y_pooled, pool_idx, N, Cin, Cout, H, W, relu, 0, 0, 0, 0, dbg};
    return launch_wino<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_wino_dgrad(const float *gy, const float *act, const float *u_dgrad, float *gx, int N, int Cin, int Cout,
                               int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cout, Cin, H, W));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0);
    WinoArgs a{gy, act, nullptr, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0, 0, nullptr};
    return act ? launch_wino<1>(a, st3d::as_stream(stream)) : launch_wino<0>(a, st3d::as_stream(stream));
}

extern "C" int st3d_wino_dgrad_chain(const float *gy, const float *act, const uint8_t *pool_idx, const float *pooled,
                                     const float *u_dgrad, const float *out_gate, const float *add_target, float add_coef,
                                     float *gx, int N, int Cin, int Cout, int H, int W, st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cout, Cin, H, W));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0 && ((uintptr_t)out_gate & 15) == 0 && ((uintptr_t)add_target & 15) == 0);
    ST3D_CHECK_ARG(!(pool_idx && act));            // a pooled gradient is gated by `pooled`, a full-resolution one by `act`
    ST3D_CHECK_ARG(!add_target || out_gate);       // the added term is addc * (out_gate - add_target)
    WinoArgs a{gy, pool_idx ? pooled : act, pool_idx, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0, 0,
               nullptr, out_gate, add_target, add_coef};
    hipStream_t s = st3d::as_stream(stream);
    if (pool_idx) return pooled ? launch_wino4<2>(a, s) : launch_wino4<3>(a, s);
    return act ? launch_wino<1>(a, s) : launch_wino<0>(a, s);
}

extern "C" int st3d_wino_dgrad_unpool(const float *gy_pooled, const uint8_t *pool_idx, const float *pooled,
                                      const float *u_dgrad, float *gx, int N, int Cin, int Cout, int H, int W,
                                      st3d_stream_t stream) {
    ST3D_CHECK_ARG(gy_pooled && pool_idx && pooled && u_dgrad && gx);
    ST3D_CHECK_ARG(N > 0 && shape_ok(Cout, Cin, H, W));
    ST3D_CHECK_ARG(((uintptr_t)u_dgrad & 15) == 0);
    WinoArgs a{gy_pooled, pooled, pool_idx, u_dgrad, nullptr, gx, nullptr, nullptr, N, Cout, Cin, H, W, 0, 0, 0, 0, 0, nullptr};
    return launch_wino<2>(a, st3d::as_stream(stream));
}
