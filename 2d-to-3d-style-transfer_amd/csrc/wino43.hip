// wino43.hip -- conv3x3(pad 1) forward / input-gradient as Winograd F(4x4,3x3) on the gfx950 fp32 matrix pipe
// (round 3): 36 multiplies per 4x4 output tile = 2.25 per output pixel, against 4 for the F(2x2,3x3) of wino.hip and 9
// for the direct convolution.  Still fp32 products and fp32 accumulation; the larger transform constants cost about one
// decimal digit (measured <= 1.3e-5 of the output scale at K = 512 against an fp64 convolution; F(2x2,3x3): 1e-6), inside
// the 3e-5 the Winograd kernels are tested to.  Used for every layer with Cin >= 64 and W % 64 == 0 (or W % 32 == 0 with
// H % 8 == 0: conv5_1 at 512 x 512); other maps stay on wino4_kernel.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A       per (cout, 4x4 tile), 6x6 Winograd domain xi = (a, b)
//
// Mapping.  One 768-thread workgroup (12 waves, three per SIMD, one workgroup per CU) = 64 cout x 16 tiles (4 rows x 64
// columns of pixels, or 8 x 32: Geo43).  Wave w = (a = w >> 1, bh = w & 1) accumulates row a of the domain for b in {3 bh .. 3 bh + 2} and
// all 64 couts on v_mfma_f32_16x16x4_f32 (M = 16 cout, N = 16 tiles, K = 4 input channels): 3 b x 4 cout blocks = 12
// accumulator tiles = 48 VGPRs, and every B operand feeds FOUR MFMAs.
//   * B operands V = B^T d B.  The ROW half of the transform (T_a = sum_r B^T[a][r] d[r], 13 VALU per column for all six
//     a) is done ONCE per patch column by the staging threads on its way into LDS -- the ring holds T[ci][a][column], not
//     the raw patch -- so a wave reads only ITS row a (6 floats per (tile, channel)) and does the column half for its
//     three b (6-7 VALU).  That is the same ~1 vector-ALU instruction per 64 MFMA cycles in the waves' loop as
//     wino4_kernel has, for 2.25 instead of 4 MFMA-multiplies per output.
//   * A operands U = G g G^T (fp64 -> fp32, packed [wave][b][lane][4 cout blocks] per 4-channel k-step: three 16-byte loads
//     per lane, each 1 KB contiguous per wave) come straight from L2 through a buffer descriptor, three k-steps ahead; they
//     never touch LDS.
//   * Staging: 16 channels per stage, 3-deep ring, one barrier per stage.  A half-item = (channel, column pair): six row
//     loads -> gate (MODE) -> row transform of both columns in packed fp32 math -> six 8-byte LDS stores.  576 half-items
//     per stage on 768 threads: every wave stages the same amount every stage, interleaved with its MFMAs -- with one
//     workgroup per CU a wave that works alone behind its MFMAs (staging in turns was tried first) holds up all twelve at
//     the barrier.
//   * PERSISTENT workgroups, one per CU: workgroup (cout tile, slot) walks the pixel tiles slot, slot + nslots, ...; the
//     stream of stages runs on from one tile into the next (the ring holds the next tile's first two stages when a tile's
//     last MFMA issues), the exchange region is separate from the ring, and only the epilogue stands between two tiles.
//   * Epilogue: along b in registers (partial over the wave's three b), then four LDS exchange passes = (cout half) x
//     (output-column pair) through [6 a][32 co][16 tiles][2 bh][2 columns], packed fp32 math on both sides; a reader (the
//     first 8 waves) owns one whole 4x4 output tile per half, so the 2x2 max-pool (+argmax), ReLU, the producer-side gates
//     of the backward chain and 16-byte row stores all happen in registers, as in wino4_kernel.  Outputs and gate reads are
//     non-temporal; slots are numbered XCD-major (see wino43_body).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KS6 = 16;                  // input channels per stage (four k-steps of 4): one barrier and one staging turn per 48 MFMAs
constexpr int NK6 = KS6 / 4;
constexpr int EX6 = 6 * 32 * 16 * 2 * 2;         // exchange floats per pass: [a 6][co 32][tile 16][bh 2][column 2]
// Tile geometry.  A workgroup step covers 16 4x4 output tiles laid out TC across x TR = 16 / TC down:
//   TC = 16: 4 x 64 pixels (W % 64 == 0: conv1_2 .. conv4_4 at 512^2), TC = 8: 8 x 32 pixels (W % 32 == 0, H % 8 == 0: conv5_1 at
//   512^2, conv4_x at the reference's default 768^2).  The ring holds, per channel and tile row, the six row-transformed rows
//   of the patch columns x0 - 4 .. x0 + 4 TC + 3, stored from index 1 of a PITCH-float row: a tile's six columns 4 t + 3 .. + 8
//   then start 16-byte aligned (one 16-byte + one 8-byte LDS read).
template <int TC>
struct Geo43 {
    static constexpr int TR = 16 / TC;
    static constexpr int ROWS = 4 * TR, COLS = 4 * TC;       // output pixels per workgroup step
    static constexpr int PAIRS = 2 * TC + 4;                 // column pairs per patch row
    static constexpr int PITCH = 4 * TC + 12;                // 76 / 44
    static constexpr int CHS = TR * 6 * PITCH;               // floats per channel: 456 / 528
    static constexpr int TSTAGE = KS6 * CHS;                 // floats per ring stage: 29 / 34 KB
    static constexpr int HALF_ITEMS = KS6 * TR * PAIRS;      // staging half-items (channel, tile row, column pair) per stage: 576 / 640 of 768 threads
    static constexpr int SMEM = 3 * TSTAGE + EX6;            // 134 / 147 KB of dynamic LDS: the ring AND the exchange region -- a persistent
                                                             // workgroup keeps staging its next tile while the finished one is written out
    static_assert(HALF_ITEMS <= 768 && SMEM * 4 <= 160 * 1024, "tile geometry does not fit the workgroup");
};
constexpr int NT6 = 768;                 // 12 waves

struct Wino43Args {
    const float *x;       // MODE 0: (N,Cin,H,W); MODE 3: pooled-resolution gradient (N,Cin,H/2,W/2), already gated
    const uint8_t *idx;   // MODE 3: pool argmax
    const float *U;       // packed [ct][kstep][wave 12][b 3][lane 64][cb 4]  (wino43_pack_kernel)
    const float *bias;    // (Cout) or nullptr
    float *y;             // (N,Cout,H,W) or nullptr (EPI 1 may skip the full-resolution store)
    float *yp;            // EPI 1: pooled output (N,Cout,H/2,W/2)
    uint8_t *yidx;        // EPI 1: argmax
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_y, n_ct;
    const float *gate;    // (N,Cout,H,W) or nullptr: outputs are zeroed where gate <= 0 (the consumer's ReLU gate, see wino.hip)
    const float *addt;    // with gate: outputs become gate > 0 ? y + addc * (gate - addt) : 0
    float addc;
    int xpc;                        // XCDs per cout tile (slots are renumbered XCD-major, see wino43_body); 1 = as dealt
    unsigned magic_x, magic_y;      // floor(2^32 / tiles_x) + 1, likewise tiles_y: pix / tiles_x = umulhi(pix, magic_x) for pix * tiles_x < 2^32 (tiles_x >= 2)
};

// T_a = sum_r B^T[a][r] d[r] for a = 0..5 (and, applied to t0..t5, the column half V_b = sum_c B^T[b][c] t[c])
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
// (T = f32x2: two patch columns at once in packed fp32 math -- v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: 14 instructions
//  for both columns; the fp32 MFMA does not co-execute with the vector ALU)
template <typename T>
__device__ __forceinline__ void bt6(T d0, T d1, T d2, T d3, T d4, T d5, T &t0, T &t1, T &t2, T &t3, T &t4, T &t5) {
    const T p = d4 - 4.f * d2, q = d3 - 4.f * d1;
    t1 = p + q; t2 = p - q;
    t0 = 4.f * d0 + (d4 - 5.f * d2);
    const T r = d4 - d2, s = 2.f * (d3 - d1);
    t3 = r + s; t4 = r - s;
    t5 = 4.f * d1 + (d5 - 5.f * d3);
}

// MODE 0: plain input.  MODE 3: input = 2x2 max-unpool of a pooled-resolution gradient (routing by the argmax bytes).
// EPI 1: MaxPool2d(2,2) (+argmax) fused into the epilogue.  GATE as in wino4_kernel (0 none, 1 output gate, 2 + content term).
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 5           // diagnostic build: gate reads out of range, stores in place
#define W43_GATE_VO(v) (a.N >= 0 ? kOob : (v))
#else
#define W43_GATE_VO(v) (v)
#endif

// (BH = the wave's half of the domain's columns, a template parameter: the kernel branches ONCE per wave into the body of its
//  half, so the column transform of every k-step and the b-direction of the epilogue are straight-line code in the same basic
//  block as the k-step's MFMAs instead of two scalar branches behind them)
template <int MODE, int EPI, int GATE, int BH, int TC>
__device__ __forceinline__ void wino43_body(const Wino43Args &a, float *smem) {
    using G = Geo43<TC>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: every role test below is a scalar branch)
    const int wa = wave >> 1;                           // row a of the domain
    constexpr int bh = BH;                              // b half
    const int tx = lane & 15, kq = lane >> 4;           // MFMA n index (tile) / k index (channel within the k-step)

    // PERSISTENT workgroups (one per CU): workgroup (ct, slot) walks the pixel tiles slot, slot + nslots, ... of ONE cout tile,
    // so its filter operands are the same 36 x 64 x K slab for every tile and the stream of stages simply continues from one
    // tile into the next: the ring already holds the next tile's first two stages when a tile's last MFMA issues, the
    // operand requests wrap around, and only the output exchange stands between two tiles (no prologue, no relaunch).
    // XCD-MAJOR SLOTS.  Workgroups are dealt round-robin over the 8 XCDs, so with the slots in tile order each XCD gets every
    // xpc-th tile of a row of tiles: horizontal neighbours, which share the 16-byte slivers at both ends of every 288-byte
    // patch row (4 cache lines touched for 2.25 lines of data), sit in DIFFERENT L2s and both fetch both lines (measured:
    // 1.5 - 1.8x the compulsory L2 miss traffic at conv1_2 .. conv3_x).  Renumbered, an XCD's slots of one round are a
    // contiguous run of tiles -- whole rows of tiles, several deep: slivers and halo rows hit in its own L2.
    const int ct = blockIdx.x % a.n_ct, nslots = gridDim.x / a.n_ct;
    int slot = blockIdx.x / a.n_ct;
    if (a.xpc > 1) slot = (slot % a.xpc) * (nslots / a.xpc) + slot / a.xpc;
    const int npix = a.tiles_x * a.tiles_y * a.N;
    const int co0 = ct * 64;
    int n = 0, x0 = 0, y0 = 0;                 // the tile whose patch columns are being STAGED (runs two stages ahead)
    auto decode = [&](int pix, int &tn, int &ty0, int &tx0) __attribute__((always_inline)) {
        // (scalar multiply-high by the host's reciprocals: a runtime integer division costs ~30 vector instructions, and every
        //  instruction between two tiles' MFMAs is exposed)
        const int r2 = a.tiles_x == 1 ? pix : (int)__umulhi((unsigned)pix, a.magic_x);
        const int txi = pix - r2 * a.tiles_x;
        tn = a.tiles_y == 1 ? r2 : (int)__umulhi((unsigned)r2, a.magic_y);
        tx0 = txi * G::COLS; ty0 = (r2 - tn * a.tiles_y) * G::ROWS;
    };
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int Hp = H >> 1, Wp = W >> 1;
    constexpr bool UNPOOL = MODE == 3;
    const size_t in_plane = UNPOOL ? (size_t)Hp * Wp : HW;
    const int nstages = a.Cin / KS6;
    const unsigned kOob = 0x80000000u;

    // ---- staging: EVERY thread stages one half-item per stage = (channel ci, tile row tr, column pair): rows gy = y0 - 1 +
    // 4 tr + r, columns x0 - 4 + s_col, + 1.  576 (640) half-items on 768 threads: all twelve waves carry the same staging
    // work, so it sits in the same basic block as their MFMAs (interleaved, no tail behind them) and nobody is late at the barrier.
    const int s_ci = tid / (G::TR * G::PAIRS), s_rem = tid - s_ci * (G::TR * G::PAIRS);
    const int s_tr = s_rem / G::PAIRS;
    const int s_col = 2 * (s_rem - s_tr * G::PAIRS);     // column offset inside the patch row
    const bool s_on = tid < G::HALF_ITEMS;
    unsigned voff[6];
    unsigned rowbit[UNPOOL ? 6 : 1];
#pragma unroll
    for (int r = 0; r < 6; ++r)
        if (UNPOOL) rowbit[r] = (unsigned)((r + 1) & 1) << 1;        // gy = y0 - 1 + 4 tr + r with y0 % 4 == 0: odd for even r
    const unsigned img_bytes = (unsigned)((size_t)a.Cin * in_plane * 4);
    __amdgpu_buffer_rsrc_t rx, ridx;
    // addresses of the staged tile `pix` (>= npix: past this workgroup's last tile -- everything out of range, loads return 0)
    auto stage_tile = [&](int pix) __attribute__((always_inline)) {
        const bool live = pix < npix;
        decode(live ? pix : 0, n, y0, x0);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int gy = y0 - 1 + 4 * s_tr + r, gx0 = x0 - 4 + s_col;
            const bool ok = live && s_on && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
            if (UNPOOL) voff[r] = ok ? (unsigned)((s_ci * in_plane + (size_t)(gy >> 1) * Wp + (gx0 >> 1)) * 4) : kOob;   // one pooled element covers the row's two columns
            else voff[r] = ok ? (unsigned)((s_ci * in_plane + (size_t)gy * W + gx0) * 4) : kOob;
        }
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x + (size_t)n * a.Cin * in_plane), 0, img_bytes, 0x00020000);
        ridx = rx;
        if (UNPOOL)
            ridx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.idx + (size_t)n * a.Cin * in_plane), 0, img_bytes / 4, 0x00020000);
    };
    int spix = slot;                           // pixel tile being staged
    stage_tile(spix);
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 10       // 10: staging loads out of range (issued, no memory traffic)
    if (a.N >= 0) spix = npix;
#endif
    const int loff = s_on ? (s_ci * G::CHS + s_tr * 6 * G::PITCH + s_col + 1) : 0;      // + a * PITCH per transformed row (odd index: two 4-byte stores in one ds_write2)
    const unsigned stage_bytes = (unsigned)(KS6 * in_plane * 4);
    const int nksteps = a.Cin / 4;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.U + (size_t)ct * nksteps * (12 * 64 * 12)), 0, (unsigned)((size_t)nksteps * 12 * 64 * 12 * 4), 0x00020000);
    const unsigned uvoff = (unsigned)(wave * 3072 + lane * 16);

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
        f32x2 t[6], d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (UNPOOL) {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) d[r][j2] = ((x.i[r] & 0xffu) == (rowbit[r] | (unsigned)j2)) ? x.g[r] : 0.f;
            } else {
                d[r] = x.v[r];
            }
        }
        bt6<f32x2>(d[0], d[1], d[2], d[3], d[4], d[5], t[0], t[1], t[2], t[3], t[4], t[5]);
        float *dst = &smem[buf * G::TSTAGE + loff];
        if (s_on) {
#pragma unroll
            for (int aa = 0; aa < 6; ++aa) { dst[aa * G::PITCH] = t[aa][0]; dst[aa * G::PITCH + 1] = t[aa][1]; }
        }
    };

    // ---- B operands: this lane's row-a values of (tile tx, channel 4 kk + kq): columns 4 tx + 3 .. 4 tx + 8 of the strip row
    const int tbase = kq * G::CHS + ((tx / TC) * 6 + wa) * G::PITCH + 4 * (tx % TC);
    struct Trow { f32x4 m; f32x2 n; };       // t0..t3, t4..t5
    auto tread = [&](int buf, int kk, Trow &o) __attribute__((always_inline)) {
        const float *p = &smem[buf * G::TSTAGE + kk * (4 * G::CHS) + tbase];
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
        // (a tile's last three requests run past its slab: clamped and never used -- the next tile's first three sets are
        //  requested from the epilogue, where they need no registers across the exchange passes)
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 1       // diagnostic build: every k-step reads the SAME (cache-resident) filter operands -- wrong results, timing only
        const unsigned so = 0u;
#else
        const unsigned so = (unsigned)min(kstep, nksteps - 1) * (unsigned)(12 * 64 * 12 * 4);
#endif
#pragma unroll
        for (int b = 0; b < 3; ++b) u.q[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvoff + 1024 * b, so, 0));
    };

    f32x4 acc[3][4];       // [b][cout block]
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging.  A CU's vector loads come back in issue order ACROSS its waves: a patch row that comes from HBM holds up
    // every filter operand requested after it, by any wave.  Measured on the nine config-2 layers (tools/w43_ab_run.sh, sum of
    // forward + input gradient): all twelve waves request in k-step 0 BEHIND that k-step's operand request 6.75 ms; in
    // front of it 6.95; the three waves of a SIMD in different k-steps 7.3 (three hold-ups per stage instead of one); the
    // request a k-step earlier / the store a stage later through a ring of four 6.85 - 6.95; every request at the head of its
    // k-step instead of behind the MFMAs 7.08.  With the loads out of range (no memory traffic) the sum is 6.1, with no
    // staging at all 5.8: the round trip of the patch rows is worth 10 % (20 % at conv1_2, whose input comes from HBM).
    // Workgroups of different slots started a fraction of a stage apart (so that the CUs do not all request in the same
    // microsecond): no change at any delay.  Filter operands FOUR k-steps ahead (into the set just consumed): 6.65 against
    // 6.42 -- more requests in flight make it worse, not better.
    Staged xs;
    // ---- prologue (once per workgroup): stages 0 and 1 of its first tile, both requests in flight together
    {
        Staged x0s, x1s;
        gload(0, x0s);
        gload(1, x1s);
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

    // One stage = 16 input channels = four k-steps of 12 MFMAs, one barrier; k-step kk uses filter-operand set kk and
    // requests the set of three k-steps ahead.  pb = ring position of the stage being computed, sc = the stage (of the
    // staged tile) requested in k-step 0 and stored, row-transformed, in k-step 3.
#define W6_MFMAS(u, v)                                                                                            \
    _Pragma("unroll") for (int b = 0; b < 3; ++b)                                                                 \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                          \
            acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.q[b][cb], v[b], acc[b][cb], 0, 0, 0);
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 6        // 6: no staging in the loop
#define W6_GLOAD(sc) ((void)0)
#define W6_LSTORE(buf) ((void)0)
#elif defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 11     // 11: staging loads waited for, no transform / LDS stores
#define W6_GLOAD(sc) gload(sc, xs)
#define W6_LSTORE(buf) do { for (int r = 0; r < (UNPOOL ? 1 : 6); ++r) asm volatile("" :: "v"(xs.v[r])); if (UNPOOL) for (int r = 0; r < 6; ++r) asm volatile("" :: "v"(xs.g[r]), "v"(xs.i[r])); } while (0)
#else
#define W6_GLOAD(sc) gload(sc, xs)
#define W6_LSTORE(buf) lstore(buf, xs)
#endif
    auto stage = [&](int c, int sc, int pb) __attribute__((always_inline)) {
        const int pb1 = pb == 2 ? 0 : pb + 1, pb2 = pb1 == 2 ? 0 : pb1 + 1;
#pragma unroll
        for (int kk = 0; kk < NK6; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            // the next k-step's row (the last one prefetches k-step 0 of the NEXT stage: staged a barrier ago)
#if !(defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 9)        // 9: no B-operand reads / column transform in the loop
            if (kk + 1 < NK6) tread(pb, kk + 1, trow); else tread(pb1, 0, trow);
#endif
            // (B operands ping-pong between vcur / vnext by k-step parity: NK6 is even, so every stage starts on vcur)
            if (kk & 1) { W6_MFMAS(u4[kk], vnext) } else { W6_MFMAS(u4[kk], vcur) }
#if !(defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 7)        // 7: no filter-operand loads in the loop
            uload(NK6 * c + kk + 3, u4[(kk + 3) & 3]);
#endif
            if (kk == 0) W6_GLOAD(sc);
#if !(defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 9)
            if (kk & 1) vcompute(trow, vcur); else vcompute(trow, vnext);
#else
            if (kk & 1) { vcur[0] = vnext[1]; vcur[1] = vnext[2]; vcur[2] = vnext[0]; } else { vnext[0] = vcur[1]; vnext[1] = vcur[2]; vnext[2] = vcur[0]; }
#endif
            if (kk == 3) W6_LSTORE(pb2);
            __builtin_amdgcn_sched_barrier(0);
        }
#if !(defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 8)        // 8: no barrier between stages
        __syncthreads();
#endif
    };
    // ---- epilogue.  Along b (this wave's three b): z_j = sum_b A^T[j][b] m_b, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    //   bh 0 (b 0 1 2): z0 = m0 + s, z1 = z3 = d, z2 = s            with s = m1 + m2, d = m1 - m2
    //   bh 1 (b 3 4 5): z0 = s, z1 = 2 d, z2 = 4 s, z3 = 8 d + m5   with s = m3 + m4, d = m3 - m4
    // Four exchange passes = (cout half h) x (output column pair jp) through [a 6][co 32][tile 16][bh 2][column 2]: a reader
    // (512 threads: one 4x4 output tile of the half each) reads (bh 0: z_j z_j+1, bh 1: z_j z_j+1) in ONE 16-byte read, adds
    // the halves and applies A^T along a on the column PAIR -- packed fp32 math on both sides (writers: pairs of couts), a
    // quarter of the vector instructions of one-column passes with per-lane selects.  A half's tile is complete after its
    // two passes and leaves at once: 16 output registers, and the gate of the next half travels under its passes.
    float *ex = smem + 3 * G::TSTAGE;                // its own region: the ring keeps streaming the next tile while a tile is written out
    const bool reader = tid < 512;
    auto epilogue = [&](int n, int y0, int x0) __attribute__((always_inline)) {        // (the COMPUTED tile's coordinates)
    const unsigned out_bytes = (unsigned)((size_t)a.Cout * HW * 4);
    const size_t HpWp = (size_t)Hp * Wp;
    const int col_l = tid >> 4, tl = tid & 15;     // reader: cout within the half, tile
    const int oy = y0 + 4 * (tl / TC), ox = x0 + 4 * (tl % TC);
    __amdgpu_buffer_rsrc_t rg = ru, ry = ru, ryp = ru, ryi = ru, rt = ru;
    if (GATE >= 1) rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.gate + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
    if (a.y) ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.Cout * HW, 0, out_bytes, 0x00020000);
    if (GATE == 2) rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.addt + (size_t)n * a.Cout * HW), 0, out_bytes, 0x00020000);
    if (EPI == 1) {
        ryp = __builtin_amdgcn_make_buffer_rsrc(a.yp + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp * 4), 0x00020000);
        if (a.yidx) ryi = __builtin_amdgcn_make_buffer_rsrc(a.yidx + (size_t)n * a.Cout * HpWp, 0, (unsigned)(a.Cout * HpWp), 0x00020000);
    }
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();
    // the reader's bias and the consumer's ReLU gate (GATE >= 1) of a half are requested two passes before they are used
    // (one workgroup per CU: an exposed HBM round trip idles the whole CU)
    unsigned vo = kOob;
    float bsum = 0.f;
    f32x4 gq[GATE >= 1 ? 4 : 1];
    auto request = [&](int h) __attribute__((always_inline)) {
        const int col = 32 * h + col_l;
        const bool inb = reader && oy < H && ox < W;   // H, W multiples of the step: every reader (kept for the descriptor sentinel)
        vo = inb ? (unsigned)((((size_t)co0 + col) * HW + (size_t)oy * W + ox) * 4) : kOob;
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 4       // diagnostic build: gate reads and output stores out of range (issued, no memory traffic)
        if (a.N >= 0) vo = kOob;
#endif
        bsum = (reader && a.bias) ? a.bias[co0 + col] : 0.f;
        if (GATE >= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, W43_GATE_VO(vo), (unsigned)(i * W * 4), 2));
        }
    };
    request(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 rowv[4];                             // [row i] x 4 columns of this reader's tile
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
            for (int cbi = 0; cbi < 2; ++cbi) {
                const int cb = 2 * h + cbi;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 m0 = f32x2{acc[0][cb][2 * hh], acc[0][cb][2 * hh + 1]};
                    const f32x2 m1 = f32x2{acc[1][cb][2 * hh], acc[1][cb][2 * hh + 1]};
                    const f32x2 m2 = f32x2{acc[2][cb][2 * hh], acc[2][cb][2 * hh + 1]};
                    f32x2 za, zb;                  // columns 2 jp, 2 jp + 1 for couts r = 2 hh, 2 hh + 1
                    if (bh == 0) {
                        asm volatile("");          // (a real scalar branch: both arms computed and selected per lane otherwise)
                        const f32x2 sm = m1 + m2, dm = m1 - m2;
                        za = jp == 0 ? m0 + sm : sm;
                        zb = dm;
                    } else {
                        asm volatile("");
                        const f32x2 sm = m0 + m1, dm = m0 - m1;
                        za = jp == 0 ? sm : 4.f * sm;
                        zb = jp == 0 ? 2.f * dm : 8.f * dm + m2;
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int col = cbi * 16 + 4 * kq + 2 * hh + rr;
                        // (tiles 8..15 keep their halves swapped: sixteen lanes' 8-byte stores then cover all 32 banks;
                        //  the reader adds the halves, in either order)
                        float *pz = &ex[((wa * 32 + col) * 16 + tx) * 4 + 2 * (bh ^ (tx >> 3))];
                        pz[0] = za[rr];
                        pz[1] = zb[rr];
                    }
                }
            }
            __syncthreads();
            if (reader) {
                f32x2 zs[6];
#pragma unroll
                for (int aa = 0; aa < 6; ++aa) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(&ex[((aa * 32 + col_l) * 16 + tl) * 4]);
                    zs[aa] = f32x2{v[0], v[1]} + f32x2{v[2], v[3]};
                }
                const f32x2 s12 = zs[1] + zs[2], d12 = zs[1] - zs[2], s34 = zs[3] + zs[4], d34 = zs[3] - zs[4];
                const f32x2 r0 = zs[0] + s12 + s34, r1 = d12 + 2.f * d34, r2 = s12 + 4.f * s34, r3 = d12 + 8.f * d34 + zs[5];
                rowv[0][2 * jp] = r0[0]; rowv[0][2 * jp + 1] = r0[1];
                rowv[1][2 * jp] = r1[0]; rowv[1][2 * jp + 1] = r1[1];
                rowv[2][2 * jp] = r2[0]; rowv[2][2 * jp + 1] = r2[1];
                rowv[3][2 * jp] = r3[0]; rowv[3][2 * jp + 1] = r3[1];
            }
            __syncthreads();
        }
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 3       // diagnostic build: exchange passes, no global loads / stores -- wrong results, timing only
        if (a.N < 0) { float sacc = 0.f; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sacc += rowv[i][j]; ex[tid] = sacc; }
        if (h == 1) { uload(0, u4[0]); uload(1, u4[1]); uload(2, u4[2]); }
        continue;
#endif
        // the accumulators are dead: the next tile's first three filter-operand sets travel under the last stores
        if (h == 1) { uload(0, u4[0]); uload(1, u4[1]); uload(2, u4[2]); }
        if (reader) {
            const int col = 32 * h + col_l;
            const bool inb = vo != kOob;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = rowv[i][j] + bsum;
                    if (GATE == 0) v = __builtin_fmaxf(v, relu_floor);
                    rowv[i][j] = v;
                }
                if (GATE >= 1) {
                    const f32x4 g = gq[i];
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
                // Non-temporal stores (and gate reads, above): what a tile writes is read next by ANOTHER kernel, out of the
                // memory-side cache at best -- kept out of L2 it leaves the XCD's 4 MB to the filter operands and the patch rows
                // the cout tiles share (-2.6 % on the nine-layer sum; non-temporal PATCH loads, which lose that sharing: +12 %).
                // (row offset in the VECTOR offset, scalar offset 0: with an SGPR scalar offset the compiler assumes the 16-byte
                //  store has read its data registers at issue and reuses them for the next row at once -- on gfx950 the last
                //  lanes of the store then picked up the next row's values now and then: rowv[1][0] came out as rowv[2][2])
                if (a.y) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rowv[i]), ry, inb ? vo + (unsigned)(i * W * 4) : kOob, 0, 2);
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
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, best), ryp, vp, (unsigned)(pi * Wp * 4), 2);
                    if (a.yidx) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bidx, ryi, vp == kOob ? kOob : vp / 4, (unsigned)(pi * Wp), 2);
                }
            }
        }
        if (h == 0) request(1);
    }
    };      // epilogue

    // ---- the stream: tile after tile of this workgroup's slot
    int pb = 0;
    for (int cpix = slot; cpix < npix; cpix += nslots) {
        for (int c = 0; c < nstages; ++c) {
            int sc = c + 2;
            if (sc == nstages) { spix += nslots; stage_tile(spix); }      // the staging runs on into the next tile's patch
            if (sc >= nstages) sc -= nstages;
            stage(c, sc, pb);
            pb = pb == 2 ? 0 : pb + 1;
        }
        int cn, cy0, cx0;
        decode(cpix, cn, cy0, cx0);
#if defined(ST3D_W43_DIAG) && ST3D_W43_DIAG == 2       // diagnostic build: no epilogue at all -- no results, timing only
        if (a.N < 0) { float sacc = 0.f; for (int b = 0; b < 3; ++b) for (int cb = 0; cb < 4; ++cb) for (int r = 0; r < 4; ++r) sacc += acc[b][cb][r]; ex[tid] = sacc; }
        uload(0, u4[0]); uload(1, u4[1]); uload(2, u4[2]);
#else
        epilogue(cn, cy0, cx0);
#endif
        // (the staged tile's addresses are recomputed rather than carried through the epilogue's register peak)
        asm volatile("" : "+s"(spix));
        stage_tile(spix);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int MODE, int EPI, int GATE, int TC>
__global__ __launch_bounds__(NT6, 3) void wino43_kernel(const Wino43Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) wino43_body<MODE, EPI, GATE, 1, TC>(a, smem);
    else wino43_body<MODE, EPI, GATE, 0, TC>(a, smem);
}

// w (Cout,Cin,3,3) -> U = G g G^T (6x6, fp64 -> fp32), forward and transposed (180-degree rotated filter, channel roles
// swapped), in the A-operand order of wino43_kernel for GEMM (M = out channel m, K = in channel k):
//   [ct = m/64][kstep = k/4][wave = 2 a + bh][bi][lane = (m%16) + 16 (k%4)][cb]   with b = 3 bh + bi, cb = (m%64)/16
__device__ __forceinline__ size_t upack43_index(int m, int k, int aa, int bb, int K) {
    const int ct = m >> 6, col = m & 63, cb = col >> 4, m16 = col & 15;
    const int kstep = k >> 2, kq = k & 3;
    const int wave = 2 * aa + (bb >= 3 ? 1 : 0), bi = bb % 3;
    const int lane = m16 + 16 * kq;
    // ([wave][bi][lane][cb]: every 16-byte operand load of a wave reads 1 KB of contiguous memory -- eight full cache lines;
    //  with the lane's three b next to each other it touched all 24 lines of the 3 KB block, a third of each)
    return (((((size_t)ct * (K >> 2) + kstep) * 12 + wave) * 3 + bi) * 64 + lane) * 4 + cb;
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

// tiles across per workgroup step (Geo43): 16 where the row is a multiple of 64 pixels, else 8 for multiples of 32; 0 = not covered
int tc43(int H, int W) {
    if (H > 0 && W > 0 && (H % 4) == 0 && (W % 64) == 0) return 16;
    if (H > 0 && W > 0 && (H % 8) == 0 && (W % 32) == 0) return 8;
    return 0;
}
bool shape_ok43(int Cin, int Cout, int H, int W) {
    return Cin >= 4 * KS6 && (Cin % KS6) == 0 && (Cout % 64) == 0 && tc43(H, W) != 0 &&
           (unsigned long long)Cin * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31) &&
           (unsigned long long)Cout * (unsigned long long)H * (unsigned long long)W * 4ull < (1ull << 31);
}

template <int MODE, int TC>
int launch_wino43_tc(Wino43Args a, hipStream_t s) {
    using G = Geo43<TC>;
    a.tiles_x = a.W / G::COLS;
    a.tiles_y = a.H / G::ROWS;
    a.n_ct = a.Cout / 64;
    a.magic_x = (unsigned)((1ull << 32) / (unsigned)a.tiles_x + 1ull);      // (tiles_x == 1: unused)
    a.magic_y = (unsigned)((1ull << 32) / (unsigned)a.tiles_y + 1ull);
    // one persistent workgroup per CU: n_ct cout tiles x nslots slots, each slot walking npix / nslots pixel tiles
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const long npix = (long)a.tiles_x * a.tiles_y * a.N;
    long nslots = cus / a.n_ct;
    if (const char *e = getenv("ST3D_W43_SLOTS")) nslots = atol(e);       // lab: 0 = one workgroup per tile
    if (nslots < 1 || nslots > npix) nslots = npix;
    const long blocks = (long)a.n_ct * nslots;
    a.xpc = 1;
    if (a.n_ct <= 8 && 8 % a.n_ct == 0 && nslots % (8 / a.n_ct) == 0) a.xpc = 8 / a.n_ct;
    if (const char *e = getenv("ST3D_W43_XCD")) { if (atoi(e) == 0) a.xpc = 1; }        // A/B: 0 = slots in tile order
    constexpr size_t kSmem = (size_t)G::SMEM * sizeof(float);     // dynamic LDS above the 64 KB static limit: opt in once per instantiation
    auto go = [&](auto kernel) -> int {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem);
        if (attr != hipSuccess) { st3d::set_error("wino43: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(attr)); return ST3D_E_HIP; }
        kernel<<<(unsigned)blocks, NT6, kSmem, s>>>(a);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    };
    if (a.yp) {
        if (MODE != 0 || a.gate) { st3d::set_error("wino43: the fused pool belongs to the plain forward"); return ST3D_E_INVALID; }
        return go(wino43_kernel<0, 1, 0, TC>);
    }
    if (a.gate && a.addt) {
        if (MODE != 0) { st3d::set_error("wino43: the content-target term rides on ungated input (MODE 0) only"); return ST3D_E_INVALID; }
        return go(wino43_kernel<0, 0, 2, TC>);
    }
    if (a.gate) return go(wino43_kernel<MODE, 0, 1, TC>);
    return go(wino43_kernel<MODE, 0, 0, TC>);
}

template <int MODE>
int launch_wino43(Wino43Args a, hipStream_t s) {
    return tc43(a.H, a.W) == 16 ? launch_wino43_tc<MODE, 16>(a, s) : launch_wino43_tc<MODE, 8>(a, s);
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
