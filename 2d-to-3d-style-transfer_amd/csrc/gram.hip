// gram.hip -- Gram matrix forward/backward on the gfx950 fp32 matrix pipe.
//   forward : G_b = F_b F_b^T, F_b = feat[b] viewed (C, HW)        (style_transfer.py:31-35)
//   backward: dF_b (+)= coef * D_b F_b, D_b = G_b - S_b symmetric   (autograd of losses.py:36-39)
// One batched, split-K GEMM kernel (v_mfma_f32_32x32x2_f32, 4 waves as 2x2, k-major LDS
// tiles so both operands are lane-contiguous ds_read_b32) serves both:
//   forward  = "NT": A = F (M=C, K=HW contiguous), B = F^T given as [N][K]; K split over
//              workgroups (HW is up to 2^20 while the output is only CxC); only tiles on or
//              above the diagonal are computed; each split writes its own slab and an ordered
//              reduce kernel sums the slabs (bitwise reproducible, no float atomics).
//   backward = "NN": A = D (M=C, K=C), B = F as [K][N=HW].
// Arithmetic intensity is 32 flop/B per 128x128 tile (A/B tiles re-read through L2), at the
// fp32 ridge for C = 64 where F (the largest activation) is streamed exactly once.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KCH0 = 32;          // K chunk of the general kernels; the short-K backward of C = 64 / 128 stages 64 at once

struct GemmArgs {
    const float *A; const float *B; float *C;
    int M, N, K;
    int lda, ldb, ldc;
    size_t sA, sB, sC;      // batch strides (elements)
    size_t sSplit;          // split-K slab stride of C (elements)
    int nsplit, kper;       // K range per split
    int tiles_m, tiles_n, tri;
    float coef; int accumulate;
    int gate;               // backward only: zero the output where B (= F, the post-ReLU activation) is <= 0
};

// load 4 consecutive elements p[0..3] along a contiguous axis, zero past `remain`
__device__ __forceinline__ float4 load4(const float *p, int remain, bool aligned) {
    if (remain >= 4 && aligned) return *reinterpret_cast<const float4 *>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (remain > 0) v.x = p[0];
    if (remain > 1) v.y = p[1];
    if (remain > 2) v.z = p[2];
    if (remain > 3) v.w = p[3];
    return v;
}

// MT/NT: 32x32 MFMA tiles per wave along M/N.  BMODE 0: B is [N][K]; 1: B is [K][N].
// DIAG 1: the only tile of a Gram forward whose C fits one tile (C = 64, 128) -- the B tile IS the A tile, fetched and
// staged once (compile-time: the run-time test cost the multi-tile layers more than it saved them).
// FAST: every tile is whole and every access 16-byte aligned (the VGG shapes; checked by the launchers): no bounds code in
// the loop, which then is one basic block per phase and gets an explicit interleaved schedule (one MFMA, one LDS read; the
// next chunk's eight global loads under the first MFMAs) instead of whatever falls out of ~50 predicated branches.
// (the body takes its block coordinates and LDS as arguments: gemm_kernel passes blockIdx, the multi-layer Gram forward --
// gram_multi_kernel -- a work item decoded from its table)
template <int MT, int NT, int BMODE, int DIAG = 0, int KCH = 32, bool FAST = false>
struct GemmSmem {
    static constexpr int TM = 2 * MT * 32, TN = 2 * NT * 32;
    static constexpr int LA = TM + 1, LB = (BMODE == 0) ? TN + 1 : TN;
    static constexpr int FLOATS = KCH * LA + KCH * LB;
};

template <int MT, int NT, int BMODE, int DIAG = 0, int KCH = 32, bool FAST = false>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, const int bx, const int by, const int bz, float *smem) {
    constexpr int TM = 2 * MT * 32, TN = 2 * NT * 32;
    constexpr int LA = TM + 1;
    constexpr int LB = (BMODE == 0) ? TN + 1 : TN;
    constexpr int A4 = TM * KCH / 4 / 256;      // float4 loads per thread for the A tile
    constexpr int B4 = TN * KCH / 4 / 256;
    float *As = smem, *Bs = smem + KCH * LA;    // (KCH * LA is a multiple of 4 floats for every instantiation with vector LDS stores: BMODE 1 uses LB = TN)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    int ti, tj;
    if (g.tri) {   // upper-triangular tile pairs (ti <= tj)
        int t = bx; ti = 0;
        while (t >= g.tiles_n - ti) { t -= g.tiles_n - ti; ++ti; }
        tj = ti + t;
    } else {
        ti = bx / g.tiles_n; tj = bx % g.tiles_n;
    }
    const int split = by, b = bz;
    const int m0 = ti * TM, n0 = tj * TN;
    // a diagonal tile of the multi-tile Gram forward: the lower-left wave would compute the mirror image of the upper-right
    // one -- it sits the MFMAs out and the reduce mirrors at wave-tile granularity (FAST launches only)
    const bool mirror_wave = FAST && BMODE == 0 && !DIAG && g.tri && ti == tj && wm > wn;
    static_assert(!DIAG || (BMODE == 0 && MT == NT), "DIAG is the single-tile Gram forward");
    constexpr bool diag = DIAG != 0;
    const int kbeg = split * g.kper, kend = min(g.K, kbeg + g.kper);

    const float *Ab = g.A + b * g.sA;
    const float *Bb = g.B + b * g.sB;
    const bool a_al = ((g.lda & 3) == 0) && ((((uintptr_t)Ab) & 15) == 0);
    const bool b_al = ((g.ldb & 3) == 0) && ((((uintptr_t)Bb) & 15) == 0);

    // C = coef * A B (+ C): the accumulators START from the destination tile when accumulating (its loads are in flight
    // under the first operand tiles instead of a dependent load -> add -> store chain behind the last MFMA) and coef is
    // folded into the A tile on its way into LDS, so the epilogue is a plain store.
    float *Cb = g.C + b * g.sC + split * g.sSplit;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
                if (g.accumulate) {
                    const int gm = m0 + wm * (MT * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int gn = n0 + wn * (NT * 32) + q * 32 + l31;
                    if (FAST || (gm < g.M && gn < g.N)) v = Cb[(size_t)gm * g.ldc + gn];
                }
                acc[m][q][r] = v;
            }

    unsigned gmask[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q) gmask[m][q] = 0xffffu;
    float4 av[A4], bv[B4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {       // A tile: TM rows x 32 k, 8 float4 per row
            const int e = tid + i * 256, row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
            const int gm = m0 + row, gk = k0 + kq;
            if (FAST) av[i] = *reinterpret_cast<const float4 *>(Ab + (size_t)gm * g.lda + gk);
            else av[i] = (gm < g.M) ? load4(Ab + (size_t)gm * g.lda + gk, kend - gk, a_al && ((gk & 3) == 0))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (diag) return;
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int e = tid + i * 256;
            if (BMODE == 0) {                // [N][K]: TN rows x 32 k
                const int row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
                const int gn = n0 + row, gk = k0 + kq;
                if (FAST) bv[i] = *reinterpret_cast<const float4 *>(Bb + (size_t)gn * g.ldb + gk);
                else bv[i] = (gn < g.N) ? load4(Bb + (size_t)gn * g.ldb + gk, kend - gk, b_al && ((gk & 3) == 0))
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {                         // [K][N]: 32 k rows x TN cols
                const int kr = e / (TN / 4), nq = (e - kr * (TN / 4)) * 4;
                const int gk = k0 + kr, gn = n0 + nq;
                if (FAST) bv[i] = *reinterpret_cast<const float4 *>(Bb + (size_t)gk * g.ldb + gn);
                else bv[i] = (gk < kend) ? load4(Bb + (size_t)gk * g.ldb + gn, g.N - gn, b_al && ((gn & 3) == 0))
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int e = tid + i * 256, row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
            const float cf = BMODE == 1 ? g.coef : 1.f;        // the forward has no coefficient: no multiply
            As[(kq + 0) * LA + row] = cf * av[i].x; As[(kq + 1) * LA + row] = cf * av[i].y;
            As[(kq + 2) * LA + row] = cf * av[i].z; As[(kq + 3) * LA + row] = cf * av[i].w;
        }
        if (diag) return;
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int e = tid + i * 256;
            if (BMODE == 0) {
                const int row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
                Bs[(kq + 0) * LB + row] = bv[i].x; Bs[(kq + 1) * LB + row] = bv[i].y;
                Bs[(kq + 2) * LB + row] = bv[i].z; Bs[(kq + 3) * LB + row] = bv[i].w;
            } else {
                const int kr = e / (TN / 4), nq = (e - kr * (TN / 4)) * 4;
                *reinterpret_cast<float4 *>(&Bs[kr * LB + nq]) = bv[i];
            }
        }
    };

    if (kbeg < kend) load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KCH) {
        __syncthreads();                     // previous chunk's reads done
        store_tiles();
        __syncthreads();
        if (k0 + KCH < kend) load_tiles(k0 + KCH);   // in flight during the MFMAs below
        if (BMODE == 1 && g.gate) {
            // the ReLU gate of the OUTPUT rows (channels m) is the sign of F[m][n]: those rows pass through LDS as the
            // B chunk whose k range holds m (K = M = C here), so the wave picks its 16 x NT gate bits up on the way
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int kb = m0 + wm * (MT * 32) + m * 32;
                if (kb >= k0 && kb < k0 + KCH) {
                    const float *pg = Bs + (kb - k0 + 4 * lhi) * LB + wn * (NT * 32) + l31;
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        unsigned mk = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) mk |= (pg[((r & 3) + 8 * (r >> 2)) * LB + q * 32] > 0.f ? 1u : 0u) << r;
                        gmask[m][q] = mk;
                    }
                }
            }
        }
        const float *pa = As + lhi * LA + wm * (MT * 32) + l31;
        const float *pb = (diag ? As : Bs) + lhi * LB + wn * (NT * 32) + l31;      // (LA == LB when TM == TN)
        if (mirror_wave) continue;           // (still stages its share of the tiles and meets the barriers)
#pragma unroll
        for (int kk = 0; kk < KCH / 2; ++kk) {
            float a[MT], bb[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = pa[(kk * 2) * LA + m * 32];
#pragma unroll
            for (int q = 0; q < NT; ++q) bb[q] = pb[(kk * 2) * LB + q * 32];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[q], acc[m][q], 0, 0, 0);
        }
        if (FAST) {
            // schedule of the region since the barrier: one MFMA, then one LDS read (operands of a later k-step) and, while
            // they last, one of the next chunk's global loads
#pragma unroll
            for (int i = 0; i < (KCH / 2) * MT * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT + MT * NT - 1) / (MT * NT), 0);      // LDS reads per MFMA, rounded up
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (mirror_wave) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * (MT * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (!FAST && gm >= g.M) continue;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int gn = n0 + wn * (NT * 32) + q * 32 + l31;
                if (FAST || gn < g.N) {
                    Cb[(size_t)gm * g.ldc + gn] = (BMODE == 1 && !((gmask[m][q] >> r) & 1u)) ? 0.f : acc[m][q][r];
                }
            }
        }
}

template <int MT, int NT, int BMODE, int DIAG = 0, int KCH = 32, bool FAST = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[(GemmSmem<MT, NT, BMODE, DIAG, KCH, FAST>::FLOATS + 3) & ~3];
    gemm_body<MT, NT, BMODE, DIAG, KCH, FAST>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Gram backward of the loss plan (st3d_gram_bwd_gated: D = G - S is SYMMETRIC, whole 128 x 64 tiles, aligned).  Same tile
// shape and arithmetic as gemm_kernel<2, 1, 1> -- 128 channels x 64 pixels per workgroup, each wave 64 x 32, accumulators
// started from the destination tile, coef folded into A, gate bits from the F chunk in LDS -- but written like the Winograd
// loop: every global access goes through a buffer descriptor with a 32-bit lane offset and a scalar chunk / row offset (no
// 64-bit address arithmetic, no bounds code), the A tile is read ALONG m from row k of D (= column k, by symmetry), so it
// lands k-major in LDS with 16-byte stores instead of 16 transposing 4-byte ones, and the chunk loop is one basic block
// with an explicit MFMA / LDS-read interleave and the next chunk's loads first.
__global__ __launch_bounds__(256, 4) void gram_bwd_sym_kernel(const GemmArgs g) {
    constexpr int TM = 128, TN = 64, KCH = 32, LA = TM + 4, LB = TN;
    constexpr unsigned kOob = 0x80000000u;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float As[KCH * LA];
    __shared__ __attribute__((aligned(16))) float Bs[KCH * LB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ti = blockIdx.x / g.tiles_n, tj = blockIdx.x % g.tiles_n, b = blockIdx.z;
    const int m0 = ti * TM, n0 = tj * TN;
    const unsigned rowD = (unsigned)g.lda * 4u, rowF = (unsigned)g.ldb * 4u;      // bytes per row of D / of F and the result
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A + b * g.sA), 0,
                                                                       (unsigned)g.K * rowD, 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B + b * g.sB), 0,
                                                                       (unsigned)g.K * rowF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(g.C + b * g.sC, 0, (unsigned)g.M * rowF, 0x00020000);
    // staging items: A 32 k-rows x 32 float4 (4 per thread), B 32 k-rows x 16 float4 (2 per thread)
    unsigned voA[4], voB[2];
    int loA[4], loB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256, kr = e >> 5, mq = (e & 31) * 4;
        voA[i] = (unsigned)kr * rowD + (unsigned)(m0 + mq) * 4u;
        loA[i] = kr * LA + mq;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256, kr = e >> 4, nq = (e & 15) * 4;
        voB[i] = (unsigned)kr * rowF + (unsigned)(n0 + nq) * 4u;
        loB[i] = kr * LB + nq;
    }
    // this lane's rows of the result: m0 + 64 wm + 32 m + (r & 3) + 8 (r >> 2) + 4 lhi, column n0 + 32 wn + lane31
    const unsigned voC = (unsigned)(m0 + wm * 64 + 4 * lhi) * rowF + (unsigned)(n0 + wn * 32 + l31) * 4u;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[m][r] = g.accumulate ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                           rC, voC, (unsigned)(m * 32 + (r & 3) + 8 * (r >> 2)) * rowF, 0))
                                     : 0.f;
    unsigned gmask[2] = {0xffffu, 0xffffu};
    f32x4 av[4], bv[2];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rD, voA[i], (unsigned)k0 * rowD, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) bv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rF, voB[i], (unsigned)k0 * rowF, 0));
    };
    gload(0);
    const float *pa = As + lhi * LA + wm * 64 + l31;
    const float *pb = Bs + lhi * LB + wn * 32 + l31;
    for (int k0 = 0; k0 < g.K; k0 += KCH) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&As[loA[i]]) = g.coef * av[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(&Bs[loB[i]]) = bv[i];
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        gload(min(k0 + KCH, g.K - KCH));            // (the last chunk re-requests itself: no branch in the loop)
        if (g.gate) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m0 + wm * 64 + m * 32 == k0) {
                    unsigned mk = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mk |= (Bs[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LB + wn * 32 + l31] > 0.f ? 1u : 0u) << r;
                    gmask[m] = mk;
                }
        }
#pragma unroll
        for (int kk = 0; kk < KCH / 2; ++kk) {
            const float a0 = pa[(kk * 2) * LA], a1 = pa[(kk * 2) * LA + 32], bb = pb[(kk * 2) * LB];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb, acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float o = ((gmask[m] >> r) & 1u) ? acc[m][r] : 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rC, voC,
                                                  (unsigned)(m * 32 + (r & 3) + 8 * (r >> 2)) * rowF, 0);
        }
}

// Gram forward of the single-tile layers (C = 64: NB = 2, C = 128: NB = 4 blocks of 32 channels), whole aligned shapes only.
// The 4 waves of gemm_kernel<.., DIAG> tile the full C x C output, so a quarter (C = 64) to three eighths (C = 128) of
// their MFMAs compute the mirror image of another wave's block.  Here only the NB (NB + 1) / 2 blocks on or above the
// diagonal exist and the waves split the K chunk instead: C = 64 -- every wave owns all 3 blocks over a quarter of the
// chunk's k-steps; C = 128 -- two block groups of 5 x two k-groups.  Each k-group writes its own slab (the ordered
// reduce sums KG x nsplit of them and mirrors at block granularity), so nothing is exchanged between waves.  One LDS
// value serves as row operand and as column operand (the tile is its own transpose partner).
template <int NB>
__device__ __forceinline__ void gram_diag_body(const GemmArgs &g, const int by, const int bz, float *As) {
    constexpr int TM = 32 * NB, LA = TM + 1, KCH = 32;
    constexpr int KG = NB == 2 ? 4 : 2;                 // k-groups
    constexpr int NBLK = NB == 2 ? 3 : 5;               // blocks per wave
    constexpr int A4 = TM * KCH / 4 / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kg = NB == 2 ? wave : (wave >> 1), bg = NB == 2 ? 0 : (wave & 1);
    const int split = by, b = bz;
    const int kbeg = split * g.kper, kend = min(g.K, kbeg + g.kper);
    const float *Ab = g.A + b * g.sA;
    float *Cb = g.C + b * g.sC + (size_t)(split * KG + kg) * g.sSplit;
    // block list (row block, column block) of this wave's block group
    constexpr int BI[2][5] = {{0, 0, 0, 0, 1}, {1, 1, 2, 2, 3}};
    constexpr int BJ[2][5] = {{0, 1, 2, 3, 1}, {2, 3, 2, 3, 3}};
    constexpr int CI[3] = {0, 0, 1}, CJ[3] = {0, 1, 1};

    f32x16 acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 av[A4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int e = tid + i * 256, row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
            av[i] = *reinterpret_cast<const float4 *>(Ab + (size_t)row * g.lda + k0 + kq);
        }
    };
    if (kbeg < kend) load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KCH) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int e = tid + i * 256, row = e / (KCH / 4), kq = (e % (KCH / 4)) * 4;
            As[(kq + 0) * LA + row] = av[i].x; As[(kq + 1) * LA + row] = av[i].y;
            As[(kq + 2) * LA + row] = av[i].z; As[(kq + 3) * LA + row] = av[i].w;
        }
        __syncthreads();
        if (k0 + KCH < kend) load_tile(k0 + KCH);
        const float *pa = As + lhi * LA + l31;
        // (the block group is wave-uniform: two straight-line copies of the loop with compile-time block indices)
        auto mfmas = [&](auto BGc) __attribute__((always_inline)) {
            constexpr int BG = decltype(BGc)::value;
#pragma unroll
            for (int kk = 0; kk < KCH / 2 / KG; ++kk) {
                const int krow = 2 * (kg * (KCH / 2 / KG) + kk);
                float v[NB];
#pragma unroll
                for (int x = 0; x < NB; ++x) v[x] = pa[krow * LA + x * 32];
#pragma unroll
                for (int i = 0; i < NBLK; ++i) {
                    const int bi = NB == 2 ? CI[i] : BI[BG][i];
                    const int bj = NB == 2 ? CJ[i] : BJ[BG][i];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[bi], v[bj], acc[i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < (KCH / 2 / KG) * NBLK; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (bg == 0) mfmas(std::integral_constant<int, 0>{});
        else mfmas(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int bi = NB == 2 ? CI[i] : (bg == 0 ? BI[0][i] : BI[1][i]);
        const int bj = NB == 2 ? CJ[i] : (bg == 0 ? BJ[0][i] : BJ[1][i]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Cb[(size_t)(bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * g.ldc + bj * 32 + l31] = acc[i][r];
    }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void gram_diag_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[32 * (32 * NB + 1)];
    gram_diag_body<NB>(g, blockIdx.y, blockIdx.z, As);
}

// ---- all style layers of a step in ONE launch (round 3).  Five launches of 500-1300 workgroups each ran one after the
// other, every one with its own ramp and tail, the HBM-bound relu1_1 layer (64 row streams of 1 MB per image) with the
// matrix pipe idle and the deep layers with HBM idle.  Here one grid holds the workgroups of every layer: a block finds
// its (layer, tile, split, image) in a small table passed as the kernel argument and runs that layer's body -- the same
// code, the same per-split slabs, so the sums are bitwise what the separate launches produce.  The relu1_1 blocks are
// dealt between the others (every `stride`-th block) so that streams and MFMA work overlap in time.
constexpr int kMaxGramItems = 8;
struct GramMulti {
    GemmArgs g[kMaxGramItems];
    int kind[kMaxGramItems];            // 0: gram_diag<2>, 1: gram_diag<4>, 2: gemm<2,2> multi-tile FAST, 3: generic (not FAST)
    int gx[kMaxGramItems], gy[kMaxGramItems];
    int first[kMaxGramItems + 1];       // block ranges of items 1.. in the "rest" numbering; item 0 is the dealt-in one
    int n_items, n0, stride;            // n0 blocks of item 0, one every `stride` blocks (stride 0: item 0 is part of the rest)
};

__global__ __launch_bounds__(256, 2) void gram_multi_kernel(const GramMulti m) {
    __shared__ __attribute__((aligned(16))) float smem[(GemmSmem<2, 2, 0, 0, 32, true>::FLOATS + 3) & ~3];
    int b = blockIdx.x, it;
    if (m.stride > 0) {
        const int q = b / m.stride, dealt = b - q * m.stride == 0 && q < m.n0;
        if (dealt) { it = 0; b = q; }
        else {
            const int before = min(q + 1, m.n0);     // item-0 blocks among the indices below this one
            b -= before; it = -1;
        }
    } else it = -1;
    if (it < 0) {
        it = m.stride > 0 ? 1 : 0;
        while (it + 1 < m.n_items && b >= m.first[it + 1]) ++it;
        b -= m.first[it];
    }
    const GemmArgs &g = m.g[it];
    const int gx = m.gx[it], gy = m.gy[it];
    const int bx = b % gx, by = (b / gx) % gy, bz = b / (gx * gy);
    switch (m.kind[it]) {
        case 0: gram_diag_body<2>(g, by, bz, smem); break;
        case 1: gram_diag_body<4>(g, by, bz, smem); break;
        case 2: gemm_body<2, 2, 0, 0, 32, true>(g, bx, by, bz, smem); break;
        default: break;
    }
}

struct ReduceItem { const float *slab; float *gram; int nsplit, C, TM, blocks_per_image; size_t sSplit, sB; };
struct ReduceMulti { ReduceItem r[kMaxGramItems]; int first[kMaxGramItems + 1]; int n_items; };

// G[b][i][j] = sum over the split slabs of the element, in a FIXED tree (4 interleaved partial sums per element, each
// over its slabs in ascending order, then ((p0+p1)+(p2+p3))) -- bitwise reproducible, and four times the loads in
// flight of a single running sum.  Only tiles on/above the diagonal are read (coalesced); each such element is also
// written to its mirror position, so the Gram is exactly symmetric.
// VW consecutive elements of a row per thread (VW = 4: 16-byte loads of the slabs, a quarter of the workgroups; C and TM
// multiples of 4) -- every element keeps its own tree, so the result does not depend on VW.
template <int VW>
__device__ __forceinline__ void gram_reduce_body(const float *__restrict__ slab, int nsplit, int C, int TM, size_t sSplit,
                                                 size_t sB, float *__restrict__ gram, const int bx, const int by,
                                                 float (*part)[64 * VW]) {
    typedef float vec __attribute__((ext_vector_type(VW)));
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t i = ((size_t)bx * 64 + e) * VW;
    const size_t CC = (size_t)C * C;
    const int b = by;
    float s[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) s[j] = 0.f;
    const int r = (i < CC) ? (int)(i / C) : 0, c = (i < CC) ? (int)(i % C) : 0;
    const bool lower = r / TM > c / TM;         // tile below the diagonal: written by its mirror image's threads
    if (i < CC && !lower) {
        const float *p = slab + b * sB + (size_t)r * C + c;
        auto ld = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const vec *>(p + k * sSplit); };
        int k = q;
        for (; k + 12 < nsplit; k += 16) {
            const vec a0 = ld(k), a1 = ld(k + 4), a2 = ld(k + 8), a3 = ld(k + 12);
#pragma unroll
            for (int j = 0; j < VW; ++j) { s[j] += a0[j]; s[j] += a1[j]; s[j] += a2[j]; s[j] += a3[j]; }
        }
        for (; k < nsplit; k += 4) {
            const vec a0 = ld(k);
#pragma unroll
            for (int j = 0; j < VW; ++j) s[j] += a0[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) part[q][e * VW + j] = s[j];
    __syncthreads();
    if (q == 0 && i < CC && !lower) {
        vec v;
#pragma unroll
        for (int j = 0; j < VW; ++j) v[j] = (part[0][e * VW + j] + part[1][e * VW + j]) + (part[2][e * VW + j] + part[3][e * VW + j]);
        *reinterpret_cast<vec *>(gram + b * CC + i) = v;
        if (r / TM < c / TM) {                                               // the mirrored (never computed) tile
#pragma unroll
            for (int j = 0; j < VW; ++j) gram[b * CC + (size_t)(c + j) * C + r] = v[j];
        }
    }
}

template <int VW>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float *__restrict__ slab, int nsplit, int C, int TM,
                                                          size_t sSplit, size_t sB, float *__restrict__ gram) {
    __shared__ float part[4][64 * VW];
    gram_reduce_body<VW>(slab, nsplit, C, TM, sSplit, sB, gram, blockIdx.x, blockIdx.y, part);
}

// the slab reductions of all layers in one launch (same fixed tree per element as gram_reduce_kernel)
template <int VW>
__global__ __launch_bounds__(256) void gram_reduce_multi_kernel(const ReduceMulti m) {
    __shared__ float part[4][64 * VW];
    int b = blockIdx.x, it = 0;
    while (it + 1 < m.n_items && b >= m.first[it + 1]) ++it;
    b -= m.first[it];
    const ReduceItem &r = m.r[it];
    gram_reduce_body<VW>(r.slab, r.nsplit, r.C, r.TM, r.sSplit, r.sB, r.gram, b % r.blocks_per_image, b / r.blocks_per_image, part);
}

// 16-byte slab reads need C (row length, slab strides) and the tile height in multiples of 4
inline int gram_reduce_vw(int C, int TM, size_t sSplit, size_t sB) { return (C % 4 == 0 && TM % 4 == 0 && sSplit % 4 == 0 && sB % 4 == 0) ? 4 : 1; }

// K split: enough workgroups to fill the chip a few times over (256 CUs x 2 resident workgroups x 1..4), but
// at least 8 K-chunks of work per workgroup and at most 256 slabs.
// scale > 1 (the multi-layer launch): a layer no longer has to fill the chip on its own -- the other layers' blocks run
// beside it -- so it takes 1/scale of the workgroups, each over scale x the pixels: 1/scale of the slab traffic
// (a slab is C x C floats per split: at scale 1 the five layers write and re-read 0.4 GB of slabs per step, against 1.0 GB
// of activations read).
int gram_split(int B, int C, int HW, int *kper, int scale = 1) {
    const int nt = (C % 128 == 0) ? C / 128 : (C + 63) / 64;
    const int pairs = nt * (nt + 1) / 2;
    // workgroups aimed for, measured per style layer of config 2 (tools/gram_sweep.py): the deep layers (few tiles,
    // short K) want more, thinner splits; C = 128 fewer
    // (round 2 re-measured with the split count itself, ST3D_GRAM_NSPLIT: 256 channels at 128^2 32 splits 145 vs 162 us at
    // 64; 512 at 64^2 16 splits 144 vs 149 at 26; 512 at 32^2 8 splits 69 vs 76 at 4)
    const int target = (C >= 512 ? 1280 : (C == 256 ? 768 : (C == 128 ? 512 : 1024))) / scale;
    int ns = (target + pairs * B - 1) / (pairs * B);
    if (const char *ev = getenv("ST3D_GRAM_TARGET_WGS")) {       // tuning knob (tools/gram_sweep.py): workgroups aimed for
        const int t = atoi(ev);
        if (t > 0) ns = (t + pairs * B - 1) / (pairs * B);
    }
    const int ns_bytes = (HW + 2048 * scale - 1) / (2048 * scale);      // never more than 2048 (x scale) pixels per workgroup
    if (ns < ns_bytes) ns = ns_bytes;
    const int ns_max = (HW + 4 * KCH0 - 1) / (4 * KCH0);   // at least 4 K-chunks (128 pixels) of work per workgroup
    if (ns > ns_max) ns = ns_max;
    if (ns > 256) ns = 256;
    if (const char *ev = getenv("ST3D_GRAM_NSPLIT")) {           // tuning knob (tools/gram_sweep.py): the split count itself
        const int t = atoi(ev);
        if (t > 0) ns = t;
    }
    if (ns < 1) ns = 1;
    int kp = (HW + ns - 1) / ns;
    kp = (kp + KCH0 - 1) / KCH0 * KCH0;
    ns = (HW + kp - 1) / kp;
    *kper = kp;
    return ns;
}

}  // namespace

// the single-tile layers may run gram_diag_kernel, which writes KG slabs per split (4 for C = 64, 2 for C = 128)
static int gram_kgroups(int C) { return C == 64 ? 4 : (C == 128 ? 2 : 1); }

extern "C" size_t st3d_gram_workspace_bytes(int B, int C, int HW) {
    int kper;
    const int ns = gram_split(B, C, HW, &kper);
    return (size_t)B * ns * gram_kgroups(C) * C * C * sizeof(float);
}

// Launch description of one Gram forward: which kernel (kind as in GramMulti; 3 = one of the generic instantiations), its
// arguments and grid, and the reduce that follows (slabs to sum, mirror granularity).
struct GramFwdPlan { GemmArgs g; int kind, gx, gy, TM; bool fast; int red_nsplit, red_tm; };

static GramFwdPlan gram_fwd_plan(const float *feat, int B, int C, int HW, void *workspace, int scale = 1) {
    GramFwdPlan q;
    GemmArgs &g = q.g;
    memset(&g, 0, sizeof(g));
    g.nsplit = gram_split(B, C, HW, &g.kper, scale);
    g.A = feat; g.B = feat; g.C = reinterpret_cast<float *>(workspace);
    g.M = C; g.N = C; g.K = HW; g.lda = HW; g.ldb = HW; g.ldc = C;
    g.sA = g.sB = (size_t)C * HW;
    g.sSplit = (size_t)C * C; g.sC = g.sSplit * g.nsplit;
    g.tri = 1; g.coef = 1.f; g.accumulate = 0;
    const int TM = q.TM = (C % 128 == 0) ? 128 : 64;
    g.tiles_m = g.tiles_n = st3d::cdiv(C, TM);
    q.gx = g.tiles_n * (g.tiles_n + 1) / 2; q.gy = g.nsplit;
    // whole tiles, whole 32-pixel chunks in every split, 16-byte aligned rows: the branch-free instantiation
    static const bool allow_fast = [] { const char *e = getenv("ST3D_GRAM_FAST"); return !(e && e[0] == '0'); }();
    q.fast = allow_fast && C % TM == 0 && HW % 32 == 0 && g.kper % 32 == 0 && (((uintptr_t)feat) & 15) == 0;
    static const bool tri = [] { const char *e = getenv("ST3D_GRAM_DIAG_TRI"); return !(e && e[0] == '0'); }();
    q.kind = 3; q.red_nsplit = g.nsplit;
    if (q.fast && tri && (C == 64 || C == 128)) {     // upper blocks only, waves split the K chunk (gram_diag_kernel)
        const int KG = gram_kgroups(C);
        g.sC = g.sSplit * g.nsplit * KG;
        q.kind = C == 64 ? 0 : 1; q.gx = 1; q.red_nsplit = g.nsplit * KG; q.red_tm = 32;
        return q;
    }
    if (q.fast && g.tiles_n > 1 && TM == 128) q.kind = 2;
    // mirror granularity: the multi-tile FAST launch leaves the lower-left 64 x 64 of its diagonal tiles unwritten
    q.red_tm = q.kind == 2 ? 64 : TM;
    return q;
}

extern "C" int st3d_gram_fwd(const float *feat, int B, int C, int HW, void *workspace, size_t workspace_bytes, float *gram,
                             st3d_stream_t stream) {
    ST3D_CHECK_ARG(feat && workspace && gram);
    ST3D_CHECK_ARG(B > 0 && C > 0 && HW > 0);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_gram_workspace_bytes(B, C, HW));
    hipStream_t s = st3d::as_stream(stream);
    const GramFwdPlan q = gram_fwd_plan(feat, B, C, HW, workspace);
    const GemmArgs &g = q.g;
    const int TM = q.TM;
    const bool fast = q.fast;
    dim3 grid(q.gx, q.gy, B);
    if (q.kind <= 1) {
        if (C == 64) gram_diag_kernel<2><<<dim3(1, g.nsplit, B), 256, 0, s>>>(g);
        else gram_diag_kernel<4><<<dim3(1, g.nsplit, B), 256, 0, s>>>(g);
        ST3D_LAUNCH_CHECK();
        if (gram_reduce_vw(C, q.red_tm, g.sSplit, g.sC) == 4)
            gram_reduce_kernel<4><<<dim3(st3d::cdiv((long)C * C, 256), B), 256, 0, s>>>(g.C, q.red_nsplit, C, q.red_tm, g.sSplit, g.sC, gram);
        else
            gram_reduce_kernel<1><<<dim3(st3d::cdiv((long)C * C, 64), B), 256, 0, s>>>(g.C, q.red_nsplit, C, q.red_tm, g.sSplit, g.sC, gram);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
    if (g.tiles_n == 1) {           // one (diagonal) tile: A and B tiles coincide
        if (TM == 128) { if (fast) gemm_kernel<2, 2, 0, 1, 32, true><<<grid, 256, 0, s>>>(g); else gemm_kernel<2, 2, 0, 1><<<grid, 256, 0, s>>>(g); }
        else { if (fast) gemm_kernel<1, 1, 0, 1, 32, true><<<grid, 256, 0, s>>>(g); else gemm_kernel<1, 1, 0, 1><<<grid, 256, 0, s>>>(g); }
    } else if (TM == 128) { if (fast) gemm_kernel<2, 2, 0, 0, 32, true><<<grid, 256, 0, s>>>(g); else gemm_kernel<2, 2, 0><<<grid, 256, 0, s>>>(g); }
    else gemm_kernel<1, 1, 0><<<grid, 256, 0, s>>>(g);
    ST3D_LAUNCH_CHECK();
    if (gram_reduce_vw(C, q.red_tm, g.sSplit, g.sC) == 4)
            gram_reduce_kernel<4><<<dim3(st3d::cdiv((long)C * C, 256), B), 256, 0, s>>>(g.C, q.red_nsplit, C, q.red_tm, g.sSplit, g.sC, gram);
        else
            gram_reduce_kernel<1><<<dim3(st3d::cdiv((long)C * C, 64), B), 256, 0, s>>>(g.C, q.red_nsplit, C, q.red_tm, g.sSplit, g.sC, gram);
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}

// Every style layer's Gram in one launch pair (gram_multi_kernel + gram_reduce_multi_kernel).  items: host array; every
// item gets its own slab region of `workspace` (st3d_gram_multi_workspace_bytes = the sum), because the layers run
// concurrently.  Results are bitwise those of st3d_gram_fwd per item (same bodies, same splits, same reduce tree).  Items
// the fused kernel has no body for (shapes outside the whole-tile fast path) run through st3d_gram_fwd one by one.
extern "C" size_t st3d_gram_multi_workspace_bytes(const st3d_gram_item *items, int count) {
    size_t tot = 0;
    for (int i = 0; items && i < count; ++i)
        tot += (st3d_gram_workspace_bytes(items[i].B, items[i].C, items[i].HW) + 255) & ~(size_t)255;
    return tot;
}

extern "C" int st3d_gram_fwd_multi(const st3d_gram_item *items, int count, void *workspace, size_t workspace_bytes,
                                   st3d_stream_t stream) {
    ST3D_CHECK_ARG(items && count > 0 && count <= kMaxGramItems && workspace);
    ST3D_CHECK_ARG(workspace_bytes >= st3d_gram_multi_workspace_bytes(items, count) && ((uintptr_t)workspace & 255) == 0);
    hipStream_t s = st3d::as_stream(stream);
    static const bool fuse = [] { const char *e = getenv("ST3D_GRAM_MULTI"); return !(e && e[0] == '0'); }();
    // relu1_1-like items (kind 0: one row stream per channel, HBM-bound) are dealt between the blocks of the others
    static const int deal = [] { const char *e = getenv("ST3D_GRAM_MULTI_DEAL"); return e ? atoi(e) : 0; }();
    // (read per call, not cached: tests compare scale 1 with st3d_gram_fwd bit for bit)
    const int scale = [] { const char *e = getenv("ST3D_GRAM_MULTI_SCALE"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 16 ? v : 2; }();
    GramMulti gm;
    ReduceMulti rm;
    memset(&gm, 0, sizeof(gm));
    memset(&rm, 0, sizeof(rm));
    char *ws = static_cast<char *>(workspace);
    int order[kMaxGramItems], nf = 0;
    GramFwdPlan plans[kMaxGramItems];
    size_t woff[kMaxGramItems];
    size_t off = 0;
    for (int i = 0; i < count; ++i) {
        const st3d_gram_item &it = items[i];
        ST3D_CHECK_ARG(it.feat && it.gram && it.B > 0 && it.C > 0 && it.HW > 0);
        woff[i] = off;
        int sc = scale;
        if (const char *e = getenv("ST3D_GRAM_MULTI_SCALES")) {        // tuning knob (tools/gram_multi_sweep.py): "c64,c128,c256,c512"
            int v[4] = {scale, scale, scale, scale};
            sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
            sc = it.C <= 64 ? v[0] : (it.C <= 128 ? v[1] : (it.C <= 256 ? v[2] : v[3]));
            if (sc < 1 || sc > 16) sc = scale;
        }
        plans[i] = gram_fwd_plan(it.feat, it.B, it.C, it.HW, ws + off, fuse ? sc : 1);
        if (plans[i].kind > 2) plans[i] = gram_fwd_plan(it.feat, it.B, it.C, it.HW, ws + off, 1);      // (runs through st3d_gram_fwd below)
        off += (st3d_gram_workspace_bytes(it.B, it.C, it.HW) + 255) & ~(size_t)255;
    }
    // fused items: the dealt-in one (first kind-0 item) goes to slot 0, the rest by descending work per block
    int dealt = -1;
    for (int i = 0; i < count && deal; ++i)
        if (fuse && plans[i].kind == 0 && dealt < 0) dealt = i;
    if (dealt >= 0) order[nf++] = dealt;
    for (int i = 0; i < count; ++i)
        if (fuse && plans[i].kind <= 2 && i != dealt) order[nf++] = i;
    const int rest0 = dealt >= 0 ? 1 : 0;
    for (int a = rest0; a < nf; ++a)          // (insertion sort, <= 8 items) blocks with the longest K range first
        for (int b = a + 1; b < nf; ++b) {
            auto work = [&](int i) { return (long)plans[i].g.kper * (plans[i].kind == 0 ? 3 : plans[i].kind == 1 ? 10 : 16); };
            if (work(order[b]) > work(order[a])) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
        }
    if (nf > 0) {
        int restblocks = 0;
        for (int k = 0; k < nf; ++k) {
            const GramFwdPlan &q = plans[order[k]];
            gm.g[k] = q.g; gm.kind[k] = q.kind; gm.gx[k] = q.gx; gm.gy[k] = q.gy;
            const int blocks = q.gx * q.gy * items[order[k]].B;
            if (k == 0 && dealt >= 0) { gm.n0 = blocks; gm.first[0] = 0; continue; }
            gm.first[k] = restblocks;
            restblocks += blocks;
        }
        gm.first[nf] = restblocks;
        gm.n_items = nf;
        const int total = gm.n0 + restblocks;
        gm.stride = (dealt >= 0 && gm.n0 > 0) ? (total / gm.n0 > 0 ? total / gm.n0 : 1) : 0;
        if (restblocks == 0) gm.stride = gm.n0 > 0 ? 1 : 0;
        gram_multi_kernel<<<total, 256, 0, s>>>(gm);
        ST3D_LAUNCH_CHECK();
        int rblocks = 0, vw = 4;
        for (int k = 0; k < nf; ++k) {
            const GramFwdPlan &q = plans[order[k]];
            if (gram_reduce_vw(items[order[k]].C, q.red_tm, q.g.sSplit, q.g.sC) != 4) vw = 1;
        }
        for (int k = 0; k < nf; ++k) {
            const int i = order[k];
            const GramFwdPlan &q = plans[i];
            const int bpi = (int)st3d::cdiv((long)items[i].C * items[i].C, 64 * vw);
            rm.r[k] = ReduceItem{q.g.C, items[i].gram, q.red_nsplit, items[i].C, q.red_tm, bpi, q.g.sSplit, q.g.sC};
            rm.first[k] = rblocks;
            rblocks += bpi * items[i].B;
        }
        rm.first[nf] = rblocks; rm.n_items = nf;
        if (vw == 4) gram_reduce_multi_kernel<4><<<rblocks, 256, 0, s>>>(rm);
        else gram_reduce_multi_kernel<1><<<rblocks, 256, 0, s>>>(rm);
        ST3D_LAUNCH_CHECK();
    }
    for (int i = 0; i < count; ++i)
        if (!(fuse && plans[i].kind <= 2)) {
            const st3d_gram_item &it = items[i];
            ST3D_TRY(st3d_gram_fwd(it.feat, it.B, it.C, it.HW, ws + woff[i], st3d_gram_workspace_bytes(it.B, it.C, it.HW), it.gram, stream));
        }
    return ST3D_OK;
}

static int gram_bwd_launch(const float *D, const float *feat, int B, int C, int HW, float coef, int accumulate, int gate,
                           float *gfeat, st3d_stream_t stream);

extern "C" int st3d_gram_bwd(const float *D, const float *feat, int B, int C, int HW, float coef, int accumulate,
                             float *gfeat, st3d_stream_t stream) {
    return gram_bwd_launch(D, feat, B, C, HW, coef, accumulate, 0, gfeat, stream);
}

extern "C" int st3d_gram_bwd_gated(const float *D, const float *feat, int B, int C, int HW, float coef, int accumulate,
                                   float *gfeat, st3d_stream_t stream) {
    ST3D_CHECK_ARG(C % 32 == 0);        // the gate bits are picked up per 32-row block of the activation
    return gram_bwd_launch(D, feat, B, C, HW, coef, accumulate, 1, gfeat, stream);
}

static int gram_bwd_launch(const float *D, const float *feat, int B, int C, int HW, float coef, int accumulate, int gate,
                           float *gfeat, st3d_stream_t stream) {
    ST3D_CHECK_ARG(D && feat && gfeat);
    ST3D_CHECK_ARG(B > 0 && C > 0 && HW > 0);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.gate = gate;
    g.A = D; g.B = feat; g.C = gfeat;
    g.M = C; g.N = HW; g.K = C; g.lda = C; g.ldb = HW; g.ldc = HW;
    g.sA = (size_t)C * C; g.sB = g.sC = (size_t)C * HW;
    g.nsplit = 1; g.kper = (C + 63) / 64 * 64; g.sSplit = 0;
    g.tri = 0; g.coef = coef; g.accumulate = accumulate;
    hipStream_t s = st3d::as_stream(stream);
    // 128-row tiles only where they leave enough workgroups (>= 1024) and D is wide (measured, tools/gram_bwd_sweep.py:
    // C = 128 at 256^2 runs 18 % faster on 64-row tiles, conv5_1's 32^2 14 %; C >= 256 at >= 64^2 is indifferent)
    const char *force = getenv("ST3D_GRAM_BWD_MT");             // tuning knob: "1" / "2" force 64- / 128-row tiles
    bool tall = C % 128 == 0 && C >= 256 && (long)(C / 128) * st3d::cdiv(HW, 128) * B >= 1024;
    if (force && force[0] == '1') tall = false;
    if (force && force[0] == '2' && C % 128 == 0) tall = true;
    // C = 128 at large HW: 128 rows x 64 pixels per workgroup -- F is streamed once (64-row tiles read it twice) while the
    // grid keeps as many workgroups as the 64 x 128 tiling
    // (round 2, after the gates moved here: 128 x 64 measured fastest for every C % 128 == 0 -- 256 at 128^2 199 vs 220 us,
    // 512 at 64^2 183 vs 192, 512 at 32^2 54 vs 58)
    bool wide = C % 128 == 0;
    if (force) wide = force[0] == '3' && C % 128 == 0;
    static const bool k64 = [] { const char *e = getenv("ST3D_GRAM_BWD_K64"); return e && e[0] == '1'; }();
    if (force && force[0] == '4') {             // 64 rows x 256 pixels (1 KB row segments)
        g.tiles_m = st3d::cdiv(C, 64); g.tiles_n = st3d::cdiv(HW, 256);
        gemm_kernel<1, 4, 1><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
    // (the branch-free FAST instantiation is forward-only: measured 3-8 % SLOWER on the backward shapes, with the explicit
    // schedule -- one or two LDS reads per MFMA -- and without it)
    static const bool sym = [] { const char *e = getenv("ST3D_GRAM_BWD_SYM"); return !(e && e[0] == '0'); }();
    if (gate && sym && !force && !k64 && C % 128 == 0 && HW % 64 == 0 && (size_t)C * HW * 4 < (1ull << 31) &&
        (((uintptr_t)D | (uintptr_t)feat | (uintptr_t)gfeat) & 15) == 0) {
        // the plan's call (symmetric D, whole tiles): the lean kernel
        g.tiles_m = C / 128; g.tiles_n = HW / 64;
        gram_bwd_sym_kernel<<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
        ST3D_LAUNCH_CHECK();
        return ST3D_OK;
    }
    if (wide) {
        g.tiles_m = C / 128; g.tiles_n = st3d::cdiv(HW, 64);
        if (k64) gemm_kernel<2, 1, 1, 0, 64><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
        else gemm_kernel<2, 1, 1><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
    } else if (C <= 128 && !tall && k64) {
        // short K (= C): the whole reduction (C = 64) or half of it is staged at once, so a workgroup has its operand tile and
        // the accumulate tile in flight together instead of one 32-row chunk after the other
        g.tiles_m = st3d::cdiv(C, 64); g.tiles_n = st3d::cdiv(HW, 128);
        gemm_kernel<1, 2, 1, 0, 64><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
    } else if (tall) {
        g.tiles_m = C / 128; g.tiles_n = st3d::cdiv(HW, 128);
        gemm_kernel<2, 2, 1><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
    } else {
        g.tiles_m = st3d::cdiv(C, 64); g.tiles_n = st3d::cdiv(HW, 128);
        gemm_kernel<1, 2, 1><<<dim3(g.tiles_m * g.tiles_n, 1, B), 256, 0, s>>>(g);
    }
    ST3D_LAUNCH_CHECK();
    return ST3D_OK;
}
