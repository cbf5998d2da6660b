// plan.hip -- host-side engine of libst3d: error reporting, the frozen VGG-19 weight store
// (utils.py:48-52) and the fused perceptual-loss plan (losses.py:12-44 /
// style_transfer.py:59-83): forward to conv5_1 keeping the post-ReLU taps, Gram + content
// losses, and the hand-scheduled backward to d loss / d image.  The graph is static, so the
// backward is a fixed launch sequence over preallocated workspaces (no autograd tape, no
// allocation per step); what the reference recomputes every call but does not depend on the
// optimised parameters (content features, style Grams: losses.py:18-25) is set once through
// st3d_plan_set_content / st3d_plan_set_style.
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace st3d {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace st3d

extern "C" int st3d_version(void) { return 100; }
extern "C" const char *st3d_last_error(void) { return st3d::g_err; }

extern "C" int st3d_device_info(int device, int *cu_count, size_t *hbm_bytes, char *name, int name_len) {
    hipDeviceProp_t p;
    ST3D_HIP(hipGetDeviceProperties(&p, device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return ST3D_OK;
}

namespace {

// torchvision vgg19().features layout: module index -> kind
constexpr int kModules = 37;
const int kConvIdx[16] = {0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34};
const int kConvCin[16] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512};
const int kConvCout[16] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512};
const int kPoolIdx[5] = {4, 9, 18, 27, 36};
// style taps (style_transfer.py:12-19 minus conv4_2) and the content tap
const int kStyleTap[5] = {0, 5, 10, 19, 28};
constexpr int kContentTap = 21;

int conv_slot(int module_idx) {
    for (int i = 0; i < 16; ++i)
        if (kConvIdx[i] == module_idx) return i;
    return -1;
}
int pool_slot(int module_idx) {
    for (int i = 0; i < 5; ++i)
        if (kPoolIdx[i] == module_idx) return i;
    return -1;
}

}  // namespace

struct st3d_vgg {
    float *wf[16];     // direct implicit-GEMM packs (conv.hip)
    float *wd[16];
    float *uf[16];     // Winograd packs (wino.hip), nullptr where the layer shape is not supported
    float *ud[16];
    float *u6f[16];    // Winograd F(4x4,3x3) packs (wino43.hip), nullptr where not supported
    float *u6d[16];
    int wino43_mink;   // layers with at least this many input channels (K of the GEMM) run F(4x4,3x3) where the shape allows;
                       // 0 = never (ST3D_WINO43=0); ST3D_WINO43_MINK overrides the measured default
    float *bias[16];
    bool set[16];
    bool use_wino;     // ST3D_CONV=direct forces the direct kernels (A/B runs)
    bool pregate;      // ST3D_PREGATE=0: every input-gradient launch applies its own ReLU gate (A/B runs)
    bool fuse_tap0;    // ST3D_TAP0_FUSED=0: relu1_1 Gram backward and conv1_1 input gradient as separate launches (A/B runs)
};

extern "C" int st3d_vgg_create(st3d_vgg **out) {
    ST3D_CHECK_ARG(out);
    st3d_vgg *v = new st3d_vgg();
    memset(v, 0, sizeof(*v));
    const char *mode = getenv("ST3D_CONV");
    v->use_wino = !(mode && strcmp(mode, "direct") == 0);
    const char *t0 = getenv("ST3D_TAP0_FUSED");
    v->fuse_tap0 = !(t0 && t0[0] == '0');
    const char *pg = getenv("ST3D_PREGATE");
    v->pregate = !(pg && pg[0] == '0');
    const char *w6 = getenv("ST3D_WINO43"), *w6k = getenv("ST3D_WINO43_MINK");
    v->wino43_mink = (w6 && w6[0] == '0') ? 0 : (w6k ? atoi(w6k) : 64);
    for (int i = 0; i < 16; ++i) {
        const size_t n = st3d_conv3x3_packed_floats(kConvCout[i], kConvCin[i]);
        bool ok = hipMalloc(&v->wf[i], n * sizeof(float)) == hipSuccess && hipMalloc(&v->wd[i], n * sizeof(float)) == hipSuccess &&
                  hipMalloc(&v->bias[i], kConvCout[i] * sizeof(float)) == hipSuccess;
        // both directions must be Winograd-able (dgrad swaps the channel roles)
        if (ok && st3d_wino_supported(kConvCin[i], kConvCout[i], 4, 4) && st3d_wino_supported(kConvCout[i], kConvCin[i], 4, 4)) {
            const size_t nu = st3d_wino_packed_floats(kConvCout[i], kConvCin[i]);
            ok = hipMalloc(&v->uf[i], nu * sizeof(float)) == hipSuccess && hipMalloc(&v->ud[i], nu * sizeof(float)) == hipSuccess;
        }
        if (ok && v->use_wino && v->wino43_mink > 0 && kConvCin[i] % 64 == 0 && kConvCout[i] % 64 == 0 &&
            (kConvCin[i] >= v->wino43_mink || kConvCout[i] >= v->wino43_mink)) {
            const size_t nu = st3d_wino43_packed_floats(kConvCout[i], kConvCin[i]);
            ok = hipMalloc(&v->u6f[i], nu * sizeof(float)) == hipSuccess && hipMalloc(&v->u6d[i], nu * sizeof(float)) == hipSuccess;
        }
        if (!ok) {
            st3d::set_error("st3d_vgg_create: hipMalloc failed");
            st3d_vgg_destroy(v);
            return ST3D_E_NOMEM;
        }
    }
    *out = v;
    return ST3D_OK;
}

extern "C" int st3d_vgg_set_conv(st3d_vgg *vgg, int module_idx, const float *w, const float *b, st3d_stream_t stream) {
    ST3D_CHECK_ARG(vgg && w && b);
    const int s = conv_slot(module_idx);
    ST3D_CHECK_ARG(s >= 0);
    ST3D_TRY(st3d_conv3x3_pack(w, kConvCout[s], kConvCin[s], vgg->wf[s], vgg->wd[s], stream));
    if (vgg->uf[s]) ST3D_TRY(st3d_wino_pack(w, kConvCout[s], kConvCin[s], vgg->uf[s], vgg->ud[s], stream));
    if (vgg->u6f[s]) ST3D_TRY(st3d_wino43_pack(w, kConvCout[s], kConvCin[s], vgg->u6f[s], vgg->u6d[s], stream));
    ST3D_HIP(hipMemcpyAsync(vgg->bias[s], b, kConvCout[s] * sizeof(float), hipMemcpyDeviceToDevice, st3d::as_stream(stream)));
    vgg->set[s] = true;
    return ST3D_OK;
}

extern "C" int st3d_vgg_destroy(st3d_vgg *vgg) {
    if (!vgg) return ST3D_OK;
    for (int i = 0; i < 16; ++i) {
        if (vgg->wf[i]) (void)hipFree(vgg->wf[i]);
        if (vgg->wd[i]) (void)hipFree(vgg->wd[i]);
        if (vgg->uf[i]) (void)hipFree(vgg->uf[i]);
        if (vgg->ud[i]) (void)hipFree(vgg->ud[i]);
        if (vgg->u6f[i]) (void)hipFree(vgg->u6f[i]);
        if (vgg->u6d[i]) (void)hipFree(vgg->u6d[i]);
        if (vgg->bias[i]) (void)hipFree(vgg->bias[i]);
    }
    delete vgg;
    return ST3D_OK;
}

struct st3d_plan {
    st3d_vgg *vgg;
    int B, S;
    // per module: output activation (conv: post-ReLU; relu: alias of its conv; pool: pooled)
    float *act[kModules];
    int C[kModules], H[kModules], W[kModules];
    uint8_t *pidx[5];
    float *gbuf[2];
    size_t gbuf_floats;
    // targets
    float *content_target;   // (B, 512, S/8, S/8)
    float *style_gram[5];     // (B or 1, C, C)
    int style_batch;
    bool have_content, have_style;
    // per-step scratch
    float *gram[5], *D[5];
    void *gram_ws;
    size_t gram_ws_bytes;
    float *partials;
    size_t bytes;
    int last_n;
    // profiling
    bool prof;
    struct Ev { int fam, module; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> pool;
    float fam_ms[ST3D_PROFILE_FAMILIES];
    int fam_n[ST3D_PROFILE_FAMILIES];
    std::vector<int> l_tag;        // per launch since the last read: family * 100 + VGG module index
    std::vector<float> l_ms;
    // HIP-graph replay of the loss step (st3d_plan_graph): the launch sequence is static, so after one ordinary call it
    // is captured once per (n, batch_denom, weights) and replayed; it works on plan-owned staging buffers because a
    // captured kernel's pointers are baked in while the caller's tensors move
    int use_graph;
    hipGraphExec_t gexec;
    hipStream_t cap_stream;
    float *g_in, *g_grad, *g_loss;
    struct { int n, denom, want_grad, warm; float sw, cw; } gkey;
};

namespace {

template <typename T>
int dev_alloc(st3d_plan *p, T **ptr, size_t count) {
    const size_t bytes = count * sizeof(T);
    if (hipMalloc(reinterpret_cast<void **>(ptr), bytes) != hipSuccess) {
        st3d::set_error("st3d_plan_create: hipMalloc(%zu bytes) failed", bytes);
        return ST3D_E_NOMEM;
    }
    p->bytes += bytes;
    // ST3D_POISON_PLAN=1 (tests): the plan's buffers start as 0xFF.. (NaN / -1) instead of whatever hipMalloc returns, so a
    // launch that reads a buffer before the sequence has written it changes the result
    static const bool poison = [] { const char *e = getenv("ST3D_POISON_PLAN"); return e && e[0] == '1'; }();
    if (poison) (void)hipMemset(*ptr, 0xFF, bytes);
    return ST3D_OK;
}

struct Scope {   // HIP-event bracket around one kernel family (only when profiling is on)
    st3d_plan *p; int fam, module; hipStream_t s; hipEvent_t a, b; bool on;
    Scope(st3d_plan *p_, int fam_, hipStream_t s_, int module_ = 99) : p(p_), fam(fam_), module(module_), s(s_), on(p_->prof) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!p->pool.empty()) { e = p->pool.back(); p->pool.pop_back(); } else { (void)hipEventCreate(&e); }
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~Scope() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        p->evs.push_back({fam, module, a, b});
    }
};

// F_CONV_*: the Winograd launches; F_CONVX_*: the convs Winograd does not cover (conv1_1, odd shapes, ST3D_CONV=direct)
// F_CONV43_*: the launches that ran Winograd F(4x4,3x3) (wino43.hip: 2.25 instead of 4 MFMA-multiplies per output)
enum { F_CONV_FWD = 0, F_CONV_DGRAD = 1, F_POOL = 2, F_GRAM_FWD = 3, F_GRAM_BWD = 4, F_ELEM = 5, F_CONVX_FWD = 6, F_CONVX_DGRAD = 7,
       F_CONV43_FWD = 8, F_CONV43_DGRAD = 9 };

// F(4x4,3x3) for this GEMM?  K = channels reduced over (Cin forward, Cout for the input gradient), M = channels produced
bool use_wino43(const st3d_vgg *v, int cs, int K, int M, int H, int W) {
    return v->use_wino && v->wino43_mink > 0 && v->u6f[cs] && K >= v->wino43_mink && st3d_wino43_supported(K, M, H, W);
}

// keep_full: also materialise the full-resolution output of convs whose 2x2 pool is fused into
// their epilogue (needed only when a caller asks for that activation: st3d_plan_forward).
int forward(st3d_plan *p, const float *imgs, int n, int upto, bool keep_full, hipStream_t s) {
    const float *x = imgs;
    int Cin = 3, H = p->S, W = p->S;
    for (int m = 0; m <= upto; ++m) {
        const int cs = conv_slot(m), ps = pool_slot(m);
        if (cs >= 0) {
            if (!p->vgg->set[cs]) {
                st3d::set_error("st3d_plan_forward: weights of module %d were never set", m);
                return ST3D_E_STATE;
            }
            const bool wino = p->vgg->use_wino && p->vgg->uf[cs] && st3d_wino_supported(Cin, kConvCout[cs], H, W);
            const bool w43 = wino && use_wino43(p->vgg, cs, Cin, kConvCout[cs], H, W);
            Scope sc(p, w43 ? F_CONV43_FWD : (wino ? F_CONV_FWD : F_CONVX_FWD), s, m);
            if (wino) {
                const int pool_m = m + 2;          // conv, relu, pool
                const int pps = (pool_m <= upto) ? pool_slot(pool_m) : -1;
                float *yfull = (pps < 0 || keep_full) ? p->act[m] : nullptr;
                if (w43)
                    ST3D_TRY(st3d_wino43_fwd(x, p->vgg->u6f[cs], p->vgg->bias[cs], yfull, pps >= 0 ? p->act[pool_m] : nullptr,
                                             pps >= 0 ? p->pidx[pps] : nullptr, n, Cin, kConvCout[cs], H, W, 1, s));
                else
                ST3D_TRY(st3d_wino_fwd(x, p->vgg->uf[cs], p->vgg->bias[cs], yfull, pps >= 0 ? p->act[pool_m] : nullptr,
                                       pps >= 0 ? p->pidx[pps] : nullptr, n, Cin, kConvCout[cs], H, W, 1, s));
                Cin = kConvCout[cs];
                if (pps >= 0) {                     // pool output produced by the conv epilogue: skip modules m+1, m+2
                    x = p->act[pool_m];
                    H /= 2; W /= 2;
                    m = pool_m;
                } else {
                    x = p->act[m];
                }
            } else {
                ST3D_TRY(st3d_conv3x3_fwd(x, p->vgg->wf[cs], p->vgg->bias[cs], p->act[m], n, Cin, kConvCout[cs], H, W, 1, s));
                x = p->act[m];
                Cin = kConvCout[cs];
            }
        } else if (ps >= 0) {
            Scope sc(p, F_POOL, s, m);
            ST3D_TRY(st3d_maxpool2x2_fwd(x, p->act[m], p->pidx[ps], n, Cin, H, W, s));
            x = p->act[m];
            H /= 2; W /= 2;
        }   // ReLU modules are fused into their conv
    }
    p->last_n = n;
    return ST3D_OK;
}

// g = (accumulate ? g : 0) + x
__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ x, size_t n, int accumulate, float *__restrict__ g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = accumulate ? g[i] + x[i] : x[i];
}

// backward of MaxPool2d(2,2) as a stand-alone pass: out (planes,H,W) = scatter of gp (planes,H/2,W/2) to the argmax
// positions (+ `add`, a full-resolution gradient arriving at the same tensor, when given).  Only the differentiable
// get_features needs it (a tap on a conv that feeds a pool); the loss plan fuses the unpool into the dgrad kernel.
__global__ __launch_bounds__(256) void unpool_add_kernel(const float *__restrict__ gp, const uint8_t *__restrict__ idx,
                                                         const float *__restrict__ add, size_t planes, int H, int W,
                                                         float *__restrict__ out) {
    const int Hp = H / 2, Wp = W / 2;
    const size_t n = planes * (size_t)Hp * Wp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int xo = (int)(i % Wp), yo = (int)((i / Wp) % Hp);
    const size_t pl = i / ((size_t)Wp * Hp);
    const float g = gp ? gp[i] : 0.f;
    const int k = gp ? idx[i] : -1;
    const size_t base = pl * H * W + (size_t)(2 * yo) * W + 2 * xo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t o = base + (size_t)(q >> 1) * W + (q & 1);
        out[o] = (q == k ? g : 0.f) + (add ? add[o] : 0.f);
    }
}

bool dgrad_is_wino(const st3d_plan *p, int cs) {
    const int m = kConvIdx[cs];
    return p->vgg->use_wino && p->vgg->ud[cs] && st3d_wino_supported(kConvCout[cs], kConvCin[cs], p->H[m], p->W[m]);
}

// one input-gradient launch of conv slot cs: g (gradient w.r.t. the conv's post-ReLU output, or w.r.t. the output of
// the pool behind it when pooled) -> dst (gradient w.r.t. the conv's input).  pregated: g is already zero where the gate
// this launch would apply is closed; out_gate: zero dst where this tensor is <= 0 (the next link's gate, Winograd only)
int dgrad_step(st3d_plan *p, int cs, const float *g, bool g_is_pooled, int pool_of_g, float *dst, int n, hipStream_t s,
               bool pregated = false, const float *out_gate = nullptr, const float *add_target = nullptr, float add_coef = 0.f) {
    const int m = kConvIdx[cs];
    const int H = p->H[m], W = p->W[m];
    const bool wino = dgrad_is_wino(p, cs);
    // F(4x4,3x3) takes an already gated gradient only (what the producer-gated chain hands on)
    const bool w43 = wino && pregated && use_wino43(p->vgg, cs, kConvCout[cs], kConvCin[cs], H, W);
    Scope sc(p, w43 ? F_CONV43_DGRAD : (wino ? F_CONV_DGRAD : F_CONVX_DGRAD), s, m);
    if (w43) {
        ST3D_TRY(st3d_wino43_dgrad_chain(g, g_is_pooled ? p->pidx[pool_of_g] : nullptr, p->vgg->u6d[cs], out_gate, add_target,
                                         add_coef, dst, n, kConvCin[cs], kConvCout[cs], H, W, s));
        return ST3D_OK;
    }
    if (wino && (pregated || out_gate)) {
        const uint8_t *pidx = g_is_pooled ? p->pidx[pool_of_g] : nullptr;
        const float *pooled = (g_is_pooled && !pregated) ? p->act[kPoolIdx[pool_of_g]] : nullptr;
        const float *act = (!g_is_pooled && !pregated) ? p->act[m] : nullptr;
        ST3D_TRY(st3d_wino_dgrad_chain(g, act, pidx, pooled, p->vgg->ud[cs], out_gate, add_target, add_coef, dst, n,
                                       kConvCin[cs], kConvCout[cs], H, W, s));
        return ST3D_OK;
    }
    if (g_is_pooled) {
        if (wino)
            ST3D_TRY(st3d_wino_dgrad_unpool(g, p->pidx[pool_of_g], p->act[kPoolIdx[pool_of_g]], p->vgg->ud[cs], dst, n,
                                            kConvCin[cs], kConvCout[cs], H, W, s));
        else
            ST3D_TRY(st3d_conv3x3_dgrad_unpool(g, p->pidx[pool_of_g], p->act[kPoolIdx[pool_of_g]], p->vgg->wd[cs], dst, n,
                                               kConvCin[cs], kConvCout[cs], H, W, s));
    } else {
        if (wino)
            ST3D_TRY(st3d_wino_dgrad(g, p->act[m], p->vgg->ud[cs], dst, n, kConvCin[cs], kConvCout[cs], H, W, s));
        else
            ST3D_TRY(st3d_conv3x3_dgrad(g, p->act[m], p->vgg->wd[cs], dst, n, kConvCin[cs], kConvCout[cs], H, W, s));
    }
    return ST3D_OK;
}

}  // namespace

extern "C" int st3d_plan_create(st3d_plan **out, st3d_vgg *vgg, int B, int S) {
    ST3D_CHECK_ARG(out && vgg);
    ST3D_CHECK_ARG(B > 0 && S >= 16);      // any size: shapes the Winograd kernels do not cover (odd H, W % 4) run on the direct ones, pools floor like MaxPool2d
    st3d_plan *p = new st3d_plan();
    p->vgg = vgg; p->B = B; p->S = S; p->bytes = 0;
    for (int m = 0; m < kModules; ++m) { p->act[m] = nullptr; p->C[m] = p->H[m] = p->W[m] = 0; }
    for (int i = 0; i < 5; ++i) { p->pidx[i] = nullptr; p->style_gram[i] = p->gram[i] = p->D[i] = nullptr; }
    p->gbuf[0] = p->gbuf[1] = nullptr; p->content_target = nullptr; p->gram_ws = nullptr; p->partials = nullptr;
    p->have_content = p->have_style = false; p->style_batch = 0; p->last_n = 0; p->prof = false;
    p->use_graph = 0; p->gexec = nullptr; p->cap_stream = nullptr; p->g_in = p->g_grad = p->g_loss = nullptr; memset(&p->gkey, 0, sizeof(p->gkey));
    memset(p->fam_ms, 0, sizeof(p->fam_ms)); memset(p->fam_n, 0, sizeof(p->fam_n));
    int rc = ST3D_OK;
    int C = 3, H = S, W = S;
    size_t gmax = 0, wsmax = 0;
    for (int m = 0; m < kModules && rc == ST3D_OK; ++m) {
        const int cs = conv_slot(m), ps = pool_slot(m);
        if (cs >= 0) {
            C = kConvCout[cs];
            rc = dev_alloc(p, &p->act[m], (size_t)B * C * H * W);
            if ((size_t)B * C * H * W > gmax) gmax = (size_t)B * C * H * W;
        } else if (ps >= 0) {
            H /= 2; W /= 2;
            rc = dev_alloc(p, &p->act[m], (size_t)B * C * H * W);
            if (rc == ST3D_OK) rc = dev_alloc(p, &p->pidx[ps], (size_t)B * C * H * W);
        } else {
            p->act[m] = p->act[m - 1];   // in-place ReLU: the tap tensor IS the post-ReLU output
        }
        p->C[m] = C; p->H[m] = H; p->W[m] = W;
    }
    p->gbuf_floats = gmax;
    for (int i = 0; i < 2 && rc == ST3D_OK; ++i) rc = dev_alloc(p, &p->gbuf[i], gmax);
    if (rc == ST3D_OK) rc = dev_alloc(p, &p->content_target, (size_t)B * p->C[kContentTap] * p->H[kContentTap] * p->W[kContentTap]);
    for (int i = 0; i < 5 && rc == ST3D_OK; ++i) {
        const int m = kStyleTap[i];
        const size_t cc = (size_t)B * p->C[m] * p->C[m];
        rc = dev_alloc(p, &p->style_gram[i], cc);
        if (rc == ST3D_OK) rc = dev_alloc(p, &p->gram[i], cc);
        if (rc == ST3D_OK) rc = dev_alloc(p, &p->D[i], cc);
    }
    // every style layer has its own split-K slab region: the five Grams of a step run as ONE launch (st3d_gram_fwd_multi).
    // Sized for the worst batch 1..B (the split count is rounded per batch size, so n < B can need a little more than B).
    for (int n = 1; n <= B && rc == ST3D_OK; ++n) {
        st3d_gram_item items[5];
        for (int i = 0; i < 5; ++i) items[i] = st3d_gram_item{nullptr, nullptr, n, p->C[kStyleTap[i]], p->H[kStyleTap[i]] * p->W[kStyleTap[i]]};
        const size_t ws = st3d_gram_multi_workspace_bytes(items, 5);
        if (ws > wsmax) wsmax = ws;
    }
    p->gram_ws_bytes = wsmax;
    if (rc == ST3D_OK) { float *t = nullptr; rc = dev_alloc(p, &t, wsmax / sizeof(float)); p->gram_ws = t; }
    if (rc == ST3D_OK) rc = dev_alloc(p, &p->partials, (size_t)8 * st3d_reduce_partials());
    if (rc == ST3D_OK) rc = dev_alloc(p, &p->g_in, (size_t)B * 3 * S * S);
    if (rc == ST3D_OK) rc = dev_alloc(p, &p->g_grad, (size_t)B * 3 * S * S);
    if (rc == ST3D_OK) rc = dev_alloc(p, &p->g_loss, (size_t)4);
    if (rc != ST3D_OK) { st3d_plan_destroy(p); return rc; }
    *out = p;
    return ST3D_OK;
}

extern "C" int st3d_plan_destroy(st3d_plan *p) {
    if (!p) return ST3D_OK;
    for (int m = 0; m < kModules; ++m)
        if (p->act[m] && (m == 0 || p->act[m] != p->act[m - 1])) (void)hipFree(p->act[m]);
    for (int i = 0; i < 5; ++i) {
        if (p->pidx[i]) (void)hipFree(p->pidx[i]);
        if (p->style_gram[i]) (void)hipFree(p->style_gram[i]);
        if (p->gram[i]) (void)hipFree(p->gram[i]);
        if (p->D[i]) (void)hipFree(p->D[i]);
    }
    for (int i = 0; i < 2; ++i)
        if (p->gbuf[i]) (void)hipFree(p->gbuf[i]);
    if (p->content_target) (void)hipFree(p->content_target);
    if (p->gram_ws) (void)hipFree(p->gram_ws);
    if (p->partials) (void)hipFree(p->partials);
    if (p->g_in) (void)hipFree(p->g_in);
    if (p->g_grad) (void)hipFree(p->g_grad);
    if (p->g_loss) (void)hipFree(p->g_loss);
    if (p->gexec) (void)hipGraphExecDestroy(p->gexec);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    for (auto &e : p->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : p->pool) (void)hipEventDestroy(e);
    delete p;
    return ST3D_OK;
}

extern "C" size_t st3d_plan_bytes(const st3d_plan *p) { return p ? p->bytes : 0; }

extern "C" int st3d_plan_forward(st3d_plan *p, const float *imgs, int n, int upto_module, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && imgs);
    ST3D_CHECK_ARG(n > 0 && n <= p->B && upto_module >= 0 && upto_module < kModules);
    st3d::TraceRange tr("vgg_forward");
    return forward(p, imgs, n, upto_module, true, st3d::as_stream(stream));
}

extern "C" int st3d_plan_activation(st3d_plan *p, int module_idx, float **ptr, int *C, int *H, int *W) {
    ST3D_CHECK_ARG(p && ptr && module_idx >= 0 && module_idx < kModules);
    *ptr = p->act[module_idx];
    if (C) *C = p->C[module_idx];
    if (H) *H = p->H[module_idx];
    if (W) *W = p->W[module_idx];
    return ST3D_OK;
}

extern "C" int st3d_plan_set_content(st3d_plan *p, const float *content, int n, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && content && n > 0 && n <= p->B);
    hipStream_t s = st3d::as_stream(stream);
    ST3D_TRY(forward(p, content, n, kContentTap, false, s));
    const size_t cnt = (size_t)n * p->C[kContentTap] * p->H[kContentTap] * p->W[kContentTap];
    ST3D_HIP(hipMemcpyAsync(p->content_target, p->act[kContentTap], cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
    p->have_content = true;
    return ST3D_OK;
}

// content features (conv4_2 of the content images) out of / into the plan: lets a caller that alternates between several
// view batches keep each batch's target and skip recomputing it (a device copy instead of a VGG forward to conv4_2)
extern "C" int st3d_plan_get_content_features(st3d_plan *p, float *out, int n, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && out && n > 0 && n <= p->B);
    if (!p->have_content) {
        st3d::set_error("st3d_plan_get_content_features: no content target set");
        return ST3D_E_STATE;
    }
    const size_t cnt = (size_t)n * p->C[kContentTap] * p->H[kContentTap] * p->W[kContentTap];
    ST3D_HIP(hipMemcpyAsync(out, p->content_target, cnt * sizeof(float), hipMemcpyDeviceToDevice, st3d::as_stream(stream)));
    return ST3D_OK;
}

extern "C" int st3d_plan_set_content_features(st3d_plan *p, const float *feat, int n, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && feat && n > 0 && n <= p->B);
    const size_t cnt = (size_t)n * p->C[kContentTap] * p->H[kContentTap] * p->W[kContentTap];
    ST3D_HIP(hipMemcpyAsync(p->content_target, feat, cnt * sizeof(float), hipMemcpyDeviceToDevice, st3d::as_stream(stream)));
    p->have_content = true;
    return ST3D_OK;
}

// the five style taps' Grams (of the n images of the last forward) in one launch pair
static int grams_of_taps(st3d_plan *p, int n, float *const *out, hipStream_t s) {
    st3d_gram_item items[5];
    for (int i = 0; i < 5; ++i) {
        const int m = kStyleTap[i];
        items[i] = st3d_gram_item{p->act[m], out[i], n, p->C[m], p->H[m] * p->W[m]};
    }
    return st3d_gram_fwd_multi(items, 5, p->gram_ws, p->gram_ws_bytes, s);
}

extern "C" int st3d_plan_set_style(st3d_plan *p, const float *style, int style_batch, int n, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && style && n > 0 && n <= p->B && (style_batch == 1 || style_batch == n));
    hipStream_t s = st3d::as_stream(stream);
    ST3D_TRY(forward(p, style, style_batch, 28, false, s));
    {
        Scope sc(p, F_GRAM_FWD, s);
        ST3D_TRY(grams_of_taps(p, style_batch, p->style_gram, s));
    }
    if (p->style_batch != style_batch && p->gexec) {       // a captured loss step indexes the style Grams with the old batch stride
        (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; p->gkey.warm = 0;
    }
    p->style_batch = style_batch;
    p->have_style = true;
    return ST3D_OK;
}

static int plan_loss_enqueue(st3d_plan *p, const float *current, int n, int batch_denom, float style_weight,
                             float content_weight, float *loss_out, float *grad_current, hipStream_t s);

extern "C" int st3d_plan_graph(st3d_plan *p, int enable) {
    ST3D_CHECK_ARG(p);
    p->use_graph = enable ? 1 : 0;
    if (!enable && p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    p->gkey.warm = 0;
    return ST3D_OK;
}

extern "C" int st3d_plan_loss(st3d_plan *p, const float *current, int n, int batch_denom, float style_weight,
                              float content_weight, float *loss_out, float *grad_current, st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && current && loss_out);
    ST3D_CHECK_ARG(n > 0 && n <= p->B && batch_denom >= n);
    if (!p->have_content || !p->have_style) {
        st3d::set_error("st3d_plan_loss: content/style targets not set");
        return ST3D_E_STATE;
    }
    ST3D_CHECK_ARG(p->style_batch == 1 || p->style_batch == n);
    hipStream_t s = st3d::as_stream(stream);
    if (!p->use_graph || p->prof) return plan_loss_enqueue(p, current, n, batch_denom, style_weight, content_weight, loss_out, grad_current, s);

    // ---- graph replay
    const size_t img = (size_t)n * 3 * p->S * p->S;
    const int want_grad = grad_current ? 1 : 0;
    const bool same = p->gexec && p->gkey.n == n && p->gkey.denom == batch_denom && p->gkey.want_grad == want_grad &&
                      p->gkey.sw == style_weight && p->gkey.cw == content_weight;
    if (!same) {
        if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
        const bool warm = p->gkey.warm && p->gkey.n == n && p->gkey.denom == batch_denom && p->gkey.want_grad == want_grad &&
                          p->gkey.sw == style_weight && p->gkey.cw == content_weight;
        p->gkey.n = n; p->gkey.denom = batch_denom; p->gkey.want_grad = want_grad; p->gkey.sw = style_weight; p->gkey.cw = content_weight;
        if (!warm) {            // first call with these parameters: run it plainly (loads every code object, nothing to capture yet)
            p->gkey.warm = 1;
            return plan_loss_enqueue(p, current, n, batch_denom, style_weight, content_weight, loss_out, grad_current, s);
        }
        hipGraph_t graph = nullptr;
        // captured on a stream of the plan's own: the caller's stream is usually the (uncapturable) default stream
        if (!p->cap_stream) ST3D_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
        ST3D_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = plan_loss_enqueue(p, p->g_in, n, batch_denom, style_weight, content_weight, p->g_loss,
                                         want_grad ? p->g_grad : nullptr, p->cap_stream);
        const hipError_t e = hipStreamEndCapture(p->cap_stream, &graph);
        if (rc != ST3D_OK || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            if (rc == ST3D_OK) st3d::set_error("st3d_plan_loss: stream capture failed: %s", hipGetErrorString(e));
            return rc != ST3D_OK ? rc : ST3D_E_HIP;
        }
        const hipError_t ei = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { p->gexec = nullptr; st3d::set_error("st3d_plan_loss: hipGraphInstantiate: %s", hipGetErrorString(ei)); return ST3D_E_HIP; }
    }
    ST3D_HIP(hipMemcpyAsync(p->g_in, current, img * sizeof(float), hipMemcpyDeviceToDevice, s));
    ST3D_HIP(hipGraphLaunch(p->gexec, s));
    ST3D_HIP(hipMemcpyAsync(loss_out, p->g_loss, 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (grad_current) ST3D_HIP(hipMemcpyAsync(grad_current, p->g_grad, img * sizeof(float), hipMemcpyDeviceToDevice, s));
    p->last_n = n;
    return ST3D_OK;
}

static int plan_loss_enqueue(st3d_plan *p, const float *current, int n, int batch_denom, float style_weight,
                             float content_weight, float *loss_out, float *grad_current, hipStream_t s) {
    {
        st3d::TraceRange tr("vgg_forward");
        ST3D_TRY(forward(p, current, n, 28, false, s));
    }
    st3d::TraceRange tr_loss("gram_and_losses");

    const double bd = (double)batch_denom;
    // content loss: mean over (B,C,H,W) of (F - Ft)^2            (losses.py:31)
    // style loss: sum_l mean over (B,C,C) of (G - S)^2 / (C^2 H^2) (losses.py:34-39)
    // -- the five Grams first, then all six squared-difference sums, their finish and the weighted total in one launch pair
    const int mc = kContentTap;
    const size_t chw = (size_t)p->C[mc] * p->H[mc] * p->W[mc];
    st3d_sqdiff_item items[6];
    items[0] = st3d_sqdiff_item{p->act[mc], p->content_target, nullptr, (size_t)n * chw, (size_t)n * chw,
                                (float)(1.0 / (bd * (double)chw)), 1};
    float style_coef[5];
    {
        Scope sc(p, F_GRAM_FWD, s);
        ST3D_TRY(grams_of_taps(p, n, p->gram, s));
    }
    for (int i = 0; i < 5; ++i) {
        const int m = kStyleTap[i];
        const double C = p->C[m], Hh = p->H[m];
        const size_t cc = (size_t)p->C[m] * p->C[m];
        const double norm = 1.0 / (bd * C * C) / (C * C * Hh * Hh);
        items[1 + i] = st3d_sqdiff_item{p->gram[i], p->style_gram[i], p->D[i], (size_t)n * cc,
                                        p->style_batch == 1 ? cc : (size_t)n * cc, (float)norm, 2};
        style_coef[i] = (float)(4.0 * (double)style_weight * norm);   // d/dF = 2*(dG + dG^T)/2 ... = 4 w norm D F
    }
    {
        Scope sc(p, F_ELEM, s);
        ST3D_TRY(st3d_sqdiff_sum_multi(items, 6, p->partials, loss_out, 1, 1, style_weight, content_weight, s));
    }
    if (!grad_current) return ST3D_OK;
    st3d_trace_pop();                                  // (gram_and_losses ends here; its guard pops "vgg_backward" below)
    st3d_trace_push("vgg_backward");

    // ---- backward: gradient w.r.t. the post-ReLU output of each conv, top down
    float *g = p->gbuf[0], *gn = p->gbuf[1];
    bool have_g = false, g_is_pooled = false, g_gated = false, content_done = false;
    int pool_of_g = -1;
    const float cc = (float)(2.0 * (double)content_weight / (bd * (double)chw));       // d content / d conv4_2 = cc * (F - target)
    for (int cs = 12; cs >= 0; --cs) {          // conv slots 12 (module 28) .. 0
        const int m = kConvIdx[cs];
        const int C = p->C[m], H = p->H[m], W = p->W[m];
        int st = -1;
        for (int i = 0; i < 5; ++i)
            if (kStyleTap[i] == m) st = i;
        if (cs == 0 && (st >= 0 || have_g) && p->vgg->fuse_tap0 && C == 64 && st3d_conv1_bwd_supported(H, W) &&
            p->gbuf_floats >= (size_t)n * 27 * H * W) {
            // relu1_1: style gradient + ReLU gate + conv1_1 input gradient in one pass over g and F (tap0.hip); the 27 tap
            // planes go through the idle gradient buffer
            Scope sc(p, F_CONVX_DGRAD, s, m);
            ST3D_TRY(st3d_conv1_bwd(have_g ? g : nullptr, p->act[m], st >= 0 ? p->D[st] : nullptr, st >= 0 ? style_coef[st] : 0.f,
                                    p->vgg->wd[0], gn, p->gbuf_floats * sizeof(float), grad_current, n, H, W, s));
            break;
        }
        // Producer-side ReLU gates (st3d_wino_dgrad_chain): whoever writes a gradient last zeroes it where its tensor's gate
        // is closed, so the Winograd input-gradient that consumes it streams one operand per stage
        const bool chain = p->vgg->pregate && dgrad_is_wino(p, cs) && !g_is_pooled && (C % 32) == 0;
        if (st >= 0) {
            Scope sc(p, F_GRAM_BWD, s, m);
            if (chain) {
                ST3D_TRY(st3d_gram_bwd_gated(p->D[st], p->act[m], n, C, H * W, style_coef[st], have_g ? 1 : 0, g, s));
                g_gated = true;
            } else {
                ST3D_TRY(st3d_gram_bwd(p->D[st], p->act[m], n, C, H * W, style_coef[st], have_g ? 1 : 0, g, s));
                g_gated = false;
            }
            have_g = true;
        }
        if (m == kContentTap && !content_done) {
            Scope sc(p, F_ELEM, s);
            if (chain) {
                ST3D_TRY(st3d_axpy_diff_gated(p->act[m], p->content_target, (size_t)n * C * H * W, cc, have_g ? 1 : 0, g, s));
                g_gated = true;
            } else {
                ST3D_TRY(st3d_axpy_diff(p->act[m], p->content_target, (size_t)n * C * H * W, cc, have_g ? 1 : 0, g, s));
                g_gated = false;
            }
            have_g = true;
        }
        if (!have_g) continue;
        float *dst = (cs == 0) ? grad_current : gn;
        // the tensor dst is the gradient of = this conv's forward input (previous post-ReLU output, or the pool's output):
        // gate it here when the launch that consumes it is a Winograd one
        const float *og = nullptr, *addt = nullptr;
        if (p->vgg->pregate && cs > 0 && dgrad_is_wino(p, cs) && dgrad_is_wino(p, cs - 1))
            og = p->act[pool_slot(m - 1) >= 0 ? m - 1 : m - 2];
        if (og && m - 2 == kContentTap) {       // dst is the gradient of the content tap: its own term joins in this launch's store
            addt = p->content_target;
            content_done = true;
        }
        ST3D_TRY(dgrad_step(p, cs, g, g_is_pooled, pool_of_g, dst, n, s, g_gated, og, addt, cc));
        g_gated = og != nullptr;
        // dst is the gradient w.r.t. this conv's input: either the previous conv's post-ReLU
        // output or a pool output (then the next dgrad fuses the unpool)
        g_is_pooled = (m > 0) && pool_slot(m - 1) >= 0;
        pool_of_g = g_is_pooled ? pool_slot(m - 1) : -1;
        float *t = g; g = gn; gn = t;
    }
    return ST3D_OK;
}

// Backward of st3d_plan_forward for external losses on the taps (differentiable get_features, style_transfer.py:61-83):
// grad_modules[m] (host array of kModules device pointers, NULL = no gradient) is d loss / d (output of VGG module m) for
// the n images of the LAST forward, whose activations (post-ReLU outputs, pooled values, argmax) are still in the plan.
// A conv module and the in-place ReLU behind it are the same tensor, so gradients given for either are
// summed.  -> grad_image (n,3,S,S).
extern "C" int st3d_plan_backward(st3d_plan *p, int n, int upto_module, const float *const *grad_modules, float *grad_image,
                                  st3d_stream_t stream) {
    ST3D_CHECK_ARG(p && grad_modules && grad_image);
    ST3D_CHECK_ARG(n > 0 && n <= p->B && n == p->last_n && upto_module >= 0 && upto_module < kModules);
    st3d::TraceRange tr("vgg_backward");
    hipStream_t s = st3d::as_stream(stream);
    float *g = p->gbuf[0], *gn = p->gbuf[1];
    bool have_g = false, g_is_pooled = false;
    int pool_of_g = -1;
    int top = -1;
    for (int cs = 0; cs < 16; ++cs)
        if (kConvIdx[cs] <= upto_module) top = cs;
    for (int m = 0; m < kModules; ++m)
        if (grad_modules[m] && m > upto_module) {
            st3d::set_error("st3d_plan_backward: gradient given for module %d beyond the forward's last module %d", m, upto_module);
            return ST3D_E_INVALID;
        }
    auto blocks = [](size_t cnt) { const size_t b = (cnt + 255) / 256; return (unsigned)(b > 65535 * 16 ? 65535 * 16 : b); };
    for (int cs = top; cs >= 0; --cs) {
        const int m = kConvIdx[cs];
        const int C = p->C[m], H = p->H[m], W = p->W[m];
        const size_t full = (size_t)n * C * H * W;
        const int ps = (m + 2 < kModules && m + 2 <= upto_module) ? pool_slot(m + 2) : -1;
        // (1) a gradient on the pool behind this conv joins the pooled-resolution gradient from above
        if (ps >= 0 && grad_modules[m + 2]) {
            Scope sc(p, F_ELEM, s);
            const size_t pooled = (size_t)n * C * p->H[m + 2] * p->W[m + 2];
            add_kernel<<<blocks(pooled), 256, 0, s>>>(grad_modules[m + 2], pooled, have_g ? 1 : 0, g);
            ST3D_LAUNCH_CHECK();
            have_g = true; g_is_pooled = true; pool_of_g = ps;
        }
        // (2) gradients on the conv's own (post-ReLU) output
        for (int k = 0; k < 2; ++k) {
            const float *tap = (m + k <= upto_module) ? grad_modules[m + k] : nullptr;
            if (!tap) continue;
            Scope sc(p, F_ELEM, s);
            if (have_g && g_is_pooled) {        // bring the pooled gradient to full resolution first, adding the tap on the way
                const size_t pooled = (size_t)n * C * (H / 2) * (W / 2);
                const bool odd = (H & 1) || (W & 1);          // MaxPool2d floors: the last row / column has no window
                if (odd) ST3D_HIP(hipMemsetAsync(gn, 0, full * sizeof(float), s));
                unpool_add_kernel<<<(unsigned)((pooled + 255) / 256), 256, 0, s>>>(g, p->pidx[pool_of_g], odd ? nullptr : tap,
                                                                                (size_t)n * C, H, W, gn);
                ST3D_LAUNCH_CHECK();
                if (odd) { add_kernel<<<blocks(full), 256, 0, s>>>(tap, full, 1, gn); ST3D_LAUNCH_CHECK(); }
                float *t = g; g = gn; gn = t;
                g_is_pooled = false; pool_of_g = -1;
            } else {
                add_kernel<<<blocks(full), 256, 0, s>>>(tap, full, have_g ? 1 : 0, g);
                ST3D_LAUNCH_CHECK();
            }
            have_g = true;
        }
        if (!have_g) continue;
        float *dst = (cs == 0) ? grad_image : gn;
        ST3D_TRY(dgrad_step(p, cs, g, g_is_pooled, pool_of_g, dst, n, s));
        g_is_pooled = (m > 0) && pool_slot(m - 1) >= 0;
        pool_of_g = g_is_pooled ? pool_slot(m - 1) : -1;
        float *t = g; g = gn; gn = t;
    }
    if (!have_g) ST3D_HIP(hipMemsetAsync(grad_image, 0, (size_t)n * 3 * p->S * p->S * sizeof(float), s));
    return ST3D_OK;
}

extern "C" int st3d_plan_profile(st3d_plan *p, int enable) {
    ST3D_CHECK_ARG(p);
    p->prof = enable != 0;
    return ST3D_OK;
}

static int drain_events(st3d_plan *p);

// per-launch records (family * 100 + VGG module index, milliseconds) gathered since the last call; returns how many
extern "C" int st3d_plan_profile_launches(st3d_plan *p, int *tags_out, float *ms_out, int capacity, int *count_out) {
    ST3D_CHECK_ARG(p && count_out && capacity >= 0);
    ST3D_TRY(drain_events(p));
    const int n = (int)p->l_tag.size();
    for (int i = 0; i < n && i < capacity; ++i) {
        if (tags_out) tags_out[i] = p->l_tag[i];
        if (ms_out) ms_out[i] = p->l_ms[i];
    }
    *count_out = n;
    if (capacity >= n) { p->l_tag.clear(); p->l_ms.clear(); }
    return ST3D_OK;
}

extern "C" int st3d_plan_profile_read(st3d_plan *p, float *ms_out, int *launches_out) {
    ST3D_CHECK_ARG(p);
    ST3D_TRY(drain_events(p));
    for (int i = 0; i < ST3D_PROFILE_FAMILIES; ++i) {
        if (ms_out) ms_out[i] = p->fam_ms[i];
        if (launches_out) launches_out[i] = p->fam_n[i];
        p->fam_ms[i] = 0.f;
        p->fam_n[i] = 0;
    }
    return ST3D_OK;
}

static int drain_events(st3d_plan *p) {
    for (auto &e : p->evs) {
        ST3D_HIP(hipEventSynchronize(e.b));
        float ms = 0.f;
        ST3D_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        p->fam_ms[e.fam] += ms;
        p->fam_n[e.fam] += 1;
        p->l_tag.push_back(e.fam * 100 + e.module);
        p->l_ms.push_back(ms);
        p->pool.push_back(e.a);
        p->pool.push_back(e.b);
    }
    p->evs.clear();
    if (p->l_tag.size() > (1u << 20)) { p->l_tag.clear(); p->l_ms.clear(); }      // nobody is reading them
    return ST3D_OK;
}
