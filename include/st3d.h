/*
 * st3d.h -- C ABI of libst3d.so: the MI355X (gfx950) device half of the 2D->3D
 * style-transfer optimisation step.
 *
 * The reference (EmaMule/2D-to-3D-Style-Transfer) is pure Python and has no FFI of its
 * own: what it binds for this path is PyTorch3D's `_C` extension, torchvision/ATen
 * operators and torch.optim.Adam.  Each entry point below cites the reference call site
 * (file:line, relative to the reference tree) whose device work it replaces.
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer owned by the caller unless it is marked `host`;
 *     opaque handles (st3d_vgg, st3d_plan) own their workspaces (hipMalloc at create).
 *   - every function returns 0 (ST3D_OK) or a negative ST3D_E_* code; st3d_last_error()
 *     returns a thread-local message.  No exceptions cross the boundary.
 *   - kernels are asynchronous on the given stream (a hipStream_t passed as void*; NULL =
 *     the default stream).  Handles are not thread-safe (the reference is single-threaded).
 *   - tensors are fp32, contiguous; images are NCHW exactly as utils.py:70-76 builds them.
 *
 * Run-time switches the shipped library reads from the environment.  The defaults are the
 * measured-best paths; every other value exists for same-box A/B runs and gives the same
 * results (to fp32 summation order where a split count changes).
 *   ST3D_CONV=direct            direct implicit-GEMM convolutions (conv.hip) instead of Winograd
 *   ST3D_WINO43=0, ST3D_WINO43_MINK=k   never run the F(4x4,3x3) Winograd kernel (wino43.hip) / only from k input channels (default 64)
 *   ST3D_W43_SLOTS=n            persistent workgroups per cout tile of the F(4x4,3x3) kernel (default: CUs / cout tiles)
 *   ST3D_W43_XCD=0              its slots in tile order instead of XCD-major
 *   ST3D_WINO_MAP=rr|xcd        block -> tile mapping of the F(2x2,3x3) Winograd launches
 *   ST3D_PREGATE=0              every input-gradient applies its own ReLU gate (consumer side)
 *   ST3D_TAP0_FUSED=0, ST3D_TAP0_J=1   separate relu1_1 Gram backward + conv1_1 input gradient / 4-byte accesses
 *   ST3D_GRAM_MULTI=0           the five Gram forwards as separate launches (st3d_gram_fwd_multi)
 *   ST3D_GRAM_MULTI_SCALE=n, ST3D_GRAM_MULTI_SCALES=a,b,c,d, ST3D_GRAM_MULTI_DEAL=1   split-K width / block order of the fused launch
 *   ST3D_GRAM_FAST=0, ST3D_GRAM_DIAG_TRI=0, ST3D_GRAM_TARGET_WGS, ST3D_GRAM_NSPLIT   Gram forward: generic / full-tile kernels, split counts
 *   ST3D_GRAM_BWD_MT, ST3D_GRAM_BWD_K64, ST3D_GRAM_BWD_SYM=0   Gram backward tile shapes / the general kernel
 *   ST3D_RASTER_BINS=0          flat face sweep instead of the coarse 64x64-pixel bins
 *   ST3D_POISON_PLAN=1          plan workspaces start as 0xFF (read-before-write detector of the tests)
 *   ST3D_ROCTX=1                roctx ranges around the phases of a step (st3d_trace_push / st3d_trace_pop below)
 * Read by the Python host (st3d/): ST3D_DETERMINISTIC=0 (float-atomic scatters), ST3D_GRAPH=1 (HIP-graph replay of
 * the loss step), ST3D_NEAR_PLANE=raise, ST3D_MAX_PLANS, ST3D_VGG19_WEIGHTS, ST3D_DIST_BACKEND, ST3D_NCCL.
 * Lab builds only (ST3D_LAB=1 python build.py; never shipped): ST3D_WINO_VARIANT=8 (the retired 8-wave Winograd
 * kernel, csrc/lab/wino8.inc), ST3D_WINO_DBGMODE / ST3D_WINO_STAMP (diagnostic instantiations, s_memtime stamps).
 */
#ifndef ST3D_H
#define ST3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST3D_OK 0
#define ST3D_E_INVALID (-1) /* bad argument / shape */
#define ST3D_E_HIP (-2)     /* a HIP runtime call failed */
#define ST3D_E_NOMEM (-3)   /* workspace allocation failed */
#define ST3D_E_STATE (-4)   /* call order violated (e.g. loss before targets) */

typedef void *st3d_stream_t;

int st3d_version(void);
const char *st3d_last_error(void);
/* host out-params; any may be NULL */
int st3d_device_info(int device, int *cu_count, size_t *hbm_bytes, char *name, int name_len);

/* ------------------------------------------------------------------ render: utils.py:65-77
 * (render_meshes -> PyTorch3D MeshRasterizer/SoftPhongShader configured at
 *  first_approach.py:107-113, second_approach.py:101-108: blur_radius=0, faces_per_pixel=1,
 *  ambient lights, FoV perspective cameras).  All B views of a batch go through one launch. */

/* verts (V,3) world; R (B,3,3), T (B,3) row-vector convention X_view = X R + T
 * (utils.py:142-149,161-168); out verts_ndc (B,V,3) = (x_ndc, y_ndc, z_view). */
int st3d_project_verts(const float *verts, int V, const float *R, const float *T, int B,
                       float inv_tan_half_fov, float *verts_ndc, st3d_stream_t stream);

/* bytes of scratch st3d_raster_fwd needs: per-view face records (48 B per face) + packed tile ranges (4 B per face) */
size_t st3d_raster_workspace_bytes(int B, int F);
/* the same + room for the coarse face bins of st3d_raster_fwd (lists of the faces touching each 64 x 64-pixel bin, built
 * per call): with a workspace of this size a tile sweeps its bin's list instead of every face (meshes of more than 2048
 * faces); with the smaller one st3d_raster_fwd keeps the flat sweep.  Same results either way. */
size_t st3d_raster_workspace_bytes_binned(int B, int F, int S);

/* Hard rasterisation (K=1, blur_radius=0, perspective-correct barycentrics).
 * faces (F,3) int32.  Outputs per view (B,S,S): pix_to_face int32 (-1 = background; the
 * reference's int64 is produced by the Python host on request), zbuf, bary (B,S,S,3),
 * dists (signed squared edge distance), -1 filled on background. */
int st3d_raster_fwd(const float *verts_ndc, const int32_t *faces, int B, int V, int F, int S,
                    void *workspace, size_t workspace_bytes, int32_t *pix_to_face, float *zbuf,
                    float *bary, float *dists,
                    float z_clip, int32_t *near_flag /* device int, may be NULL: OR-ed with 1 when a rasterised face has a
                    vertex nearer than z_clip -- PyTorch3D would clip it (znear / 2); this K = 1 path does not, use the
                    general kernels (st3d_face_setup_clip) then */,
                    st3d_stream_t stream);

/* Fused TexturesUV.sample_textures + ambient shading + softmax_rgb_blend (K=1) + the
 * RGB/mask extraction of utils.py:70-72.  texture (T,T,3) HWC as maps_padded()[0]
 * (utils.py:208), verts_uvs (VT,2), faces_uvs (F,3) int32.
 * rgb (B,3,S,S), mask (B,1,S,S) = (alpha > 0). */
int st3d_shade_fwd(const int32_t *pix_to_face, const float *bary, const float *zbuf,
                   const float *dists, const float *verts_uvs, const int32_t *faces_uvs,
                   const float *texture, int B, int S, int T, int F, int VT, float *rgb,
                   float *mask, st3d_stream_t stream);

/* Texture-sampling backward (grid_sampler_2d_backward + blend backward):
 * grad_rgb (B,3,S,S) -> grad_texture (T,T,3), ACCUMULATED over the B views (caller zeroes).
 * Optional outputs for the vertex path (NULL for texture-only optimisation): grad_uv (B,S,S,2)
 * = d loss / d (u,v) and grad_bary (B,S,S,3) = d loss / d barycentrics (interpolate_face_attributes
 * backward).  grad_texture may be NULL when only the vertices are optimised. */
int st3d_shade_bwd(const float *grad_rgb, const int32_t *pix_to_face, const float *bary,
                   const float *zbuf, const float *dists, const float *verts_uvs,
                   const int32_t *faces_uvs, const float *texture, int B, int S, int T, int F,
                   int VT, float *grad_texture, float *grad_uv, float *grad_bary,
                   st3d_stream_t stream);

/* Rasteriser backward (PyTorch3D RasterizeMeshesBackward, grad_bary path; reached through
 * loss.backward() at second_approach.py:188 when optimization_target is 'mesh'/'both'):
 * grad_bary (B,S,S,3) -> grad_verts_ndc (B,V,3) (zeroed by the call, then atomically summed). */
int st3d_raster_bwd(const float *grad_bary, const int32_t *pix_to_face, const float *verts_ndc,
                    const int32_t *faces, int B, int V, int F, int S, float *grad_verts_ndc,
                    st3d_stream_t stream);
/* backward of st3d_project_verts: grad_verts (V,3) (+)= sum over the B views */
int st3d_project_verts_bwd(const float *verts, int V, const float *R, const float *T, int B,
                           float inv_tan_half_fov, const float *grad_verts_ndc, int accumulate,
                           float *grad_verts, st3d_stream_t stream);

/* ---- bitwise reproducible variants of the two gradient scatters ("deterministic variant").
 * The default kernels sum with float atomics, so the last bits of the texture / vertex gradient depend on the order
 * the hardware serves them; these accumulate in 64-bit fixed point with a power-of-two scale taken from a bound on the
 * partial sums (integer addition is associative: any order gives the same bits), then convert.  Same arguments and
 * results (to fp32 rounding of the final sums) as st3d_shade_bwd / st3d_raster_bwd, plus a caller-owned workspace. */
size_t st3d_shade_bwd_det_workspace_bytes(int T);
int st3d_shade_bwd_det(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                       const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                       int B, int S, int T, int F, int VT, float *grad_texture /* accumulated into */, float *grad_uv,
                       float *grad_bary, void *workspace, size_t workspace_bytes, st3d_stream_t stream);
size_t st3d_raster_bwd_det_workspace_bytes(int B, int V, int S);
int st3d_raster_bwd_det(const float *grad_bary, const int32_t *pix_to_face, const float *verts_ndc, const int32_t *faces,
                        int B, int V, int F, int S, float *grad_verts_ndc, void *workspace, size_t workspace_bytes,
                        st3d_stream_t stream);

/* ---- general soft renderer: PyTorch3D's MeshRasterizer / SoftPhongShader under any other
 * RasterizationSettings / BlendParams than the ones the reference constructs at first_approach.py:107-113 and
 * second_approach.py:101-108 (K = 1, blur_radius = 0, default blend, served by the entry points above):
 * K = faces_per_pixel <= 8 nearest faces per pixel, blur_radius >= 0, barycentric clipping (PyTorch3D clips when
 * blur_radius > 0), cull_backfaces, perspective_correct on/off, softmax_rgb_blend over the K layers with sigma / gamma / background (host float[3]).
 * Fragment arrays are (B,S,S,K[,3]), depth-sorted, -1 filled. */
int st3d_face_setup(const float *verts_ndc, const int32_t *faces, int B, int V, int F, void *face_records,
                    size_t records_bytes /* >= st3d_raster_workspace_bytes */, st3d_stream_t stream);
/* Near-plane clipping (PyTorch3D clips against z = z_clip_value = znear / 2 for perspective cameras before rasterising):
 * two record slots per face -- the face itself or its part in front of the plane as one or two triangles
 * (B * 2F records, st3d_clip_records_bytes); st3d_raster_soft_fwd then runs with records_per_face = 2, reports the ORIGINAL
 * face in pix_to_face, converts the barycentrics back to it and writes the record slot of every fragment to frag_slot
 * (B,S,S,K), which st3d_raster_soft_bwd needs to differentiate through the clip. */
size_t st3d_clip_records_bytes(int B, int F);
int st3d_face_setup_clip(const float *verts_ndc, const int32_t *faces, int B, int V, int F, float z_clip,
                         int perspective_correct, void *face_records, size_t records_bytes, st3d_stream_t stream);
int st3d_raster_soft_fwd(const float *face_records, int B, int F, int S, int K, float blur_radius, int clip_bary,
                         int cull_backfaces /* skip faces whose NDC area is negative */,
                         int perspective_correct /* 0: screen-space barycentrics */,
                         int records_per_face /* 1: st3d_face_setup records; 2: st3d_face_setup_clip records */,
                         int32_t *frag_slot /* (B,S,S,K) or NULL */,
                         int32_t *pix_to_face, float *zbuf, float *bary, float *dists, st3d_stream_t stream);
int st3d_shade_soft_fwd(const int32_t *pix_to_face, const float *bary, const float *zbuf, const float *dists,
                        const float *verts_uvs, const int32_t *faces_uvs, const float *texture, int B, int S, int T,
                        int K, float sigma, float gamma, const float *background, float *rgb, float *alpha,
                        st3d_stream_t stream);
/* grad_rgb (B,3,S,S) -> grad_texture (T,T,3) accumulated, per-layer grad_bary (B,S,S,K,3), grad_zbuf, grad_dists
 * (B,S,S,K); any output may be NULL */
int st3d_shade_soft_bwd(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                        const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                        int B, int S, int T, int K, float sigma, float gamma, const float *background,
                        float *grad_texture, float *grad_bary, float *grad_zbuf, float *grad_dists,
                        st3d_stream_t stream);
/* PyTorch3D RasterizeMeshesBackward: (grad_bary, grad_zbuf, grad_dists) -> grad_verts_ndc (B,V,3), zeroed by the call */
int st3d_raster_soft_bwd(const float *grad_bary, const float *grad_zbuf, const float *grad_dists,
                         const int32_t *pix_to_face, const float *verts_ndc, const int32_t *faces, int B, int V, int F,
                         int S, int K, int clip_bary, int perspective_correct,
                         const int32_t *frag_slot /* from the clipped forward, or NULL */, float z_clip,
                         float *grad_verts_ndc, st3d_stream_t stream);
/* The two scatters above with bitwise reproducible results: the same per-tile LDS binning (per texel / per face) in
 * 64-bit fixed point (csrc/det.h), like st3d_shade_bwd_det / st3d_raster_bwd_det for the specialised K = 1 path.
 * grad_texture must be given (it is accumulated into); workspace 16-byte aligned.  A non-finite input gradient gives a
 * NaN result (never a laundered finite one). */
size_t st3d_shade_soft_bwd_det_workspace_bytes(int T);
int st3d_shade_soft_bwd_det(const float *grad_rgb, const int32_t *pix_to_face, const float *bary, const float *zbuf,
                            const float *dists, const float *verts_uvs, const int32_t *faces_uvs, const float *texture,
                            int B, int S, int T, int K, float sigma, float gamma, const float *background,
                            float *grad_texture, float *grad_bary, float *grad_zbuf, float *grad_dists,
                            void *workspace, size_t workspace_bytes, st3d_stream_t stream);
size_t st3d_raster_soft_bwd_det_workspace_bytes(int B, int V, int S);
int st3d_raster_soft_bwd_det(const float *grad_bary, const float *grad_zbuf, const float *grad_dists,
                             const int32_t *pix_to_face, const float *verts_ndc, const int32_t *faces, int B, int V, int F,
                             int S, int K, int clip_bary, int perspective_correct, const int32_t *frag_slot, float z_clip,
                             float *grad_verts_ndc, void *workspace, size_t workspace_bytes, st3d_stream_t stream);

/* apply_background, utils.py:19-30: out = img*mask + bg*(1-mask); bg (B,3,S,S) or, with
 * bg_batch == 1, one (3,S,S) image broadcast over the batch.  Optional grad path is the
 * same kernel applied to the gradient with bg = NULL (out = g*mask). */
int st3d_apply_background(const float *img, const float *mask, const float *bg, int bg_batch,
                          int B, int S, float *out, st3d_stream_t stream);

/* ------------------------------------------------------------------ VGG-19 features:
 * utils.py:48-52 (get_vgg) + style_transfer.py:10-27 (get_features).  conv3x3 pad 1 +
 * bias + ReLU on fp32 MFMA (v_mfma_f32_32x32x2_f32), maxpool 2x2. */

/* packed-weight sizes in floats for a (Cout,Cin,3,3) filter */
size_t st3d_conv3x3_packed_floats(int Cout, int Cin);
/* w (Cout,Cin,3,3) -> w_fwd [9][Cin4][CoutP] and w_dgrad [9][Cout4][CinP] (rotated 180 deg,
 * channel-transposed); either output may be NULL. */
int st3d_conv3x3_pack(const float *w, int Cout, int Cin, float *w_fwd, float *w_dgrad,
                      st3d_stream_t stream);
/* y = relu?(conv3x3(x, w) + bias): x (N,Cin,H,W), y (N,Cout,H,W) */
int st3d_conv3x3_fwd(const float *x, const float *w_fwd_packed, const float *bias, float *y,
                     int N, int Cin, int Cout, int H, int W, int relu, st3d_stream_t stream);
/* gx = conv3x3^T(mask(gy)): gradient w.r.t. the conv INPUT (weights are frozen,
 * utils.py:50-51: no wgrad).  gy (N,Cout,H,W) is the gradient w.r.t. the POST-ReLU output;
 * act (same shape, the saved post-ReLU output) gates it (act>0) when non-NULL. */
int st3d_conv3x3_dgrad(const float *gy, const float *act, const float *w_dgrad_packed, float *gx,
                       int N, int Cin, int Cout, int H, int W, st3d_stream_t stream);
/* The bottom of the VGG backward fused (csrc/tap0.hip): gx (N,3,H,W) = conv1_1^T(gate(gy + coef * D act)) where act
 * (N,64,H,W) is the saved relu1_1 output (gate: act > 0), gy (N,64,H,W) the gradient arriving from conv1_2 (NULL = none)
 * and D (N,64,64) the style-loss difference Gram of the relu1_1 tap (NULL = no tap) -- what st3d_gram_bwd(accumulate)
 * followed by st3d_conv3x3_dgrad compute (autograd of losses.py:36-39 + second_approach.py:188), in one pass over gy and
 * act.  w_dgrad_packed: conv1_1's dgrad pack of st3d_conv3x3_pack.  workspace: st3d_conv1_bwd_workspace_bytes. */
int st3d_conv1_bwd_supported(int H, int W);
size_t st3d_conv1_bwd_workspace_bytes(int N, int H, int W);
int st3d_conv1_bwd(const float *gy, const float *act, const float *D, float coef, const float *w_dgrad_packed,
                   void *workspace, size_t workspace_bytes, float *gx, int N, int H, int W, st3d_stream_t stream);
/* As above, but gy is given at POOLED resolution (N,Cout,H/2,W/2) together with the pool's
 * argmax (uint8 0..3 = dy*2+dx) and pooled values: fuses max-unpool + ReLU gate into the load. */
int st3d_conv3x3_dgrad_unpool(const float *gy_pooled, const uint8_t *pool_idx, const float *pooled,
                              const float *w_dgrad_packed, float *gx, int N, int Cin, int Cout,
                              int H, int W, st3d_stream_t stream);
/* Winograd F(2x2,3x3) variants of the three conv entry points above (exact-fp32 MFMA products,
 * 2.25x fewer of them); for Cin % 8 == 0, Cout % 64 == 0, H even, W % 4 == 0 -- every VGG
 * layer but conv1_1.  u_fwd / u_dgrad = G g G^T in the kernel's MFMA-operand order (st3d_wino_pack).
 * st3d_wino_fwd can fuse the following MaxPool2d(2,2): y_pooled (N,Cout,H/2,W/2) + pool_idx
 * (either NULL = no pooling); y may then be NULL to skip the full-resolution store. */
int st3d_wino_supported(int Cin, int Cout, int H, int W);
size_t st3d_wino_packed_floats(int Cout, int Cin);
int st3d_wino_pack(const float *w, int Cout, int Cin, float *u_fwd, float *u_dgrad, st3d_stream_t stream);
int st3d_wino_fwd(const float *x, const float *u_fwd, const float *bias, float *y, float *y_pooled,
                  uint8_t *pool_idx, int N, int Cin, int Cout, int H, int W, int relu, st3d_stream_t stream);
int st3d_wino_dgrad(const float *gy, const float *act, const float *u_dgrad, float *gx, int N, int Cin,
                    int Cout, int H, int W, st3d_stream_t stream);
int st3d_wino_dgrad_unpool(const float *gy_pooled, const uint8_t *pool_idx, const float *pooled,
                           const float *u_dgrad, float *gx, int N, int Cin, int Cout, int H, int W,
                           st3d_stream_t stream);
/* One link of the backward chain with the ReLU gates moved to the PRODUCER of each gradient, so that the 8-64 stages of
 * the consumer's K loop stream one operand instead of two or three (measured 3-13 % per launch):
 *   input : gy, full resolution gated by act (NULL = gy is already gated), or -- pool_idx != NULL -- at pooled resolution,
 *           un-pooled through pool_idx and gated by pooled > 0 (pooled == NULL = already gated);
 *   output: gx = conv^T(...), zeroed where out_gate (N,Cin,H,W: the conv's own forward input, post-ReLU or pool output)
 *           is <= 0 when out_gate != NULL -- i.e. gx is handed on already gated for the next link.  With add_target
 *           (same shape, needs out_gate) the content-loss gradient add_coef * (out_gate - add_target) of that tensor
 *           (losses.py:24-28; what st3d_axpy_diff adds) joins gx before the gate, in the same store. */
int st3d_wino_dgrad_chain(const float *gy, const float *act, const uint8_t *pool_idx, const float *pooled,
                          const float *u_dgrad, const float *out_gate, const float *add_target, float add_coef,
                          float *gx, int N, int Cin, int Cout, int H, int W, st3d_stream_t stream);
/* The same convolutions as Winograd F(4x4,3x3) (csrc/wino43.hip, round 3): 2.25 instead of 4 MFMA-multiplies per output
 * pixel, fp32 throughout (<= 1.3e-5 of the output scale against an fp64 convolution at K = 512).  Shapes: Cin >= 64,
 * Cin % 16 == 0, Cout % 64 == 0 (both % 64 for the pack), and H % 4 == 0 with W % 64 == 0 or H % 8 == 0 with W % 32 == 0
 * (the workgroup's step is 4 x 64 or 8 x 32 pixels), each tensor < 2^31 bytes per image
 * (st3d_wino43_supported).  Own filter pack (36 floats per weight).  One persistent workgroup per CU.
 * st3d_wino43_dgrad_chain takes an already gated gradient (or, with pool_idx, the pooled-resolution gradient) exactly as
 * st3d_wino_dgrad_chain does with act == pooled == NULL. */
int st3d_wino43_supported(int Cin, int Cout, int H, int W);
size_t st3d_wino43_packed_floats(int Cout, int Cin);
int st3d_wino43_pack(const float *w, int Cout, int Cin, float *u_fwd, float *u_dgrad, st3d_stream_t stream);
int st3d_wino43_fwd(const float *x, const float *u_fwd, const float *bias, float *y, float *y_pooled, uint8_t *pool_idx,
                    int N, int Cin, int Cout, int H, int W, int relu, st3d_stream_t stream);
int st3d_wino43_dgrad_chain(const float *gy, const uint8_t *pool_idx, const float *u_dgrad, const float *out_gate,
                            const float *add_target, float add_coef, float *gx, int N, int Cin, int Cout, int H, int W,
                            st3d_stream_t stream);
/* MaxPool2d(2,2): y (N,C,H,W) -> p (N,C,H/2,W/2) (+ argmax idx, may be NULL) */
int st3d_maxpool2x2_fwd(const float *y, float *p, uint8_t *idx, int N, int C, int H, int W,
                        st3d_stream_t stream);

/* ------------------------------------------------------------------ Gram / losses:
 * style_transfer.py:31-35 (gram_matrix), losses.py:31-42 */

size_t st3d_gram_workspace_bytes(int B, int C, int HW);
/* gram (B,C,C) = F F^T, F = feat (B,C,HW); unnormalised; split-K fp32 MFMA + ordered
 * (deterministic) slab reduction. */
int st3d_gram_fwd(const float *feat, int B, int C, int HW, void *workspace, size_t workspace_bytes,
                  float *gram, st3d_stream_t stream);
/* The Grams of ALL style layers of a step (losses.py:34-37 loops over them; style_transfer.py:47-49,
 * 66-69 likewise) in one launch pair: items is a HOST array (<= 8), every item is one
 * st3d_gram_fwd problem; workspace >= st3d_gram_multi_workspace_bytes(items, count), 256-byte
 * aligned (each layer has its own slab region: they run concurrently).  Bitwise the results of
 * st3d_gram_fwd per item.  ST3D_GRAM_MULTI=0 runs the items one by one (A/B). */
typedef struct st3d_gram_item {
    const float *feat;   /* (B, C, HW) */
    float *gram;         /* (B, C, C) */
    int B, C, HW;
} st3d_gram_item;
size_t st3d_gram_multi_workspace_bytes(const st3d_gram_item *items, int count);
int st3d_gram_fwd_multi(const st3d_gram_item *items, int count, void *workspace, size_t workspace_bytes,
                        st3d_stream_t stream);
/* gfeat (B,C,HW) (+)= coef * (D F) with D (B,C,C) symmetric (D = G - S); accumulate != 0 adds
 * into gfeat. */
int st3d_gram_bwd(const float *D, const float *feat, int B, int C, int HW, float coef,
                  int accumulate, float *gfeat, st3d_stream_t stream);
/* the same, then gfeat = 0 where feat <= 0 (feat is post-ReLU: its ReLU gate, taken from the operand tile already in
 * LDS), so the gradient leaves already gated (see st3d_wino_dgrad_chain).  C % 32 == 0. */
int st3d_gram_bwd_gated(const float *D, const float *feat, int B, int C, int HW, float coef,
                        int accumulate, float *gfeat, st3d_stream_t stream);
/* loss_out[0] += scale * sum((a-b)^2) over n elements (b broadcast with period nb, nb | n);
 * if D != NULL also D = a - b.  Deterministic two-stage reduction through `partials`
 * (>= st3d_reduce_partials() floats). */
int st3d_reduce_partials(void);
int st3d_sqdiff_sum(const float *a, const float *b, size_t n, size_t nb, float scale, float *D,
                    float *partials, float *loss_out, st3d_stream_t stream);
/* Up to 8 squared-difference sums in one launch pair (the plan's loss tail): item k adds scale * sum((a - b)^2) over n
 * elements (b with period nb; D = a - b when non-NULL) into loss_out3[slot], slot in {0, 1, 2}; zero_first clears the
 * three slots first; combine != 0 then sets loss_out3[0] = content_weight * loss_out3[1] + style_weight * loss_out3[2]
 * (losses.py:41-44).  Every item keeps the decomposition and the summation tree of st3d_sqdiff_sum: same bits.
 * partials: count * st3d_reduce_partials() floats. */
typedef struct st3d_sqdiff_item {
    const float *a; const float *b; float *D;
    size_t n, nb;
    float scale;
    int slot;
} st3d_sqdiff_item;
int st3d_sqdiff_sum_multi(const st3d_sqdiff_item *items, int count, float *partials, float *loss_out3,
                          int zero_first, int combine, float style_weight, float content_weight,
                          st3d_stream_t stream);
/* content loss backward: g (+)= coef * (a - b)  */
int st3d_axpy_diff(const float *a, const float *b, size_t n, float coef, int accumulate, float *g,
                   st3d_stream_t stream);
/* the same, then g = 0 where a <= 0: a is the post-ReLU activation the gradient g belongs to, so g leaves already gated
 * (see st3d_wino_dgrad_chain) */
int st3d_axpy_diff_gated(const float *a, const float *b, size_t n, float coef, int accumulate, float *g,
                         st3d_stream_t stream);
/* masked MSE of losses.py:68-75 ('texture' branch): loss_out[0] = mean((r*m - t*m)^2) over
 * B*3*S*S; grad_r = 2*m*m*(r - t)/(B*3*S*S) (may be NULL). */
int st3d_masked_mse(const float *rendered, const float *target, const float *mask, int B, int S,
                    float *grad_rendered, float *partials, float *loss_out, st3d_stream_t stream);
/* masked anisotropic L1 total variation of losses.py:55-65 (compute_tv_loss; every call site in the reference is
 * commented out): images (B,C,H,W), masks (B,1,H,W); loss_and_mask_sum[0] = (sum |dI/dy| m m' + sum |dI/dx| m m') /
 * sum(masks), [1] = sum(masks); grad_images (same shape, may be NULL) = d loss / d images.  partials must hold
 * 2 * st3d_reduce_partials() floats. */
int st3d_tv_loss(const float *images, const float *masks, int B, int C, int H, int W, float *partials,
                 float *loss_and_mask_sum, float *grad_images, st3d_stream_t stream);
/* losses.py:48-51 (rgb_range_loss): loss_out[0] = sum relu(v - 1) + relu(-v); grad (may be NULL) = +1 / -1 / 0 */
int st3d_range_loss(const float *values, size_t n, float *partials, float *loss_out, float *grad, st3d_stream_t stream);

/* ------------------------------------------------------------------ mesh regularisers:
 * losses.py:84-87,93-96,112-115,121-124 -- F.mse_loss(verts, target), pytorch3d.loss
 * mesh_edge_loss (target length 0), mesh_laplacian_smoothing('uniform'), mesh_normal_consistency
 * for one mesh, forward + gradient in one call.  Topology is static and precomputed by the host:
 * edges (E,2) unique undirected; CSR vertex adjacency nbr_off (V+1), nbr_idx; pairs (P,4) =
 * (v0, v1, a, b) for every two faces sharing edge (v0,v1) with opposite vertices a, b; and the
 * inverse of `pairs`: pair_off (V+1), pair_ref = for each vertex the entries (pair * 4 + column)
 * of `pairs` that name it, ascending.  Every gradient is a per-vertex gather over these static
 * lists in a fixed order -- no float atomics, bitwise reproducible (round 3).
 * weights: host float[4] = {verts_mse, edge, laplacian, normal}.  scratch >=
 * st3d_mesh_reg_scratch_floats(V, P) floats, partials >= 4*st3d_reduce_partials() floats.
 * loss_out: device float[5] = {weighted sum, mse, edge, laplacian, normal} (unweighted terms).
 * grad_verts (V,3) += weighted gradient. */
size_t st3d_mesh_reg_scratch_floats(int V, int P);
int st3d_mesh_reg(const float *verts, const float *target_verts, int V, const int32_t *edges, int E,
                  const int32_t *nbr_off, const int32_t *nbr_idx, const int32_t *pairs, int P,
                  const int32_t *pair_off, const int32_t *pair_ref,
                  const float *weights, float *scratch, float *partials, float *loss_out,
                  float *grad_verts, st3d_stream_t stream);

/* ------------------------------------------------------------------ optimiser:
 * torch.optim.Adam defaults (utils.py:185-195, style_transfer.py:57) */
int st3d_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                   int step, float lr, float beta1, float beta2, float eps, st3d_stream_t stream);

/* ------------------------------------------------------------------ fused perceptual engine:
 * losses.py:12-44 (compute_perceptual_loss) and the loop body of style_transfer.py:59-83.
 * One object per (VGG weights); one plan per (batch, image size). */
typedef struct st3d_vgg st3d_vgg;
typedef struct st3d_plan st3d_plan;

int st3d_vgg_create(st3d_vgg **out);
/* module_idx in torchvision's vgg19().features numbering (0,2,5,7,...,34); w (Cout,Cin,3,3), b (Cout) */
int st3d_vgg_set_conv(st3d_vgg *vgg, int module_idx, const float *w, const float *b,
                      st3d_stream_t stream);
int st3d_vgg_destroy(st3d_vgg *vgg);

/* workspace for batches of up to B images of S x S (any S >= 16; sizes off the fast path -- odd intermediate
 * resolutions, W % 4 != 0 -- run on the direct conv / separate pool kernels, pools floor like MaxPool2d) */
int st3d_plan_create(st3d_plan **out, st3d_vgg *vgg, int B, int S);
int st3d_plan_destroy(st3d_plan *plan);
size_t st3d_plan_bytes(const st3d_plan *plan);
/* forward of imgs (n,3,S,S), n <= B, through modules 0..upto_module (post-ReLU taps) */
int st3d_plan_forward(st3d_plan *plan, const float *imgs, int n, int upto_module, st3d_stream_t stream);
/* device pointer + shape of the activation after module_idx (a conv index = its post-ReLU
 * output, a pool index = the pooled output) of the last st3d_plan_forward */
int st3d_plan_activation(st3d_plan *plan, int module_idx, float **ptr, int *C, int *H, int *W);
/* targets (losses.py:18-25): conv4_2 features of content (B,3,S,S); Grams of style
 * (style_batch == 1: one image broadcast over the batch, as second_approach.py:157 repeats it) */
int st3d_plan_set_content(st3d_plan *plan, const float *content, int n, st3d_stream_t stream);
/* the content target (conv4_2 features, n x 512 x S/8 x S/8 floats) out of / into the plan: a caller alternating between
 * several view batches (second_approach.py:145-160 with n_views > batch_size) can keep each batch's target instead of
 * recomputing it every step */
int st3d_plan_get_content_features(st3d_plan *plan, float *out, int n, st3d_stream_t stream);
int st3d_plan_set_content_features(st3d_plan *plan, const float *features, int n, st3d_stream_t stream);
int st3d_plan_set_style(st3d_plan *plan, const float *style, int style_batch, int n, st3d_stream_t stream);
/* loss (losses.py:28-42) of current (n,3,S,S) and, if grad_current != NULL, d loss/d current.
 * batch_denom = the batch size the means divide by (n, or the GLOBAL batch when views are
 * sharded over ranks).  loss_out: device float[3] = {total, content_loss, style_loss}. */
int st3d_plan_loss(st3d_plan *plan, const float *current, int n, int batch_denom, float style_weight,
                   float content_weight, float *loss_out, float *grad_current, st3d_stream_t stream);
/* HIP-graph replay of st3d_plan_loss: its ~70 launches form a static sequence, so after one ordinary call it is captured
 * (per n / batch_denom / weights) and replayed with one hipGraphLaunch; inputs and outputs pass through plan-owned staging
 * buffers (three extra device copies per call).  Pays off where the step is launch-bound (small images). */
int st3d_plan_graph(st3d_plan *plan, int enable);
/* Backward of st3d_plan_forward for losses computed OUTSIDE the library on its taps (the reference's own loop body,
 * style_transfer.py:61-83, calls get_features(optimized_imgs) with grad and back-propagates through it):
 * grad_modules = host array of 37 device pointers, entry m = d loss / d (output of VGG module m) shaped like
 * st3d_plan_activation(m) for the n images of the last forward (NULL = none; a conv and the in-place ReLU behind it
 * are one tensor) -> grad_image (n,3,S,S). */
int st3d_plan_backward(st3d_plan *plan, int n, int upto_module, const float *const *grad_modules, float *grad_image,
                       st3d_stream_t stream);
/* per-kernel-family timing of the next calls (HIP events on the call's stream): enable, then
 * read accumulated milliseconds + launch counts; families: 0 conv_fwd (Winograd launches) 1 conv_dgrad (Winograd)
 * 2 pool 3 gram_fwd 4 gram_bwd 5 loss/elementwise 6 convx_fwd (convs Winograd does not cover: conv1_1, odd shapes)
 * 7 convx_dgrad */
#define ST3D_PROFILE_FAMILIES 10
int st3d_plan_profile(st3d_plan *plan, int enable);
int st3d_plan_profile_read(st3d_plan *plan, float *ms_out /*host [ST3D_PROFILE_FAMILIES]*/,
                           int *launches_out /*host [ST3D_PROFILE_FAMILIES]*/);
/* Per-launch records gathered while profiling was on, since the previous call: tag = family * 100 + the VGG module
 * index the launch belongs to (99 = none), HIP-event milliseconds.  *count_out = records available; they are consumed
 * when `capacity` holds them all. */
int st3d_plan_profile_launches(st3d_plan *plan, int *tags_out /*host [capacity]*/, float *ms_out /*host [capacity]*/,
                               int capacity, int *count_out);

/* ------------------------------------------------------------------ multi-GPU
 * One process per GPU; every rank renders / VGGs its slice of the view batch with the loss means divided by the GLOBAL
 * batch (batch_denom of st3d_plan_loss) and ONE SUM all-reduce of the flat fp32 gradient (3*T*T texture floats
 * [+ 3*V vertex floats]) makes the gradients identical everywhere before the replicated st3d_adam_step.  The reference
 * has no multi-GPU path (no torch.distributed / NCCL call site anywhere); the Python host of this package goes through
 * torch.distributed (backend "nccl" = RCCL on ROCm) -- these entry points give a C caller the same collective: RCCL's
 * ncclAllReduce over xGMI, bound at run time (no link-time dependency).  Rank 0 creates the id and hands the 128 bytes to
 * the other ranks by any out-of-band means (file, env, socket); each rank then calls st3d_comm_init with its HIP
 * device current. */
#define ST3D_COMM_ID_BYTES 128
typedef struct st3d_comm st3d_comm;
int st3d_comm_unique_id(unsigned char id_out[ST3D_COMM_ID_BYTES]);
int st3d_comm_init(st3d_comm **out, int rank, int world, const unsigned char unique_id[ST3D_COMM_ID_BYTES]);
int st3d_allreduce_sum_f32(st3d_comm *comm, float *buf /* device, in place */, size_t n, st3d_stream_t stream);
int st3d_comm_destroy(st3d_comm *comm);

/* Named ranges for rocprofv3 --marker-trace (the reference's only progress reporting are tqdm bars,
 * style_transfer.py:59, first_approach.py:191, second_approach.py:145).  With ST3D_ROCTX=1 in the environment the library
 * binds libroctx64 at run time and st3d_plan_loss / st3d_plan_forward / st3d_plan_backward mark their phases (vgg_forward,
 * gram_and_losses, vgg_backward); hosts bracket their own phases with push / pop (the Python host: render, render_backward,
 * allreduce, adam).  Without it every call returns at once.  st3d_trace_enabled: 1 when ranges are being emitted. */
int st3d_trace_push(const char *name);
int st3d_trace_pop(void);
int st3d_trace_enabled(void);

#ifdef __cplusplus
}
#endif
#endif /* ST3D_H */
