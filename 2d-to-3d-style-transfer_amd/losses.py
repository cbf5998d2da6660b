"""Drop-in for the reference's ``losses.py`` (same five functions, argument order, defaults and
error behaviour: reference losses.py:12,48,55,68,101) on libst3d.

``compute_perceptual_loss`` returns a scalar tensor whose ``.backward()`` (called by the caller,
second_approach.py:188) delivers d loss / d current_imgs from the fused plan: the VGG forward,
Gram/content losses and the whole backward run inside ONE autograd node (the gradient is
produced together with the loss, so ``backward()`` only scales and hands it on).  Content
features and style Grams (reference :18-25) do not depend on the optimised parameters and are
cached on the tensors' identity/version, i.e. recomputed exactly when the caller passes new or
modified content/style images.
"""
import torch
from torch.nn import functional as F  # noqa: F401  (star-import surface of the reference module)

from st3d import mesh_losses as _mesh_losses
from st3d import ops as _ops
from st3d import vgg as _vgg
from st3d.mesh_losses import mesh_edge_loss, mesh_laplacian_smoothing, mesh_normal_consistency  # noqa: F401
from style_transfer import *  # noqa: F401,F403  (the reference does the same, losses.py:5)

# Check if CUDA is available
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _PerceptualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, current, plan, style_weight, content_weight, batch_denom):
        loss, grad = plan.loss(current, style_weight, content_weight, batch_denom=batch_denom,
                               want_grad=current.requires_grad)
        ctx.grad = grad
        ctx.parts = loss.clone()
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.grad * grad_out if ctx.grad is not None else None
        return g, None, None, None, None


#method for the second approach
def compute_perceptual_loss(current_imgs, content_imgs, style_imgs, model, style_weight=1e6, content_weight=1, *,
                            batch_denom=None):

    # Ensure content_imgs and style_imgs are batched tensors
    assert current_imgs.shape[0] == content_imgs.shape[0] == style_imgs.shape[0]
    if not isinstance(model, _vgg.Vgg19Features):
        raise TypeError("compute_perceptual_loss needs the st3d VGG returned by utils.get_vgg()")
    if not current_imgs.is_cuda:
        raise RuntimeError("st3d runs on the GPU (libst3d); got CPU tensors -- there is no CPU fallback")

    B, S = current_imgs.shape[0], current_imgs.shape[2]
    plan = model.plan(B, S)
    plan.set_content(content_imgs)          # conv4_2 of content      (reference :18)
    plan.set_style(style_imgs, B)           # Grams of style features (reference :19-25)

    # batch_denom: the batch the means divide by -- the GLOBAL batch when views are sharded over ranks
    return _PerceptualFn.apply(current_imgs, plan, float(style_weight), float(content_weight), batch_denom)


class _FusedLossFn(torch.autograd.Function):
    """A loss whose HIP call returns the value and its gradient together: backward only scales."""

    @staticmethod
    def forward(ctx, x, op, *extra):
        loss, grad = op(x.detach(), *extra, want_grad=x.requires_grad)
        ctx.grad, ctx.n_extra = grad, len(extra)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.grad * grad_out if ctx.grad is not None else None
        return (g, None) + (None,) * ctx.n_extra


def _on_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("st3d runs on the GPU (libst3d); got CPU tensors -- there is no CPU fallback")
    return t.to(torch.float32)


def rgb_range_loss(mesh):
    """Sum of the texture map's excursions outside [0,1] (reference losses.py:48-51; every call site in the
    reference is commented out).  One fused launch: value + sign gradient."""
    tex = _on_gpu(mesh.textures.maps_padded())
    return _FusedLossFn.apply(tex, _ops.range_loss)


def compute_tv_loss(images, masks):
    """Masked anisotropic L1 total variation / sum(masks) (reference losses.py:55-65; call sites commented out
    there too).  Value and d/d images from one fused pass."""
    return _FusedLossFn.apply(_on_gpu(images), _ops.tv_loss, _on_gpu(masks).detach())


def texture_l2_loss(mesh, original_map):
    """mean((texture - original)^2): the "l2 regularization w.r.t. the original texture" idea of the reference's
    notes.txt:39 (not implemented there)."""
    tex = _on_gpu(mesh.textures.maps_padded())
    return _FusedLossFn.apply(tex, _l2_to, _on_gpu(original_map).detach())


def _l2_to(x, ref, want_grad=True):
    if not want_grad:
        return _ops.sqdiff_sum(x, ref.reshape(x.shape), scale=1.0 / x.numel()), None
    loss, diff = _ops.sqdiff_sum(x, ref.reshape(x.shape), scale=1.0 / x.numel(), want_diff=True)
    return loss, diff * (2.0 / x.numel())


class _MaskedMseFn(torch.autograd.Function):
    """F.mse_loss(rendered*masks, target*masks) (reference :71-75) as one fused reduction."""

    @staticmethod
    def forward(ctx, rendered, masks, target):
        loss, grad = _ops.masked_mse(rendered.detach(), target.detach(), masks.detach(), want_grad=rendered.requires_grad)
        ctx.grad = grad
        return loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        return (ctx.grad * grad_out if ctx.grad is not None else None), None, None


def _masked_mse(rendered, masks, target_rendered, batch_denom=None):
    if not rendered.is_cuda:
        raise RuntimeError("st3d runs on the GPU (libst3d); got CPU tensors -- there is no CPU fallback")
    loss = _MaskedMseFn.apply(rendered, masks, target_rendered)
    if batch_denom is not None and batch_denom != rendered.shape[0]:
        # views sharded over ranks: the mean over this rank's views becomes its share of the mean over the whole batch
        loss = loss * (rendered.shape[0] / float(batch_denom))
    return loss


def _mesh_terms(verts, target_verts, mesh, weights):
    """The four view-independent regularisers both approaches add for 'mesh'/'both'
    (reference losses.py:84-87,93-96,112-115,121-124), same weights-dict keys."""
    return _mesh_losses.mesh_terms(verts, target_verts, mesh, weights)      # one fused forward+gradient call


def compute_first_approach_loss(rendered, masks, target_rendered, verts, target_verts, mesh, weights, opt_type, *,
                                batch_denom=None):
    # 'texture' ignores main_loss_weight (reference :75); an unknown opt_type leaves `loss` unbound and
    # raises UnboundLocalError at the return, as the reference does (:98).
    # batch_denom (views sharded over ranks): only the IMAGE term is a mean over views and is scaled to this rank's
    # share of the global batch; the mesh terms are view-independent and already enter with 1/world per rank.
    if opt_type == 'texture':
        loss = _masked_mse(rendered, masks, target_rendered, batch_denom)
    elif opt_type in ('mesh', 'both'):
        loss = weights['main_loss_weight'] * _masked_mse(rendered, masks, target_rendered, batch_denom)
        loss = loss + _mesh_terms(verts, target_verts, mesh, weights)
    return loss


def compute_second_approach_loss(current, content, style, model, style_weight, content_weight, verts, target_verts, mesh,
                                 weights, opt_type, *, batch_denom=None):
    if opt_type in ('texture', 'mesh', 'both'):
        perceptual = compute_perceptual_loss(current, content, style, model, style_weight=style_weight,
                                             content_weight=content_weight, batch_denom=batch_denom)
    if opt_type == 'texture':
        loss = perceptual                                   # no main_loss_weight here (reference :103-104)
    elif opt_type in ('mesh', 'both'):
        loss = weights['main_loss_weight'] * perceptual + _mesh_terms(verts, target_verts, mesh, weights)
    return loss
