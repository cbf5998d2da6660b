"""Drop-in for the reference's ``style_transfer.py`` (same names, signatures, defaults and
return values: reference style_transfer.py:10,31,38) running on libst3d (MI355X / gfx950).

  get_features   one fused VGG forward through the plan (13 MFMA conv launches) instead of
                 37 module calls; taps are the POST-ReLU activations, exactly what the
                 reference's in-place ReLU leaves in its feature dict
  gram_matrix    split-K fp32-MFMA Gram with an autograd backward
  style_transfer targets once, then per step ONE call into the fused forward/loss/backward
                 plan and ONE fused Adam launch on the pixels (reference :59-83 builds and
                 tears down an autograd graph per step)
"""
import torch
import torch.optim as optim  # noqa: F401  (star-import surface of the reference module)
from tqdm import tqdm

from st3d import ops as _ops
from st3d import optim as _st3d_optim
from st3d import vgg as _vgg

# Check if CUDA is available
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

_DEFAULT_LAYERS = {
    '0': 'conv1_1',
    '5': 'conv2_1',
    '10': 'conv3_1',
    '19': 'conv4_1',
    '21': 'conv4_2',  # Content layer
    '28': 'conv5_1'
}


def _fused_ok(image, model):
    return (isinstance(model, _vgg.Vgg19Features) and image.is_cuda and image.dim() == 4 and image.shape[1] == 3
            and image.shape[2] == image.shape[3] and image.shape[2] >= 16)


class _FeaturesFn(torch.autograd.Function):
    """Differentiable taps: forward = the fused plan forward, backward = the plan's input-gradient chain seeded with the
    gradients the caller's loss sends into the taps (st3d_plan_backward).  The plan keeps the activations the backward
    needs; if another forward has used its buffers since (``plan.generation``), the forward is redone first."""

    @staticmethod
    def forward(ctx, image, model, modules):
        plan = model.plan(image.shape[0], image.shape[2])
        upto = max(modules)
        plan.forward(image, upto=upto)
        ctx.plan, ctx.modules, ctx.upto, ctx.generation = plan, modules, upto, plan.generation
        ctx.save_for_backward(image)
        return tuple(plan.activation(m).clone() for m in modules)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        (image,) = ctx.saved_tensors
        if plan.generation != ctx.generation:
            plan.forward(image, upto=ctx.upto)
        given = {m: g for m, g in zip(ctx.modules, grads) if g is not None}
        return plan.backward(given, ctx.upto), None, None


# Extract features using VGG19
def get_features(image, model, layers=None):
    if layers is None:
        layers = _DEFAULT_LAYERS
    if _fused_ok(image, model):
        want = {int(k): v for k, v in layers.items() if k in model._modules}
        features = {}
        if want:
            modules = tuple(int(name) for name in model._modules if int(name) in want)     # module order, as the reference fills it
            if image.requires_grad and torch.is_grad_enabled():
                # the reference's own loop body (style_transfer.py:61-83) back-propagates through these
                outs = _FeaturesFn.apply(image, model, modules)
            else:
                plan = model.plan(image.shape[0], image.shape[2])
                plan.forward(image, upto=max(want))
                outs = tuple(plan.activation(m).clone() for m in modules)
            for m, t in zip(modules, outs):
                features[want[m]] = t
        return features
    if isinstance(model, _vgg.Vgg19Features) and image.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("differentiable get_features on the st3d VGG needs square (B,3,S,S) GPU images, S >= 16")
    # any other model: the reference's generic walk (each module is whatever the caller built)
    features = {}
    x = image
    for name, layer in model._modules.items():
        x = layer(x)
        if name in layers:
            features[layers[name]] = x
    return features


class _GramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor):
        t = tensor.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(t)
        return _ops.gram_fwd(t)

    @staticmethod
    def backward(ctx, grad_gram):
        (t,) = ctx.saved_tensors
        D = (grad_gram + grad_gram.transpose(1, 2)).contiguous()       # dF = (dG + dG^T) F
        return _ops.gram_bwd(D, t, 1.0)


# Calculate Gram matrix for style representation
def gram_matrix(tensor):
    batch_size, d, h, w = tensor.size()
    if not tensor.is_cuda:
        raise RuntimeError("st3d gram_matrix runs on the GPU (libst3d); got a CPU tensor -- there is no CPU fallback")
    return _GramFn.apply(tensor)


def style_transfer(initial_optimized_imgs, content_imgs, style_imgs, model, steps=2000, style_weight=1e6, content_weight=1, lr=0.003):

    # Ensure content_imgs and style_imgs are batched tensors
    assert initial_optimized_imgs.shape[0] == content_imgs.shape[0] == style_imgs.shape[0]
    if not isinstance(model, _vgg.Vgg19Features):
        raise TypeError("style_transfer needs the st3d VGG returned by utils.get_vgg()")

    B, S = initial_optimized_imgs.shape[0], initial_optimized_imgs.shape[2]
    plan = model.plan(B, S)

    # content conv4_2 features and style Grams: computed once (reference :44-51)
    plan.set_content(content_imgs.to(device), force=True)
    plan.set_style(style_imgs.to(device), B, force=True)

    # Initialize target images  --> the ones to optimize
    optimized_imgs = initial_optimized_imgs.clone().detach().to(device).contiguous().requires_grad_(True)

    # Define optimizer (fused HIP Adam; pixels are not sharded, so no gradient all-reduce)
    optimizer = _st3d_optim.Adam([optimized_imgs], lr=lr, reduce_grads=False)

    for step in tqdm(range(steps), desc="2D Style Transfer"):
        _, grad = plan.loss(optimized_imgs, style_weight, content_weight)
        optimized_imgs.grad = grad
        optimizer.step()

    return optimized_imgs
