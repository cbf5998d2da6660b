"""oracle/perceptual_ref.py -- TEST INFRASTRUCTURE ONLY.

torch-CPU fp32 restatement of the VGG / Gram / loss half of the reference hot path:
``style_transfer.py:10-85`` (get_features, gram_matrix, style_transfer) and
``losses.py:12-98`` (compute_perceptual_loss, rgb_range_loss, compute_tv_loss,
compute_first_approach_loss 'texture' branch).  Pinned against the reference's own code:
``tests/golden/make_golden.py`` imports ``/root/reference/style_transfer.py`` and
``losses.py`` (the latter with the three absent ``pytorch3d.loss`` names stubbed) and
stores their outputs on seeded inputs in ``tests/golden/*.npz``;
``tests/test_oracle_perceptual.py`` checks this file against those vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

torchvision (unpinned, absent) supplies the VGG-19 layout in the reference (utils.py:49);
its pretrained weights need a download and are unavailable offline.  ``make_vgg19_features``
re-creates the torchvision ``vgg19().features`` layout with seeded weights.
"""
import math

import torch
import torch.nn as nn

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
TAPS = {"0": "conv1_1", "5": "conv2_1", "10": "conv3_1", "19": "conv4_1", "21": "conv4_2", "28": "conv5_1"}


def synthetic_vgg19_state(seed=0, bias_scale=0.05):
    """Seeded stand-in for the IMAGENET1K_V1 weights: He-normal weights, small normal biases,
    drawn in layer order from one CPU generator.  Keys follow torchvision: '<idx>.weight'."""
    g = torch.Generator().manual_seed(seed)
    state = {}
    cin, idx = 3, 0
    for v in VGG19_CFG:
        if v == "M":
            idx += 1
            continue
        state[f"{idx}.weight"] = torch.randn((v, cin, 3, 3), generator=g) * math.sqrt(2.0 / (cin * 9))
        state[f"{idx}.bias"] = torch.randn((v,), generator=g) * bias_scale
        cin = v
        idx += 2
    return state


def make_vgg19_features(state=None, seed=0):
    """37-module nn.Sequential with torchvision's layout: conv3x3(pad 1) + ReLU(inplace=True),
    MaxPool2d(2,2); parameters frozen (utils.py:50-51)."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    model = nn.Sequential(*layers)
    model.load_state_dict(state if state is not None else synthetic_vgg19_state(seed))
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def get_features_ref(image, model, upto=28):
    """style_transfer.py:10-27.  The tap tensors are the objects the following in-place ReLU
    overwrites, i.e. the POST-ReLU activations -- restated explicitly here,
    and stopping after module `upto` (the reference runs the unused tail 29-36 too)."""
    feats = {}
    x = image
    mods = list(model._modules.items())
    i = 0
    while i < len(mods):
        name, layer = mods[i]
        if isinstance(layer, nn.Conv2d):
            x = torch.relu(torch.nn.functional.conv2d(x, layer.weight, layer.bias, padding=1))
            if name in TAPS:
                feats[TAPS[name]] = x
            if int(name) >= upto:
                break
            i += 2          # the ReLU has been applied
        else:
            x = torch.nn.functional.max_pool2d(x, 2, 2)
            i += 1
    return feats


def gram_ref(t):
    """style_transfer.py:31-35 (unnormalised)."""
    b, c, h, w = t.shape
    f = t.reshape(b, c, h * w)
    return torch.bmm(f, f.transpose(1, 2))


def perceptual_loss_ref(current, content, style, model, style_weight=1e6, content_weight=1.0, return_parts=False):
    """losses.py:12-44."""
    assert current.shape[0] == content.shape[0] == style.shape[0]
    content_f = get_features_ref(content, model, upto=21)["conv4_2"]
    style_f = get_features_ref(style, model)
    style_grams = {k: gram_ref(v) for k, v in style_f.items() if k != "conv4_2"}
    cur = get_features_ref(current, model)
    content_loss = torch.mean((cur["conv4_2"] - content_f) ** 2)
    style_loss = 0
    for k, sg in style_grams.items():
        f = cur[k]
        g = gram_ref(f)
        style_loss = style_loss + torch.mean((g - sg) ** 2) / (f.shape[1] ** 2 * f.shape[2] ** 2)
    total = content_weight * content_loss + style_weight * style_loss
    if return_parts:
        return total, content_loss, style_loss
    return total


def first_approach_loss_texture_ref(rendered, masks, target):
    """losses.py:68-75 ('texture' branch): MSE over all B*3*S*S elements of the masked images."""
    return torch.mean((rendered * masks - target * masks) ** 2)


def tv_loss_ref(images, masks):
    """losses.py:55-65."""
    dh = images[..., :-1, :] - images[..., 1:, :]
    dw = images[..., :, :-1] - images[..., :, 1:]
    mh = masks[..., :-1, :] * masks[..., 1:, :]
    mw = masks[..., :, :-1] * masks[..., :, 1:]
    return (torch.sum(dh.abs() * mh) + torch.sum(dw.abs() * mw)) / torch.sum(masks)


def rgb_range_loss_ref(texture):
    """losses.py:48-51 on the texture map tensor."""
    return torch.sum(torch.relu(texture - 1) + torch.relu(-texture))


def style_transfer_ref(init, content, style, model, steps, style_weight=1e6, content_weight=1.0, lr=0.003):
    """style_transfer.py:38-85 (targets once, Adam on the pixels); returns the unclamped
    leaf and the per-step losses."""
    with torch.no_grad():
        content_f = get_features_ref(content, model, upto=21)["conv4_2"]
        style_f = get_features_ref(style, model)
        style_grams = {k: gram_ref(v) for k, v in style_f.items() if k != "conv4_2"}
    x = init.clone().detach().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=lr)
    losses = []
    for _ in range(steps):
        cur = get_features_ref(x, model)
        content_loss = torch.mean((cur["conv4_2"] - content_f) ** 2)
        style_loss = 0
        for k, sg in style_grams.items():
            f = cur[k]
            style_loss = style_loss + torch.mean((gram_ref(f) - sg) ** 2) / (f.shape[1] ** 2 * f.shape[2] ** 2)
        total = content_weight * content_loss + style_weight * style_loss
        opt.zero_grad()
        total.backward()
        opt.step()
        losses.append(float(total))
    return x.detach(), losses
