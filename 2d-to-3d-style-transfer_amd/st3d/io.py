"""Host-side asset I/O: Wavefront OBJ/MTL + texture image load, OBJ/MTL/PNG save.

Replaces what the reference gets from ``pytorch3d.io.load_obj`` / ``IO().save_mesh``
(first_approach.py:83-88,225; second_approach.py:77-82,202) -- semantics per SURVEY.md A.5:
1-based (or negative = relative) indices, ``f`` tokens ``v``, ``v/vt``, ``v//vn``,
``v/vt/vn``, polygons fan-triangulated ``(0, i+1, i+2)``, texture images from ``map_Kd``
as float32 HWC in [0,1].  One-time setup work; stays on the host.
"""
import os
from collections import namedtuple

import numpy as np
import torch

Faces = namedtuple("Faces", "verts_idx normals_idx textures_idx materials_idx")
Properties = namedtuple("Properties", "normals verts_uvs material_colors texture_images texture_atlas")


def _parse_mtl(path):
    """-> (material_colors: {name: {...}}, texture_files: {name: filename})."""
    colors, tex = {}, {}
    cur = None
    if not os.path.isfile(path):
        return colors, tex
    with open(path, "r") as f:
        for line in f:
            tok = line.split("#", 1)[0].split()
            if not tok:
                continue
            if tok[0] == "newmtl":
                cur = tok[1]
                colors[cur] = {}
            elif cur is None:
                continue
            elif tok[0] == "map_Kd":
                tex[cur] = " ".join(tok[1:])
            elif tok[0] in ("Ka", "Kd", "Ks"):
                key = {"Ka": "ambient_color", "Kd": "diffuse_color", "Ks": "specular_color"}[tok[0]]
                colors[cur][key] = torch.tensor([float(t) for t in tok[1:4]], dtype=torch.float32)
            elif tok[0] == "Ns":
                colors[cur]["shininess"] = torch.tensor([float(tok[1])], dtype=torch.float32)
    return colors, tex


def _load_image_f32(path):
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(np.ascontiguousarray(arr))


def load_obj(path, load_textures=True, device="cpu"):
    """Same return structure as ``pytorch3d.io.load_obj``: ``(verts, faces, aux)`` with
    ``faces.verts_idx`` / ``faces.textures_idx`` int64 (F,3) and ``aux.verts_uvs`` (VT,2),
    ``aux.texture_images`` {material: (H,W,3) float32}."""
    verts, uvs, normals = [], [], []
    f_v, f_t, f_n, f_m = [], [], [], []
    mtl_files, mat_names, cur_mat = [], [], -1
    with open(path, "r") as fh:
        for line in fh:
            if not line or line[0] == "#":
                continue
            tok = line.split()
            if not tok:
                continue
            k = tok[0]
            if k == "v":
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "vt":
                uvs.append((float(tok[1]), float(tok[2])))
            elif k == "vn":
                normals.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "mtllib":
                mtl_files.append(" ".join(tok[1:]))
            elif k == "usemtl":
                name = tok[1]
                if name not in mat_names:
                    mat_names.append(name)
                cur_mat = mat_names.index(name)
            elif k == "f":
                vi, ti, ni = [], [], []
                for t in tok[1:]:
                    parts = t.split("/")
                    a = int(parts[0])
                    vi.append(a - 1 if a > 0 else len(verts) + a)
                    if len(parts) > 1 and parts[1] != "":
                        b = int(parts[1])
                        ti.append(b - 1 if b > 0 else len(uvs) + b)
                    if len(parts) > 2 and parts[2] != "":
                        c = int(parts[2])
                        ni.append(c - 1 if c > 0 else len(normals) + c)
                for i in range(len(vi) - 2):            # fan triangulation
                    f_v.append((vi[0], vi[i + 1], vi[i + 2]))
                    f_t.append((ti[0], ti[i + 1], ti[i + 2]) if len(ti) == len(vi) else (-1, -1, -1))
                    f_n.append((ni[0], ni[i + 1], ni[i + 2]) if len(ni) == len(vi) else (-1, -1, -1))
                    f_m.append(cur_mat)

    def _t(lst, dtype, shape):
        if not lst:
            return torch.zeros(shape, dtype=dtype, device=device)
        return torch.tensor(lst, dtype=dtype, device=device)

    verts_t = _t(verts, torch.float32, (0, 3))
    has_uv = len(uvs) > 0 and all(t[0] >= 0 for t in f_t)
    faces = Faces(verts_idx=_t(f_v, torch.int64, (0, 3)),
                  normals_idx=_t(f_n, torch.int64, (0, 3)),
                  textures_idx=_t(f_t, torch.int64, (0, 3)) if has_uv else None,
                  materials_idx=_t(f_m, torch.int64, (0,)))
    colors, tex_images = {}, {}
    base = os.path.dirname(os.path.abspath(path))
    for m in mtl_files:
        c, tfiles = _parse_mtl(os.path.join(base, m))
        colors.update(c)
        if load_textures:
            for name, fn in tfiles.items():
                p = os.path.join(base, fn)
                if os.path.isfile(p):
                    tex_images[name] = _load_image_f32(p).to(device)
    aux = Properties(normals=_t(normals, torch.float32, (0, 3)) if normals else None,
                     verts_uvs=_t(uvs, torch.float32, (0, 2)) if uvs else None,
                     material_colors=colors or None,
                     texture_images=tex_images or None,
                     texture_atlas=None)
    return verts_t, faces, aux


def synthesize_uvs(verts):
    """UVs for meshes that ship without any (teapot, SURVEY.md D3 -- the reference would
    crash there, so there is no behaviour to match): spherical parametrisation about the
    centroid, per-vertex (faces_uvs == faces).  u = atan2(x,z)/(2pi)+0.5, v = acos(-y/r)/pi."""
    v = verts - verts.mean(dim=0, keepdim=True)
    r = v.norm(dim=1).clamp_min(1e-8)
    u = torch.atan2(v[:, 0], v[:, 2]) / (2 * torch.pi) + 0.5
    w = torch.acos((-v[:, 1] / r).clamp(-1, 1)) / torch.pi
    return torch.stack([u, w], dim=1).to(torch.float32)


def save_obj(path, verts, faces, verts_uvs=None, faces_uvs=None, texture_map=None, decimal_places=6):
    """Writes ``<path>`` (+ ``.mtl`` + ``.png`` next to it when a texture is given), the
    artefact set of ``IO().save_mesh(final_mesh, .../final.obj)`` (first_approach.py:225)."""
    from PIL import Image
    base, _ = os.path.splitext(path)
    stem = os.path.basename(base)
    verts = verts.detach().cpu().reshape(-1, 3).numpy()
    faces = faces.detach().cpu().reshape(-1, 3).numpy()
    with open(path, "w") as f:
        has_tex = texture_map is not None and verts_uvs is not None and faces_uvs is not None
        if has_tex:
            f.write(f"\nmtllib {stem}.mtl\nusemtl mesh\n\n")
        fmt = f"%.{decimal_places}f"
        for v in verts:
            f.write("v " + " ".join(fmt % x for x in v) + "\n")
        if has_tex:
            uv = verts_uvs.detach().cpu().reshape(-1, 2).numpy()
            fuv = faces_uvs.detach().cpu().reshape(-1, 3).numpy()
            for t in uv:
                f.write("vt " + " ".join(fmt % x for x in t) + "\n")
            for a, b in zip(faces, fuv):
                f.write("f " + " ".join(f"{int(i) + 1}/{int(j) + 1}" for i, j in zip(a, b)) + "\n")
        else:
            for a in faces:
                f.write("f " + " ".join(str(int(i) + 1) for i in a) + "\n")
    if has_tex:
        tex = texture_map.detach().cpu().reshape(texture_map.shape[-3], texture_map.shape[-2], 3)
        img = (tex.clamp(0, 1) * 255.0).round().to(torch.uint8).numpy()
        Image.fromarray(img).save(base + ".png")
        with open(base + ".mtl", "w") as f:
            f.write(f"newmtl mesh\nmap_Kd {stem}.png\nKa 1.000 1.000 1.000\nKd 1.000 1.000 1.000\n"
                    "Ks 0.000 0.000 0.000\nNs 10.0\n")
