"""oracle/mesh_ref.py -- TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the view-independent mesh regularisers the reference adds for
optimization_target 'mesh'/'both' (losses.py:84-87,93-96,112-115,121-124):
``pytorch3d.loss.mesh_edge_loss``, ``mesh_laplacian_smoothing(method='uniform')``,
``mesh_normal_consistency`` for ONE mesh (N = 1), plus ``F.mse_loss(verts, target_verts)``.

PARITY UNPINNED: PyTorch3D (unpinned version) is absent and the reference's mesh/both branches
cannot be executed here (they call the stubbed pytorch3d.loss names), so these follow the
published definitions and are pinned by hand-computable cases in
tests/test_oracle_mesh.py.  Gradients come from torch autograd (fp64 in the tests).
"""
import torch


def unique_edges(faces):
    """(F,3) int64 -> (E,2) sorted unique undirected edges [min,max], in PyTorch3D's order
    (ascending key V*e0 + e1), and face->edge map (F,3) for the half-edges (v1v2, v2v0, v0v1)."""
    f = faces.long()
    e = torch.cat([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], dim=0)          # (3F,2) as PyTorch3D stacks them
    e = torch.sort(e, dim=1).values
    V = int(f.max()) + 1 if f.numel() else 0
    key = e[:, 0] * V + e[:, 1]
    ukey, inv = torch.unique(key, sorted=True, return_inverse=True)
    edges = torch.stack([ukey // V, ukey % V], dim=1)
    F = f.shape[0]
    return edges, inv.reshape(3, F).t().contiguous()


def mesh_edge_loss_ref(verts, faces, target_length=0.0):
    edges, _ = unique_edges(faces)
    v0, v1 = verts[edges[:, 0]], verts[edges[:, 1]]
    return (((v0 - v1).norm(dim=1, p=2) - target_length) ** 2).sum() / edges.shape[0]


def mesh_laplacian_smoothing_ref(verts, faces):
    """uniform Laplacian: L[i,j] = 1/deg(i) for neighbours, L[i,i] = -1; mean_i ||(L V)_i||_2"""
    edges, _ = unique_edges(faces)
    V = verts.shape[0]
    e0, e1 = edges[:, 0], edges[:, 1]
    deg = torch.zeros(V, dtype=verts.dtype).index_add_(0, e0, torch.ones_like(e0, dtype=verts.dtype))
    deg = deg.index_add_(0, e1, torch.ones_like(e1, dtype=verts.dtype))
    acc = torch.zeros_like(verts).index_add_(0, e0, verts[e1]).index_add_(0, e1, verts[e0])
    inv = torch.where(deg > 0, 1.0 / deg.clamp_min(1), torch.zeros_like(deg))
    y = acc * inv[:, None] - verts * (deg > 0).to(verts.dtype)[:, None]
    return y.norm(dim=1).sum() / V


def face_pairs(faces):
    """All pairs of faces sharing an edge -> (P,4) int64 rows (v0, v1, a, b): the shared edge
    (v0<v1) and the two opposite vertices, pairs ordered as PyTorch3D enumerates them (edges in
    ascending order; within an edge, faces in ascending half-edge order, pairs (i<j))."""
    edges, f2e = unique_edges(faces)
    F = faces.shape[0]
    edge_idx = f2e.reshape(F * 3)                                             # per face: its 3 edges
    vert_idx = faces.long().view(F, 1, 3).expand(F, 3, 3).reshape(F * 3, 3)
    edge_idx, order = torch.sort(edge_idx, stable=True)
    vert_idx = vert_idx[order]
    rows = []
    start = 0
    counts = torch.bincount(edge_idx, minlength=edges.shape[0]).tolist()
    for e, c in enumerate(counts):
        if c >= 2:
            v0, v1 = int(edges[e, 0]), int(edges[e, 1])
            others = [int(vert_idx[start + k].sum()) - v0 - v1 for k in range(c)]
            for i in range(c):
                for j in range(i + 1, c):
                    rows.append((v0, v1, others[i], others[j]))
        start += c
    return torch.tensor(rows, dtype=torch.int64).reshape(-1, 4)


def mesh_normal_consistency_ref(verts, faces):
    pairs = face_pairs(faces)
    if pairs.shape[0] == 0:
        return verts.sum() * 0
    v0, v1, a, b = (verts[pairs[:, k]] for k in range(4))
    n0 = torch.cross(v1 - v0, a - v0, dim=1)
    n1 = -torch.cross(v1 - v0, b - v0, dim=1)
    cos = torch.nn.functional.cosine_similarity(n0, n1, dim=1)
    return (1 - cos).sum() / pairs.shape[0]


def verts_mse_ref(verts, target):
    return torch.mean((verts - target) ** 2)
