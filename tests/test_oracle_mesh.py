"""Hand-computable cases for the mesh-regulariser restatement (oracle/mesh_ref.py).
PARITY UNPINNED w.r.t. pytorch3d.loss (absent): definitions per SURVEY.md A.6.  CPU only."""
import math

import torch

from oracle import mesh_ref as M

SQUARE = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
SQ_FACES = torch.tensor([[0, 1, 2], [0, 2, 3]])


def test_unique_edges_and_pairs():
    edges, f2e = M.unique_edges(SQ_FACES)
    assert edges.tolist() == [[0, 1], [0, 2], [0, 3], [1, 2], [2, 3]]
    assert f2e.shape == (2, 3)
    assert edges[f2e[0]].tolist() == [[1, 2], [0, 2], [0, 1]]          # half-edges v1v2, v2v0, v0v1
    pairs = M.face_pairs(SQ_FACES)
    assert pairs.tolist() == [[0, 2, 1, 3]]


def test_edge_loss_unit_square():
    # four unit sides + one diagonal of length sqrt(2): mean of squared lengths = (4 + 2) / 5
    assert abs(float(M.mesh_edge_loss_ref(SQUARE, SQ_FACES)) - 1.2) < 1e-6


def test_laplacian_uniform():
    # vertex 0: neighbours 1,2,3 -> mean (2/3,2/3,0) - (0,0,0); vertex 1: neighbours 0,2 -> (.5,.5,0)-(1,0,0)
    v = SQUARE
    y0 = (v[1] + v[2] + v[3]) / 3 - v[0]
    y1 = (v[0] + v[2]) / 2 - v[1]
    y2 = (v[0] + v[1] + v[3]) / 3 - v[2]
    y3 = (v[0] + v[2]) / 2 - v[3]
    want = sum(float(y.norm()) for y in (y0, y1, y2, y3)) / 4
    assert abs(float(M.mesh_laplacian_smoothing_ref(v, SQ_FACES)) - want) < 1e-6
    # a regular hexagon fan: the centre equals the mean of its ring -> zero Laplacian there
    ring = [[math.cos(k * math.pi / 3), math.sin(k * math.pi / 3), 0.0] for k in range(6)]
    verts = torch.tensor([[0., 0, 0]] + ring)
    faces = torch.tensor([[0, k + 1, (k + 1) % 6 + 1] for k in range(6)])
    edges, _ = M.unique_edges(faces)
    assert edges.shape[0] == 12


def test_normal_consistency_flat_fold_and_right_angle():
    assert abs(float(M.mesh_normal_consistency_ref(SQUARE, SQ_FACES))) < 1e-6           # coplanar: cos = 1
    bent = SQUARE.clone()
    bent[3] = torch.tensor([0., 0.5, math.sqrt(0.75)])       # rotate triangle (0,2,3) about edge 0-2? no: generic bend
    val = float(M.mesh_normal_consistency_ref(bent, SQ_FACES))
    assert 0 < val < 2
    # fold vertex 3 onto vertex 1's side of the diagonal (mirror image): normals opposite -> 1 - (-1) = 2
    folded = SQUARE.clone()
    folded[3] = torch.tensor([1., 0, 0]) + torch.tensor([0., 0., 0.])       # mirror of (0,1,0) about the line x=y is (1,0,0)
    assert abs(float(M.mesh_normal_consistency_ref(folded, SQ_FACES)) - 2.0) < 1e-5
    # regular tetrahedron: interior dihedral angle arccos(1/3), so the consistently oriented normals of two
    # adjacent faces are 180 - 70.53 = 109.47 deg apart: cos = -1/3 -> loss 1 + 1/3 for each of the 6 pairs
    tet = torch.tensor([[1., 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]])
    tf = torch.tensor([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])
    assert M.face_pairs(tf).shape[0] == 6
    assert abs(float(M.mesh_normal_consistency_ref(tet, tf)) - (1 + 1 / 3)) < 1e-5


def test_verts_mse():
    a = torch.zeros(3, 3)
    b = torch.ones(3, 3) * 2
    assert float(M.verts_mse_ref(a, b)) == 4.0
