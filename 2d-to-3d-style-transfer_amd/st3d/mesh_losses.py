"""View-independent mesh regularisers of the 'mesh' / 'both' optimisation targets
(reference losses.py:84-87 via pytorch3d.loss).  Not built yet (SURVEY.md 8,
kernels K14/K15 -- the texture target of configs 1-4 does not reach them): they fail loudly."""


def _todo(name):
    raise NotImplementedError(f"{name}: the vertex-optimisation path (optimization_target 'mesh'/'both') is not built "
                              "yet; optimization_target 'texture' is")


def verts_mse(verts, target_verts):
    _todo("verts_mse")


def mesh_edge_loss(mesh):
    _todo("mesh_edge_loss")


def mesh_laplacian_smoothing(mesh, method="uniform"):
    _todo("mesh_laplacian_smoothing")


def mesh_normal_consistency(mesh):
    _todo("mesh_normal_consistency")
