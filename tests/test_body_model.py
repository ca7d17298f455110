"""The torch module of the ``human`` configuration (exp/body_model.py: SMPL's function over synthetic assets) against an
independent per-vertex restatement with 4x4 matrices, and the properties optim_human.py relies on."""
import numpy as np
import pytest
import torch

from epsm_mitsuba3_amd.exp import body_model as bm


@pytest.fixture(scope="module")
def layer():
    return bm.BodyLayer(center_idx=0, dtype=torch.float64)


def _rot(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _naive(a, pose, betas):
    """SMPL as published (Loper et al. 2015, eq. 2-4), one vertex at a time."""
    v_shaped = a["template"] + a["shapedirs"] @ betas
    J = a["j_regressor"] @ v_shaped
    R = [_rot(pose[3 * j:3 * j + 3]) for j in range(24)]
    feat = np.concatenate([(R[j] - np.eye(3)).reshape(-1) for j in range(1, 24)])
    v_posed = v_shaped + (a["posedirs_u"] @ (a["posedirs_w"] @ feat)).reshape(-1, 3)
    G = [None] * 24
    for j in range(24):
        M = np.eye(4)
        M[:3, :3] = R[j]
        M[:3, 3] = J[j] - (J[bm.PARENTS[j]] if j else 0)
        G[j] = M if j == 0 else G[bm.PARENTS[j]] @ M
    A = []
    for j in range(24):                       # remove the rest pose: G_j * [I | -J_j]
        M = np.eye(4); M[:3, 3] = -J[j]
        A.append(G[j] @ M)
    out = np.zeros_like(v_posed)
    for v in range(v_posed.shape[0]):
        T = sum(a["weights"][v, j] * A[j] for j in np.nonzero(a["weights"][v])[0])
        out[v] = (T @ np.append(v_posed[v], 1.0))[:3]
    joints = np.stack([G[j][:3, 3] for j in range(24)])
    return out - joints[0], joints - joints[0]


def test_counts_and_atlas():
    a = bm.build_assets()
    assert a["template"].shape == (6890, 3) and a["verts_temp"].shape == (7829,)
    assert a["faces"].shape == a["atlas_faces"].shape == (13740, 3)
    assert a["verts_temp"].min() == 1 and a["verts_temp"].max() == 6890                     # 1-based, as the .mat file
    # the atlas is the same surface: every atlas face is its model face after the re-indexing
    assert np.array_equal(a["verts_temp"][a["atlas_faces"]] - 1, a["faces"])
    assert len(np.unique(a["atlas_faces"])) == 7829                                          # every copy is used
    assert len(np.unique(a["verts_temp"][6890:])) == 939                                     # 939 distinct seam vertices
    w = a["weights"]
    assert np.allclose(w.sum(1), 1) and ((w > 0).sum(1) <= 4).all()
    assert np.allclose(a["j_regressor"].sum(1), 1)
    # closed, outward-oriented tubes: positive volume, every edge shared by exactly two faces
    f, T = a["faces"], a["template"]
    assert np.einsum("ij,ij->i", T[f[:, 0]], np.cross(T[f[:, 1]], T[f[:, 2]])).sum() > 0
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()


def test_matches_the_per_vertex_restatement(layer):
    a = bm.build_assets()
    rng = np.random.default_rng(3)
    for scale in (0.0, 0.2, 1.0):
        pose = (rng.random(72) - 0.5) * scale
        betas = rng.normal(size=10)
        v, j = layer(torch.tensor(pose)[None], torch.tensor(betas)[None])
        nv, nj = _naive(a, pose, betas)
        assert np.abs(v[0].numpy() - nv).max() < 1e-9
        assert np.abs(j[0].numpy() - nj).max() < 1e-9


def test_rest_pose_root_rotation_and_locality(layer):
    a = bm.build_assets()
    betas = torch.tensor(np.linspace(-1, 1, 10))[None]
    v0, j0 = layer(torch.zeros(1, 72, dtype=torch.float64), betas)
    vs = a["template"] + a["shapedirs"] @ betas[0].numpy()
    assert np.abs(v0[0].numpy() - (vs - a["j_regressor"][0] @ vs)).max() < 1e-12
    # root orientation: a rigid rotation about the (centred) root joint
    pose = torch.zeros(1, 72, dtype=torch.float64)
    pose[0, :3] = torch.tensor([0.3, -0.2, 0.5])
    v1, _ = layer(pose, betas)
    assert np.abs(v1[0].numpy() - v0[0].numpy() @ _rot(pose[0, :3].numpy()).T).max() < 1e-9
    # bending the left elbow (joint 18) moves the left forearm and hand only
    pose = torch.zeros(1, 72, dtype=torch.float64)
    pose[0, 54:57] = torch.tensor([0.0, 0.6, 0.0])
    v2, j2 = layer(pose, betas)
    moved = (v2 - v0)[0].norm(dim=1) > 5e-3                                       # (pose correctives move everything a little)
    x = torch.tensor(a["template"][:, 0])
    assert moved[x > 0.5].all() and not moved[x < 0.3].any()
    assert (j2 - j0)[0, [20, 22]].norm(dim=1).min() > 0.1 and (j2 - j0)[0, :18].abs().max() < 1e-12


def test_gradients_chain_through_the_atlas_copies():
    from epsm_mitsuba3_amd.optim import chain_vertex_grads
    model = bm.SMPL("cpu")
    model.smpl_layer = model.smpl_layer.double()
    g = torch.Generator().manual_seed(0)
    pose = ((torch.rand(1, 72, generator=g, dtype=torch.float64) - 0.5) * 0.2).requires_grad_()
    shape = torch.zeros(1, 10, dtype=torch.float64)
    grad = torch.randn(7829, 3, generator=g, dtype=torch.float64)
    verts = model.gen_mesh(pose, shape)
    assert verts.shape == (1, 7829, 3)
    chain_vertex_grads(verts[0], grad)                                              # optim_human.py:118-121
    # the same through the 6 890 model vertices: the two copies of a seam vertex add up
    g6890 = torch.zeros(6890, 3, dtype=torch.float64).index_add_(0, model.verts_temp - 1, grad)
    pose2 = pose.detach().clone().requires_grad_()
    v, _ = model.smpl_layer(pose2, shape)
    (v[0] * g6890).sum().backward()
    assert torch.allclose(pose.grad, pose2.grad, rtol=1e-10, atol=1e-12) and float(pose.grad.abs().max()) > 0
    # and against finite differences
    f = lambda p: (model.gen_mesh(p, shape)[0] * grad).sum()
    assert torch.autograd.gradcheck(f, (pose.detach().clone().requires_grad_(),), eps=1e-6, atol=1e-6)
