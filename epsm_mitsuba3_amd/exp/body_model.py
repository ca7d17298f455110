"""The torch module the ``human`` configuration optimises through (EPSM/exp/human.py:201-250, optim_human.py:54-60):
SMPL's FUNCTION -- shape blend shapes, joint regression, pose blend shapes, a 24-joint kinematic chain and
linear-blend skinning -- over SYNTHETIC assets, because the reference's model files (``smplpytorch`` model root,
``UV_Processed.mat``) are licensed data that is not part of its repository.

What is kept from the reference's ``SMPL`` wrapper, because the path under test depends on it:

* ``gen_mesh(pose_params (1,72), shape_params (1,10)) -> (1, 7829, 3)``: 6 890 model vertices re-indexed through
  ``verts_temp`` (1-based, as ``ALP_UV["All_vertices"]``) into the 7 829 vertices of the UV atlas, i.e. 939 vertices on chart
  seams appear twice.  The scatter of the backward pass therefore adds to rows that are different parameters of the
  renderer but the same degree of freedom of the model; ``verts[:, verts_temp - 1]`` sums them in its backward.
* ``center_idx = 0``: the posed root joint is subtracted.
* axis-angle pose, root orientation first; the standard SMPL kinematic tree.

What is synthetic: the template (ten closed tubes -- torso, head, legs, feet, arms, hands -- 6 890 vertices, 13 740 faces),
the skinning weights (distance to the bones, four per vertex), the joint regressor (Gaussian weights over the vertices
around a joint), ten smooth shape directions and a low-rank set of 207 pose-corrective directions.  The model is
y-up and in metres like SMPL."""
from __future__ import annotations

import math

import numpy as np
import torch

N_VERTS, N_ATLAS, N_JOINTS = 6890, 7829, 24
PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]     # SMPL kintree_table[0]

# designed rest joints (x: to the figure's left, y: up, z: forward)
_J = np.array([
    [0.00, 0.92, 0.00], [0.09, 0.86, 0.00], [-0.09, 0.86, 0.00], [0.00, 1.03, 0.00],
    [0.10, 0.48, 0.00], [-0.10, 0.48, 0.00], [0.00, 1.16, 0.00],
    [0.10, 0.08, -0.02], [-0.10, 0.08, -0.02], [0.00, 1.26, 0.00],
    [0.10, 0.03, 0.10], [-0.10, 0.03, 0.10], [0.00, 1.45, 0.00],
    [0.07, 1.38, 0.00], [-0.07, 1.38, 0.00], [0.00, 1.56, 0.00],
    [0.19, 1.40, 0.00], [-0.19, 1.40, 0.00], [0.45, 1.40, 0.00], [-0.45, 1.40, 0.00],
    [0.70, 1.40, 0.00], [-0.70, 1.40, 0.00], [0.79, 1.40, 0.00], [-0.79, 1.40, 0.00]])

# (name, axis start, axis end, rings, sectors, (radius a, radius b) at the start / middle / end, seam rings)
# radius a lies along `side`, radius b along axis x side; sum(rings * sectors) + 2 caps per tube = 6890
_TUBES = [
    ("torso", [0, 0.80, 0], [0, 1.50, 0], 31, 58, ((0.15, 0.10), (0.16, 0.11), (0.07, 0.06)), (8, 16, 24)),
    ("head", [0, 1.49, 0], [0, 1.76, 0], 30, 32, ((0.05, 0.05), (0.09, 0.10), (0.05, 0.06)), (10, 20)),
    ("leg_l", [0.09, 0.86, 0], [0.10, 0.06, -0.02], 40, 28, ((0.075, 0.08), (0.055, 0.06), (0.035, 0.04)), (13, 20)),
    ("leg_r", [-0.09, 0.86, 0], [-0.10, 0.06, -0.02], 40, 28, ((0.075, 0.08), (0.055, 0.06), (0.035, 0.04)), (13, 20)),
    ("foot_l", [0.10, 0.035, -0.07], [0.10, 0.03, 0.19], 10, 16, ((0.035, 0.03), (0.045, 0.03), (0.035, 0.02)), (5,)),
    ("foot_r", [-0.10, 0.035, -0.07], [-0.10, 0.03, 0.19], 10, 16, ((0.035, 0.03), (0.045, 0.03), (0.035, 0.02)), (5,)),
    ("arm_l", [0.16, 1.40, 0], [0.71, 1.40, 0], 34, 20, ((0.05, 0.05), (0.04, 0.04), (0.028, 0.03)), (11, 18)),
    ("arm_r", [-0.16, 1.40, 0], [-0.71, 1.40, 0], 34, 20, ((0.05, 0.05), (0.04, 0.04), (0.028, 0.03)), (11, 18)),
    ("hand_l", [0.70, 1.40, 0], [0.88, 1.40, 0], 8, 12, ((0.03, 0.015), (0.045, 0.015), (0.02, 0.01)), (4,)),
    ("hand_r", [-0.70, 1.40, 0], [-0.88, 1.40, 0], 8, 12, ((0.03, 0.015), (0.045, 0.015), (0.02, 0.01)), (4,)),
]


def _tube(p0, p1, rings, sectors, radii):
    """Closed tube: `rings` rings of `sectors` vertices + two cap centres; faces wound outwards."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    ax = (p1 - p0) / np.linalg.norm(p1 - p0)
    side = np.cross(ax, [0.0, 0.0, 1.0]) if abs(ax[2]) < 0.9 else np.cross(ax, [0.0, 1.0, 0.0])
    side /= np.linalg.norm(side)
    up = np.cross(ax, side)
    t = np.linspace(0.0, 1.0, rings)
    (a0, b0), (a1, b1), (a2, b2) = radii
    # quadratic through the three radii, rounded off towards the two ends
    lag = lambda x0, x1, x2: x0 * (t - 0.5) * (t - 1) * 2 - x1 * t * (t - 1) * 4 + x2 * t * (t - 0.5) * 2
    close = np.sqrt(np.clip(1.0 - (2 * t - 1) ** 8, 0.04, 1.0))
    ra, rb = lag(a0, a1, a2) * close, lag(b0, b1, b2) * close
    ang = np.linspace(0, 2 * np.pi, sectors, endpoint=False)
    c = p0[None, :] + t[:, None] * (p1 - p0)[None, :]
    v = (c[:, None, :] + (ra[:, None] * np.cos(ang)[None, :])[..., None] * side + (rb[:, None] * np.sin(ang)[None, :])[..., None] * up)
    verts = np.concatenate([v.reshape(-1, 3), p0[None], p1[None]])
    f = []
    for j in range(rings - 1):
        for i in range(sectors):
            p, q = j * sectors + i, j * sectors + (i + 1) % sectors
            f += [[p, q + sectors, q], [p, p + sectors, q + sectors]]
    nb, nt = rings * sectors, rings * sectors + 1
    for i in range(sectors):
        f.append([nb, i, (i + 1) % sectors])
        f.append([nt, (rings - 1) * sectors + (i + 1) % sectors, (rings - 1) * sectors + i])
    f = np.array(f)
    # outward orientation: flip if the first face's normal points at the axis
    a, b, c3 = verts[f[0, 0]], verts[f[0, 1]], verts[f[0, 2]]
    if np.dot(np.cross(b - a, c3 - a), (a + b + c3) / 3 - (p0 + (p1 - p0) * t[0])) < 0:
        f = f[:, ::-1]
    return verts, f


def _seg_dist(v, a, b):
    ab = b - a
    t = np.clip(((v - a) @ ab) / max(float(ab @ ab), 1e-12), 0.0, 1.0)
    return np.linalg.norm(v - (a + t[:, None] * ab), axis=1)


def build_assets(seed: int = 0):
    """Template, faces, atlas re-indexing, weights, regressor, shape / pose directions -- numpy, deterministic."""
    rng = np.random.default_rng(seed)
    verts, faces, tube_of, ring_of, col_of = [], [], [], [], []
    seams = []                                   # (vertex, kind, tube, index) candidates for atlas duplicates, in order
    base = 0
    for ti, (name, p0, p1, rings, sectors, radii, seam_rings) in enumerate(_TUBES):
        v, f = _tube(p0, p1, rings, sectors, radii)
        verts.append(v); faces.append(f + base)
        tube_of += [ti] * len(v)
        ring_of += [j for j in range(rings) for _ in range(sectors)] + [-1, -1]
        col_of += [i for _ in range(rings) for i in range(sectors)] + [-1, -1]
        for col in (0, sectors // 2):            # front / back charts: two seams along the tube
            seams += [(base + j * sectors + col, "col", ti, col) for j in range(rings)]
        base += len(v)
    base = 0
    for ti, (name, p0, p1, rings, sectors, radii, seam_rings) in enumerate(_TUBES):
        for r in seam_rings:                     # upper / lower charts: seams around the tube
            seams += [(base + r * sectors + i, "ring", ti, r) for i in range(sectors)]
        base += rings * sectors + 2
    template = np.concatenate(verts)
    faces = np.concatenate(faces)
    assert template.shape[0] == N_VERTS, template.shape
    tube_of, ring_of, col_of = np.array(tube_of), np.array(ring_of), np.array(col_of)

    # ---- atlas: 939 seam vertices appear twice; the faces on one side of a seam use the copy
    dup_of, kind_of = {}, {}
    for v, kind, ti, idx in seams:
        if v not in dup_of and len(dup_of) < N_ATLAS - N_VERTS:
            dup_of[v] = N_VERTS + len(dup_of)
            kind_of[v] = kind
    assert len(dup_of) == N_ATLAS - N_VERTS, len(dup_of)
    verts_temp = np.concatenate([np.arange(N_VERTS), np.array(sorted(dup_of, key=dup_of.get))]) + 1      # 1-based
    atlas_faces = faces.copy()
    for fi, f in enumerate(faces):
        rings_f, cols_f = ring_of[f], col_of[f]
        sectors = _TUBES[tube_of[f[0]]][4]
        for c in range(3):
            v = int(f[c])
            if v not in dup_of:
                continue
            if kind_of[v] == "col":              # the quad column that ENDS in this column
                others = cols_f[np.arange(3) != c]
                if np.any(others == (col_of[v] - 1) % sectors):
                    atlas_faces[fi, c] = dup_of[v]
            else:                                # the quad row that ends in this ring
                if np.any(rings_f[np.arange(3) != c] == ring_of[v] - 1):
                    atlas_faces[fi, c] = dup_of[v]

    # ---- skinning weights: Gaussian of the distance to the bone a joint drives (joint -> its first child; a stub for leaves)
    child = {p: j for j, p in reversed(list(enumerate(PARENTS))) if p >= 0}
    d = np.zeros((N_VERTS, N_JOINTS))
    for j in range(N_JOINTS):
        a = _J[j]
        b = _J[child[j]] if j in child else a + (a - _J[PARENTS[j]]) * 0.8
        d[:, j] = _seg_dist(template, a, b)
    w = np.exp(-(d / 0.06) ** 2)
    # a part only follows the joints of its own chain (the hands hang next to nothing in the T pose, but the two legs and the
    # collars / spine are close to each other)
    chains = {"torso": [0, 3, 6, 9, 12, 13, 14, 1, 2], "head": [12, 15], "leg_l": [1, 4, 7, 0], "leg_r": [2, 5, 8, 0],
              "foot_l": [7, 10], "foot_r": [8, 11], "arm_l": [13, 16, 18, 20, 9], "arm_r": [14, 17, 19, 21, 9],
              "hand_l": [20, 22], "hand_r": [21, 23]}
    for ti, t in enumerate(_TUBES):
        mask = np.zeros(N_JOINTS)
        mask[chains[t[0]]] = 1.0
        w[tube_of == ti] *= mask[None, :]
    order = np.argsort(-w, axis=1)
    keep = np.zeros_like(w)
    np.put_along_axis(keep, order[:, :4], 1.0, axis=1)                 # four influences per vertex, like SMPL
    w = w * keep + 1e-12 * keep
    weights = w / w.sum(1, keepdims=True)

    # ---- joint regressor: Gaussian weights over the vertices around the designed joint (rows sum to 1)
    jd = np.linalg.norm(template[None, :, :] - _J[:, None, :], axis=2)
    jr = np.exp(-(jd / 0.08) ** 2)
    thresh = -np.sort(-jr, axis=1)[:, 96:97]
    jr = np.where(jr >= thresh, jr, 0.0)
    # symmetric vertex sets regress onto the axis; what is left of the designed positions is corrected by a constant row sum
    j_regressor = jr / jr.sum(1, keepdims=True)

    # ---- shape directions: scale, height, girth, limb length, then smooth sinusoid fields
    centre = np.array([0.0, 0.92, 0.0])
    rel = template - centre
    sd = np.zeros((N_VERTS, 3, 10))
    sd[:, :, 0] = rel * 0.04
    sd[:, 1, 1] = rel[:, 1] * 0.05
    sd[:, 0, 2] = rel[:, 0] * 0.06 * (np.abs(rel[:, 0]) < 0.2); sd[:, 2, 2] = rel[:, 2] * 0.08
    sd[:, 0, 3] = rel[:, 0] * 0.05 * (np.abs(rel[:, 0]) >= 0.2)
    for m in range(4, 10):
        k = rng.normal(size=3) * 3.0
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        sd[:, :, m] = 0.008 * np.sin(template @ k + rng.uniform(0, 6.28))[:, None] * dirn[None, :]

    # ---- pose-corrective directions, rank 6: (R - I) features -> six smooth displacement fields
    pu = np.zeros((N_VERTS * 3, 6))
    for m in range(6):
        k = rng.normal(size=3) * 4.0
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        pu[:, m] = (0.01 * np.sin(template @ k + rng.uniform(0, 6.28))[:, None] * dirn[None, :]).reshape(-1)
    pw = rng.normal(size=(6, 207)) / math.sqrt(207)
    return {"template": template, "faces": faces, "atlas_faces": atlas_faces, "verts_temp": verts_temp,
            "weights": weights, "j_regressor": j_regressor, "shapedirs": sd, "posedirs_u": pu, "posedirs_w": pw}


def rodrigues(r: torch.Tensor) -> torch.Tensor:
    """(..., 3) axis-angle -> (..., 3, 3); differentiable at 0."""
    th = torch.sqrt((r * r).sum(-1, keepdim=True) + 1e-16)
    k = r / th
    z = torch.zeros_like(k[..., 0])
    K = torch.stack([z, -k[..., 2], k[..., 1], k[..., 2], z, -k[..., 0], -k[..., 1], k[..., 0], z], -1).reshape(r.shape[:-1] + (3, 3))
    eye = torch.eye(3, dtype=r.dtype, device=r.device).expand(K.shape)
    s, c = torch.sin(th)[..., None], torch.cos(th)[..., None]
    return eye + s * K + (1 - c) * (K @ K)


class BodyLayer(torch.nn.Module):
    """``smplpytorch``'s ``SMPL_Layer(center_idx=0)`` in function: ``forward(pose (B,72), th_betas (B,10)) -> (verts (B,6890,3),
    joints (B,24,3))``."""

    def __init__(self, center_idx=0, seed=0, dtype=torch.float32):
        super().__init__()
        a = build_assets(seed)
        self.center_idx = center_idx
        t = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=dtype)
        self.register_buffer("template", t(a["template"]))
        self.register_buffer("shapedirs", t(a["shapedirs"]))
        self.register_buffer("posedirs_u", t(a["posedirs_u"]))
        self.register_buffer("posedirs_w", t(a["posedirs_w"]))
        self.register_buffer("j_regressor", t(a["j_regressor"]))
        self.register_buffer("weights", t(a["weights"]))
        self.faces = a["faces"]
        self.atlas_faces = a["atlas_faces"]
        self.verts_temp = a["verts_temp"]

    def forward(self, pose: torch.Tensor, th_betas: torch.Tensor | None = None):
        B = pose.shape[0]
        dt, dev = self.template.dtype, self.template.device
        pose = pose.to(dev, dt)
        betas = torch.zeros((B, 10), dtype=dt, device=dev) if th_betas is None else th_betas.to(dev, dt)
        v_shaped = self.template[None] + torch.einsum("vcm,bm->bvc", self.shapedirs, betas)
        J = torch.einsum("jv,bvc->bjc", self.j_regressor, v_shaped)
        R = rodrigues(pose.reshape(B, N_JOINTS, 3))
        feat = (R[:, 1:] - torch.eye(3, dtype=dt, device=dev)).reshape(B, 207)
        v_posed = v_shaped + ((feat @ self.posedirs_w.T) @ self.posedirs_u.T).reshape(B, N_VERTS, 3)
        # kinematic chain: world rotation / position of every joint
        Rw, tw = [R[:, 0]], [J[:, 0]]
        for j in range(1, N_JOINTS):
            p = PARENTS[j]
            Rw.append(Rw[p] @ R[:, j])
            tw.append(tw[p] + (Rw[p] @ (J[:, j] - J[:, p])[..., None])[..., 0])
        Rw, tw = torch.stack(Rw, 1), torch.stack(tw, 1)                       # (B,24,3,3), (B,24,3)
        # skinning transform of joint j: x -> Rw_j (x - J_j) + tw_j
        off = tw - (Rw @ J[..., None])[..., 0]
        Rv = torch.einsum("vj,bjrc->bvrc", self.weights, Rw)
        tv = torch.einsum("vj,bjc->bvc", self.weights, off)
        verts = (Rv @ v_posed[..., None])[..., 0] + tv
        joints = tw
        if self.center_idx is not None:
            c = joints[:, self.center_idx:self.center_idx + 1]
            verts, joints = verts - c, joints - c
        return verts, joints


class SMPL:
    """EPSM/exp/human.py:201-250 in interface: ``gen_mesh``, ``faces`` (into the 7 829 atlas vertices), ``verts_temp``."""

    def __init__(self, device="cpu", seed=0):
        self.device = torch.device(device)
        self.smpl_layer = BodyLayer(center_idx=0, seed=seed).to(self.device)
        self.faces = self.smpl_layer.atlas_faces
        self.verts_temp = torch.tensor(self.smpl_layer.verts_temp, dtype=torch.long, device=self.device)

    def gen_mesh(self, pose_params: torch.Tensor, shape_params: torch.Tensor | None = None) -> torch.Tensor:
        verts, _ = self.smpl_layer(pose_params, th_betas=shape_params)
        return verts[:, self.verts_temp - 1]                                   # (1, 7829, 3)
