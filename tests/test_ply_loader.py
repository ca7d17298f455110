"""scene.load_ply: the `ply` shapes of the reference's experiment files (EPSM/exp/glassslab.py:150,162) load as meshes."""
import struct

import numpy as np
import pytest

from epsm_mitsuba3_amd import scene as S

V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 0.5, 1]], dtype=np.float64)
N = np.array([[0, 0, -1]] * 4 + [[0, 0, 1]], dtype=np.float64)
FACES = [[0, 1, 2, 3], [0, 1, 4], [1, 2, 4]]                 # a quad (two triangles as a fan) and two triangles
TRIS = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 4], [1, 2, 4]])


def _write(path, fmt, normals):
    props = "property float x\nproperty float y\nproperty float z\n" + ("property float nx\nproperty float ny\nproperty float nz\n" if normals else "")
    head = f"ply\nformat {fmt} 1.0\ncomment made by the test\nelement vertex {len(V)}\n{props}element face {len(FACES)}\nproperty list uchar int vertex_indices\nend_header\n"
    with open(path, "wb") as f:
        f.write(head.encode())
        if fmt == "ascii":
            for i in range(len(V)):
                f.write((" ".join(str(x) for x in (list(V[i]) + (list(N[i]) if normals else []))) + "\n").encode())
            for fc in FACES:
                f.write((f"{len(fc)} " + " ".join(map(str, fc)) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            for i in range(len(V)):
                f.write(struct.pack(e + ("6f" if normals else "3f"), *(list(V[i]) + (list(N[i]) if normals else []))))
            for fc in FACES:
                f.write(struct.pack(e + "B" + f"{len(fc)}i", len(fc), *fc))


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
@pytest.mark.parametrize("normals", [False, True])
def test_ply_round_trip(tmp_path, fmt, normals):
    p = tmp_path / "m.ply"
    _write(p, fmt, normals)
    v, n, f = S.load_ply(str(p))
    assert np.allclose(v, V) and np.array_equal(f, TRIS)
    assert (n is None) == (not normals) and (n is None or np.allclose(n, N))


def test_ply_shape_in_a_scene_dict(tmp_path):
    """`{'type': 'ply', 'filename': ...}` builds the same scene as the inline mesh with the same data (to_world applied)."""
    _write(tmp_path / "m.ply", "binary_little_endian", False)
    tw = np.eye(4); tw[:3, 3] = [0.5, -1.0, 2.0]
    base = {"type": "scene", "cam": {"type": "perspective", "fov": 40, "film": {"type": "hdrfilm", "width": 8, "height": 8}}}
    a = S.Scene.from_dict({**base, "m": {"type": "ply", "filename": str(tmp_path / "m.ply"), "to_world": tw, "bsdf": {"type": "diffuse"}}}, device="cpu")
    b = S.Scene.from_dict({**base, "m": {"type": "mesh", "vertices": V, "faces": TRIS, "to_world": tw, "bsdf": {"type": "diffuse"}}}, device="cpu")
    assert np.allclose(a.positions.numpy(), b.positions.numpy()) and a.T == b.T == 4


def test_bad_files_are_refused(tmp_path):
    (tmp_path / "x.ply").write_bytes(b"plx\n")
    with pytest.raises(ValueError, match="not a PLY"):
        S.load_ply(str(tmp_path / "x.ply"))
