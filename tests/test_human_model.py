"""The skinned stand-in for SMPL (exp/human_tube.py): rest pose, rigid root, differentiable pose."""
import torch

from epsm_mitsuba3_amd.exp.human_tube import SkinnedTube, _rodrigues


def test_rodrigues_and_skinning():
    r = torch.tensor([0.2, -0.4, 0.3], dtype=torch.float64)
    R = _rodrigues(r)
    assert torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-10) and abs(float(torch.det(R)) - 1) < 1e-10
    assert torch.allclose(R @ (r / r.norm()), r / r.norm(), atol=1e-10)                 # the axis is fixed
    m = SkinnedTube()
    v0 = m.gen_mesh(torch.zeros(1, 6))[0]
    assert torch.allclose(v0, m.rest, atol=1e-6)
    assert m.faces.min() == 0 and m.faces.max() == v0.shape[0] - 1
    pose = torch.tensor([[0.0, 0.4, 0.0, 0.0, -0.5, 0.0]], requires_grad=True)
    v = m.gen_mesh(pose)[0]
    low = m.rest[:, 2] < 0.3
    assert torch.allclose(v[low], m.rest[low], atol=1e-2)                                # the root bone does not move
    assert float((v - m.rest).abs().max()) > 0.2
    (v * torch.ones_like(v)).sum().backward()                                            # optim_human.py:120-121
    assert pose.grad is not None and float(pose.grad.abs().max()) > 0
