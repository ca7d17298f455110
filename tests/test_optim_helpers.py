"""Host-side helpers of the outer loop (EPSM/optim.py, optim_human.py)."""
import torch

from epsm_mitsuba3_amd.optim import chain_vertex_grads, resize, to_ldr
from epsm_mitsuba3_amd.params import ParamGrads


def test_chain_vertex_grads_reaches_module_parameters():
    # a toy "skinned mesh": vertices = template @ R(theta) + t, like optim_human.py's SMPL layer
    template = torch.randn(50, 3)
    theta = torch.tensor(0.3, requires_grad=True); t = torch.zeros(3, requires_grad=True)
    c, s = torch.cos(theta), torch.sin(theta)
    R = torch.stack([torch.stack([c, -s, torch.zeros(())]), torch.stack([s, c, torch.zeros(())]), torch.tensor([0.0, 0.0, 1.0])])
    verts = template @ R.T + t
    params = ParamGrads(50, 0, device="cpu", mesh_slices={"human": (0, 50)})
    params.pos += torch.randn(50, 3)
    chain_vertex_grads(verts, params.mesh_pos("human"))
    assert torch.allclose(t.grad, params.pos.sum(0), atol=1e-5)
    assert theta.grad is not None and float(theta.grad.abs()) > 0


def test_param_grads_views_share_one_flat_buffer():
    p = ParamGrads(10, 3, device="cpu", mesh_slices={"a": (0, 4), "b": (4, 10)})
    p.pos[5, 1] = 2.0; p.nrm[0, 0] = 3.0; p.alpha[2] = 4.0; p.cam_origin[1] = 5.0
    assert p.flat.numel() == 6 * 10 + 3 + 3 and float(p.flat.sum()) == 14.0
    assert float(p.mesh_pos("b")[1, 1]) == 2.0 and p.mesh_pos("a").shape == (4, 3)
    p.zero_(); assert float(p.flat.abs().sum()) == 0


def test_tone_mapping_and_resize():
    x = torch.tensor([[[0.0, 0.0031308, 1.5]]])
    y = to_ldr(x)
    assert float(y[0, 0, 0]) == 0 and abs(float(y[0, 0, 1]) - round(12.92 * 0.0031308 * 255) / 255) < 1e-6 and float(y[0, 0, 2]) == 1
    img = torch.arange(16.0).reshape(4, 4, 1).repeat(1, 1, 3)
    assert resize(img, 2).shape == (2, 2, 3) and abs(float(resize(img, 2)[0, 0, 0]) - 2.5) < 1e-5
