"""GPU parity of the normalisation / resampling kernels vs plain PyTorch fp32 on the same bf16 inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def cl(x):
    return x.bfloat16().contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape", [(2, 32, 20, 28), (3, 256, 7, 9), (2, 64, 33, 17), (1, 24, 40, 40)])
@pytest.mark.parametrize("relu", [False, True])
def test_instance_norm_fwd_bwd(shape, relu):
    from openess_amd import hip
    torch.manual_seed(1)
    x = cl(torch.randn(*shape, device="cuda") * 2 + 0.5)
    res = cl(torch.randn(*shape, device="cuda"))
    g = cl(torch.randn(*shape, device="cuda"))
    for use_res in ((False,) if relu else (False, True)):
        xi = x.clone().requires_grad_(True)
        ri = res.clone().requires_grad_(True)
        y = hip.instance_norm(xi, relu=relu, residual=ri if use_res else None)
        y.backward(g)
        xr = x.float().clone().requires_grad_(True)
        rr = res.float().clone().requires_grad_(True)
        yr = F.instance_norm(xr)
        if use_res:
            yr = yr + rr
        if relu:
            yr = F.relu(yr)
        yr.backward(g.float())
        np.testing.assert_allclose(y.float().detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
        np.testing.assert_allclose(xi.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=3e-2, atol=3e-2 * float(xr.grad.abs().max()))
        if use_res:
            np.testing.assert_allclose(ri.grad.float().cpu().numpy(), rr.grad.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape", [(2, 64, 24, 32), (4, 2048, 5, 6), (2, 256, 13, 11)])
def test_batch_norm_train_matches_torch(shape):
    from openess_amd import engine
    torch.manual_seed(2)
    C = shape[1]
    x = cl(torch.randn(*shape, device="cuda") * 1.5 + 0.3)
    res = cl(torch.randn(*shape, device="cuda"))
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    ref = torch.nn.BatchNorm2d(C).cuda().train()
    ref.load_state_dict(bn.state_dict())
    with torch.no_grad():
        y = engine.batch_norm_act(x, bn, relu=True, residual=res)
        yr = F.relu(ref(x.float()) + res.float())
    np.testing.assert_allclose(y.float().cpu().numpy(), yr.cpu().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), rtol=1e-3, atol=1e-4)
    assert int(bn.num_batches_tracked) == 1


def test_upsample_concat_fwd_bwd():
    from openess_amd import hip
    torch.manual_seed(3)
    x = cl(torch.randn(2, 16, 5, 7, device="cuda")).requires_grad_(True)
    skip = cl(torch.randn(2, 24, 10, 14, device="cuda"))
    y = hip.upsample2x_concat(x, skip)
    ref = torch.cat([F.interpolate(x.detach().float(), scale_factor=2, mode="nearest"), skip.float()], 1)
    assert torch.equal(y.float(), ref)
    g = cl(torch.randn_like(ref))
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode="nearest").backward(g[:, :16].float())
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("C", [64, 128, 256, 512, 192])     # 192: scalar fallback (C/8 lanes is not a power of two)
def test_bilinear_l2norm(C):
    from openess_amd import hip
    torch.manual_seed(4)
    x = cl(torch.randn(2, C, 11, 16, device="cuda"))
    y = hip.bilinear_l2norm(x, 4, True)
    ref = F.normalize(F.interpolate(x.float(), scale_factor=4, mode="bilinear", align_corners=True), p=2, dim=1)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=2e-3)
    y2 = hip.bilinear_l2norm(x, 4, False)
    ref2 = F.interpolate(x.float(), scale_factor=4, mode="bilinear", align_corners=True)
    np.testing.assert_allclose(y2.float().cpu().numpy(), ref2.cpu().numpy(), rtol=1e-2, atol=2e-2)
