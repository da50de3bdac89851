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


@pytest.mark.parametrize("C", [64, 256, 192])              # 192: scalar fallback kernel
def test_bilinear_l2norm_train_fwd_bwd(C):
    """The differentiable teacher head as one node (models/image_model.py:121-143: x4 bilinear, align_corners=True, then
    F.normalize): forward equals the inference kernel bit for bit and torch's fp32 result to bf16 accuracy; the input gradient
    equals torch's autograd through interpolate + normalize (bf16 tolerance) and the two-node path it replaces."""
    from openess_amd import hip
    torch.manual_seed(C)
    x = cl(torch.randn(2, C, 11, 16, device="cuda")).requires_grad_(True)
    y = hip.bilinear_l2norm_train(x, 4)
    assert torch.equal(y.detach(), hip.bilinear_l2norm(x.detach(), 4, True))
    xr = x.detach().float().requires_grad_(True)
    ref = F.normalize(F.interpolate(xr, scale_factor=4, mode="bilinear", align_corners=True), p=2, dim=1)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-2, atol=2e-3)
    g = cl(torch.randn_like(ref).bfloat16())
    y.backward(g)
    ref.backward(g.float())
    a, b = x.grad.float().cpu().numpy(), xr.grad.cpu().numpy()
    assert float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.9995
    np.testing.assert_allclose(a, b, rtol=5e-2, atol=5e-2 * np.abs(b).max())
    x2 = x.detach().clone().requires_grad_(True)
    hip.l2_normalize(hip.bilinear_resize(x2, scale_factor=4, align_corners=True)).backward(g)
    c = x2.grad.float().cpu().numpy()
    assert float((a * c).sum() / (np.linalg.norm(a) * np.linalg.norm(c))) > 0.9995


@pytest.mark.parametrize("case", [
    # B, C, H, W, Ho, Wo, align, dtype
    (2, 256, 7, 10, 110, 160, False, torch.bfloat16),     # DeepLabv3 features: OS16 -> input size (vector path)
    (2, 11, 7, 10, 110, 157, False, torch.float32),       # logits, K = 11 (scalar path), non-integer ratio
    (1, 64, 11, 16, 44, 64, True, torch.bfloat16),        # teacher x4, align_corners=True
    (1, 8, 9, 5, 9, 5, False, torch.float32),             # identity size
    (2, 12, 6, 6, 3, 4, False, torch.float32),            # downsampling (footprints skip input pixels)
    (1, 16, 1, 1, 5, 7, True, torch.float32),             # single input pixel
])
def test_bilinear_resize_fwd_bwd(case):
    """oess_resize_bilinear_nhwc_fwd / _bwd vs F.interpolate + autograd on the same operands (fp32 math in both)."""
    from openess_amd import hip
    B, C, H, W, Ho, Wo, align, dtype = case
    torch.manual_seed(Ho * Wo + C)
    x = torch.randn(B, C, H, W, device="cuda").to(dtype)
    if dtype == torch.bfloat16:
        x = cl(x)
    x.requires_grad_(True)
    y = hip.bilinear_resize(x, size=(Ho, Wo), align_corners=align)
    xr = x.detach().float().requires_grad_(True)
    ref = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=align)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert y.shape == ref.shape and y.dtype == dtype
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=tol, atol=tol)
    g = torch.randn_like(ref)
    y.backward(g.to(dtype))
    ref.backward(g.to(dtype).float())
    # backward tolerance: fp32 sums of <= (2*scale+2)^2 terms in a different order; bf16 rounds the result once
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=5e-2 if dtype == torch.bfloat16 else 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [96, 256, 64])          # 96: generic wave-per-pixel kernel; 256 / 64: 16-byte vector kernels
def test_l2_normalize_fwd_bwd(dtype, C):
    from openess_amd import hip
    torch.manual_seed(9)
    x = torch.randn(2, C, 5, 7, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    x[0, :, 0, 0] = 0                                   # zero vector: eps clamp (y = 0, gradient = g / eps)
    x.requires_grad_(True)
    y = hip.l2_normalize(x)
    xr = x.detach().float().requires_grad_(True)
    ref = F.normalize(xr, p=2, dim=1)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=tol, atol=tol)
    g = torch.randn_like(ref)
    g[0, :, 0, 0] = 0                                   # keep the 1/eps branch finite in the comparison
    y.backward(g.to(dtype))
    ref.backward(g.to(dtype).float())
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=3e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=2e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, False), (False, True)])
def test_batch_norm_train_fwd_bwd(relu, with_res):
    """oess_batchnorm_bwd_nhwc_bf16 + the forward norm kernels vs nn.BatchNorm2d(train) [+ residual] [+ ReLU] in fp32 on
    the same bf16-rounded operands: output, running statistics, dx, dgamma, dbeta, d(residual)."""
    from openess_amd import hip
    torch.manual_seed(3)
    B, C, H, W = 4, 64, 9, 13
    x = cl(torch.randn(B, C, H, W, device="cuda") * 2 + 0.5).requires_grad_(True)
    res = cl(torch.randn(B, C, H, W, device="cuda")).requires_grad_(True) if with_res else None
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C) * 0.2)
    ref_bn = torch.nn.BatchNorm2d(C).cuda().train()
    ref_bn.load_state_dict(bn.state_dict())
    y = hip.batch_norm_train(x, bn, relu=relu, residual=res)
    xr = x.detach().float().requires_grad_(True)
    rr = res.detach().float().requires_grad_(True) if with_res else None
    yr = ref_bn(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref_bn.running_mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1
    g = torch.randn_like(yr)
    # use the HIP output's own ReLU mask for the reference too (values within bf16 rounding of 0 may flip)
    if relu:
        g = g * (y.detach().float() > 0) * (yr.detach() > 0)
    y.backward(g.to(torch.bfloat16))
    yr.backward(g.to(torch.bfloat16).float())
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref_bn.weight.grad.cpu().numpy(), rtol=2e-2, atol=5e-2)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref_bn.bias.grad.cpu().numpy(), rtol=2e-2, atol=5e-2)
    if with_res:
        np.testing.assert_allclose(res.grad.float().cpu().numpy(), rr.grad.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape", [(8, 2048, 28, 40), (2, 64, 7, 5)])
def test_global_avg_pool_fwd_bwd(shape):
    """nn.AdaptiveAvgPool2d(1) of ASPPPooling (models/deeplabv3.py) on the bf16 map: values vs the fp32 mean of the same bf16
    numbers; the gradient is the broadcast quotient (an expanded view: no H x W tensor is materialised)."""
    from openess_amd import hip
    torch.manual_seed(1)
    x = cl(torch.randn(*shape, device="cuda")).requires_grad_(True)
    y = hip.global_avg_pool(x)
    assert y.shape == (shape[0], shape[1], 1, 1) and y.dtype == torch.float32
    ref = x.detach().float().mean(dim=(2, 3), keepdim=True)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    assert gx.stride(2) == 0 and gx.stride(3) == 0
    np.testing.assert_allclose(gx.float().cpu().numpy(), (g / (shape[2] * shape[3])).bfloat16().float().expand(*shape).cpu().numpy(), rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,H,W,frozen", [(2, 11, 37, 53, False), (1, 6, 8, 300, True), (2, 19, 64, 65, False), (1, 32, 5, 7, False)])
def test_linear_probe_matches_conv2d(B, K, H, W, frozen):
    """hip.linear_probe = nn.Conv2d(K, K, 1) on the fp32 logits (reference models/style_networks.py:169-170): forward within
    fp32 re-association of F.conv2d (1e-5 relative), weight / bias / input gradients within 1e-4 relative of the fp64 result,
    and bit-identical between two runs (fixed-order partial rows)."""
    from openess_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(K * 100 + W)
    conv = torch.nn.Conv2d(K, K, 1).to(dev)
    x = torch.randn(B, K, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(not frozen)
    gy = torch.randn(B, K, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    y = hip.linear_probe(x, conv)
    assert y.is_contiguous(memory_format=torch.channels_last) or K == 1
    y.backward(gy)
    got = (y.detach().clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), None if frozen else x.grad.clone())
    conv.zero_grad(); x.grad = None
    y2 = hip.linear_probe(x, conv); y2.backward(gy)
    assert torch.equal(y2, got[0]) and torch.equal(conv.weight.grad, got[1]) and torch.equal(conv.bias.grad, got[2])
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, bd)
    yd.backward(gy.double())

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(got[0], yd.detach()) < 1e-5
    assert rel(got[1], wd.grad) < 1e-4 and rel(got[2], bd.grad) < 1e-4
    if not frozen:
        assert rel(got[3], xd.grad) < 1e-5
    else:
        assert x.grad is None
