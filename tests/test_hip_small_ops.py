"""GPU parity of the small DeepLabv3-path kernels (round 5: they replace ATen max-pool / dropout and the hipBLASLt GEMV + MIOpen
BatchNorm of the ASPP image-pooling branch) against plain PyTorch on the same bf16 values.
Reference lines: models/_resnet.py:124,197 (stem max-pool), models/deeplabv3.py:305-316 (ASPPPooling), :343 (Dropout(0.1))."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 64, 37, 53), (1, 8, 8, 8), (3, 16, 1, 5), (2, 64, 110, 160)])
def test_maxpool3x3s2_forward_backward_equal_aten(shape):
    from openess_amd import hip
    torch.manual_seed(shape[2])
    B, C, H, W = shape
    x = torch.randn(B, H, W, C, device="cuda").bfloat16().permute(0, 3, 1, 2)
    x[0, :, 0, 0] = float("-inf")                          # a window whose first tap is -inf
    if H > 4 and W > 4:
        x[0, 0, 2, 2] = float("nan")                       # NaN wins and propagates
        x[-1, :, 3:5, 3:5] = 1.5                           # ties: the first maximum in row-major order takes the gradient
    xa = x.clone().requires_grad_(True)
    xb = x.float().clone().requires_grad_(True)
    ya = hip.max_pool_3x3s2(xa)
    yb = F.max_pool2d(xb, 3, 2, 1)
    assert ya.shape == yb.shape
    assert torch.equal(torch.nan_to_num(ya.float(), nan=7.0), torch.nan_to_num(yb, nan=7.0))
    g = torch.randn_like(yb).bfloat16()
    ya.backward(g.contiguous(memory_format=torch.channels_last))
    yb.backward(g.float())
    # each input element receives at most four bf16 terms summed in fp32 and rounded once: compare with the fp32 sum rounded to bf16
    np.testing.assert_allclose(xa.grad.float().cpu().numpy(), xb.grad.bfloat16().float().cpu().numpy(), rtol=1e-2, atol=1e-6)
    assert torch.equal(xa.grad != 0, xb.grad != 0)


def test_maxpool_in_the_resnet_stem_is_the_hip_kernel():
    from openess_amd.models import _resnet
    m = _resnet.resnet50().cuda()
    assert isinstance(m.maxpool, _resnet.HipMaxPool2d) and list(m.state_dict().keys())[0] == "conv1.weight"
    x = torch.randn(2, 64, 20, 28, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    assert torch.equal(m.maxpool(x).float(), F.max_pool2d(x.float(), 3, 2, 1))


def test_dropout_mask_rate_scale_and_backward():
    from openess_amd import hip
    torch.manual_seed(11)
    x = (torch.rand(2, 440, 640, 8, device="cuda") + 0.5).bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    y = hip.dropout(x, 0.1, True)
    keep = (y != 0)
    rate = float(keep.float().mean())
    n = x.numel()
    assert abs(rate - 0.9) < 5 * (0.09 / n) ** 0.5 + 2e-5, rate          # Bernoulli(1 - 6554/65536)
    ref = (x.detach().float() / 0.9).bfloat16()
    assert torch.equal(y[keep], ref[keep])
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, keep)                              # the backward pass recomputes the SAME mask
    assert torch.equal(x.grad[keep].float(), torch.full_like(x.grad[keep], 1 / 0.9).float())
    y2 = hip.dropout(x.detach(), 0.1, True)                            # next call: another mask
    assert not torch.equal(y2 != 0, keep)
    assert hip.dropout(x, 0.1, False) is x and hip.dropout(x, 0.0, True) is x
    # the per-channel keep rate is uniform (no structure along the 8-channel groups a thread owns)
    per_c = keep.float().mean(dim=(0, 2, 3))
    assert float((per_c - 0.9).abs().max()) < 3e-3


@pytest.mark.parametrize("B,Cin,Cout,hw", [(8, 2048, 256, (28, 40)), (2, 64, 32, (5, 7)), (3, 128, 256, (4, 4))])
def test_aspp_pool_branch_matches_pytorch(B, Cin, Cout, hw):
    from openess_amd.models.deeplabv3 import ASPPPooling
    torch.manual_seed(B)
    m = ASPPPooling(Cin, Cout).cuda().train()
    with torch.no_grad():
        m[2].weight.uniform_(0.5, 1.5)
        m[2].bias.normal_(0, 0.2)
    ref = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(Cin, Cout, 1, bias=False), nn.BatchNorm2d(Cout), nn.ReLU()).cuda().train()
    ref.load_state_dict(m.state_dict())
    x = torch.randn(B, hw[0], hw[1], Cin, device="cuda").bfloat16().permute(0, 3, 1, 2)
    xa = x.clone().requires_grad_(True)
    xb = x.float().clone().requires_grad_(True)
    ya = m(xa)
    yb = ref(xb).expand(-1, -1, *hw)
    assert ya.shape == yb.shape and ya.dtype == torch.bfloat16
    np.testing.assert_allclose(ya.detach().float().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(m[2].running_mean.cpu().numpy(), ref[2].running_mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m[2].running_var.cpu().numpy(), ref[2].running_var.cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert int(m[2].num_batches_tracked) == 1
    g = torch.randn(B, hw[0], hw[1], Cout, device="cuda").bfloat16().permute(0, 3, 1, 2)
    ya.backward(g)
    yb.backward(g.float())
    for a, b, name in ((m[1].weight.grad, ref[1].weight.grad, "w"), (m[2].weight.grad, ref[2].weight.grad, "gamma"),
                       (m[2].bias.grad, ref[2].bias.grad, "beta")):
        cos = float(F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
        assert cos > 0.9999, (name, cos)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-2, atol=2e-3 * float(b.abs().max()))
    cos = float(F.cosine_similarity(xa.grad.float().flatten().double(), xb.grad.flatten().double(), dim=0))
    assert cos > 0.999, cos
    # the map's gradient is a broadcast (stride 0 over H x W), as the reference's expand backward would sum into
    assert xa.grad.shape == x.shape


@pytest.mark.parametrize("inplanes,planes,hw", [(256, 64, (24, 40)), (1024, 256, (7, 10))])
def test_bottleneck_skip_gradient_rides_in_conv1_dgrad(inplanes, planes, hw, monkeypatch):
    """Bottleneck without downsample (models/_resnet.py:96-114): x feeds conv1 and the residual add.  The residual gradient is
    handed from the conv3+bn3 node to the conv1+bn1 node, whose data-gradient kernel adds it in its epilogue; the result must
    equal autograd's own sum of the two gradients (same rounding points: conv result -> bf16, + skip -> bf16)."""
    from openess_amd.models import _resnet
    torch.manual_seed(5)
    blk = _resnet.Bottleneck(inplanes, planes).cuda().train()
    x0 = torch.randn(2, hw[0], hw[1], inplanes, device="cuda").bfloat16().permute(0, 3, 1, 2)
    g = torch.randn(2, hw[0], hw[1], inplanes, device="cuda").bfloat16().permute(0, 3, 1, 2)
    res = []
    for fold in (True, False):
        if not fold:
            monkeypatch.setattr(_resnet.Bottleneck, "_one_node_path", lambda self, x: False)
        for p in blk.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    for n in res[0][2]:
        assert torch.equal(res[0][2][n], res[1][2][n]), n
    # a block WITH a downsample branch keeps autograd's sum (no hand-over slot)
    blk2 = _resnet.Bottleneck(inplanes, planes, downsample=torch.nn.Sequential(_resnet.conv1x1(inplanes, planes * 4),
                                                                                torch.nn.BatchNorm2d(planes * 4))).cuda().train()
    x = x0.clone().requires_grad_(True)
    blk2(x).backward(g)
    assert x.grad is not None and bool(torch.isfinite(x.grad.float()).all())
