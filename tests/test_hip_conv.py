"""GPU parity of the MFMA implicit-GEMM convolution vs a plain PyTorch fp32 conv2d of the same
bf16-rounded operands.  Tolerance: the kernel accumulates in fp32 and rounds the result to bf16
once (rel 2^-8); fp32 accumulation order differs from the library conv."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_conv(x_nhwc, w, bias, stride, pad, dil):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    y = F.conv2d(x, w.bfloat16().float(), bias, stride=stride, padding=pad, dilation=dil)
    return y.permute(0, 2, 3, 1)


CASES = [
    # B, H, W, Cin, Cout, R, stride, pad, dil
    (2, 20, 24, 16, 40, 3, 1, 1, 1),
    (1, 33, 47, 8, 32, 5, 1, 2, 1),        # E2VID head shape class (5->8 padded channels, Cout=32 tile)
    (2, 30, 42, 32, 64, 5, 2, 2, 1),       # E2VID encoder 5x5 stride 2 (Cout=64 tile)
    (2, 17, 19, 64, 256, 1, 1, 0, 1),      # bottleneck 1x1
    (1, 25, 31, 24, 136, 3, 1, 2, 2),      # dilated 3x3, ragged Cout tile
    (1, 28, 40, 128, 128, 3, 1, 6, 6),     # ASPP rate 6
    (3, 9, 11, 256, 512, 3, 1, 1, 1),      # ConvLSTM gates class
    (1, 64, 96, 8, 64, 7, 2, 3, 1),        # ResNet conv1 (3->8 padded)
    (2, 21, 26, 96, 128, 3, 1, 1, 1),      # Cin % 32 == 0 only: half-slab tap decode, slab halves in different taps
    (1, 19, 23, 32, 72, 5, 1, 2, 2),       # same path, ragged K (25 taps x 32 = 800 -> Kpad 832), dilated
    (1, 110, 160, 256, 512, 3, 1, 1, 1),   # full-width gate-conv class: 138 x 4 = 552 tiles, more than one round of workgroups
    (2, 220, 320, 64, 256, 1, 1, 0, 1),    # short-K 1x1 at scale: BK = 32 x 3 ring kernel, 1100 x 2 tiles
    (1, 55, 80, 512, 1024, 3, 1, 1, 1),    # 35 m-tiles (ragged) x 8 n-tiles through the 64 x 128 tile heuristic
    (2, 28, 40, 2048, 256, 3, 1, 6, 6),    # ASPP branch, rate 6, at the frame2recon_full geometry (440x640 / 16)
    (2, 28, 40, 2048, 256, 3, 1, 12, 12),  # ASPP rate 12: every tap lands inside the 28x40 map for interior pixels
    (2, 28, 40, 2048, 256, 3, 1, 18, 18),  # ASPP rate 18 (models/deeplabv3.py:137-142, else-branch rates 6/12/18)
    (1, 14, 20, 2048, 256, 3, 1, 12, 12),  # 224x320 input: rate 12 still has in-range off-centre taps
    (2, 220, 320, 512, 1024, 1, 1, 0, 1),  # large 1x1 (teacher layer class): 550 x 4 tiles of 256 x 256
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_matches_torch(case):
    from openess_amd import hip
    B, H, W, Cin, Cout, R, stride, pad, dil = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, R, R, device="cuda") / np.sqrt(Cin * R * R)
    b = torch.randn(Cout, device="cuda")
    packed = hip.pack_conv_weight(w)
    y = hip.conv2d_nhwc(x, packed, b, Cout, R, R, stride, pad, dil)
    ref = ref_conv(x, w, b, stride, pad, dil)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    # relu + fp32 output path
    y32 = hip.conv2d_nhwc(x, packed, b, Cout, R, R, stride, pad, dil, relu=True, out_f32=True)
    np.testing.assert_allclose(y32.cpu().numpy(), ref.clamp_min(0).cpu().numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("geom", [(2, 440, 640, 32, 64),      # E2VID encoder 0 at the DSEC size (patches ragged in y: 220 = 27.5 x 8)
                                  (2, 220, 320, 64, 128),     # encoder 1: two channel chunks, two output-channel tiles
                                  (2, 110, 160, 128, 256),    # encoder 2: four chunks (100 weight slabs through the 6-deep ring)
                                  (3, 37, 51, 32, 64),        # odd sizes: ragged patches in x and y, Ho = 19, Wo = 26
                                  (1, 9, 7, 96, 192)])        # map smaller than one patch, three chunks, three channel tiles
def test_conv5x5_stride2_halo_kernel(geom):
    """5x5 / stride 2 / pad 2 (E2VID's encoder ConvLayers, e2vid/model/unet.py): the 2-D input-halo kernel against the fp32
    convolution of the same operands: plain, bias + ReLU into a channel slice of a wider buffer (the ConvLSTM's cat(x, h)
    buffer takes the encoder output that way), bit-repeatable."""
    from openess_amd import hip
    B, H, W, Cin, Cout = geom
    torch.manual_seed(sum(geom))
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 5, 5, device="cuda") / np.sqrt(Cin * 25)
    bias = torch.randn(Cout, device="cuda")
    packed = hip.pack_conv_weight(w)
    ref = ref_conv(x, w, None, 2, 2, 1)
    y = hip.conv2d_nhwc(x, packed, None, Cout, 5, 5, 2, 2, 1)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    Ho, Wo = ref.shape[1], ref.shape[2]
    buf = torch.full((B, Ho, Wo, 2 * Cout + 64), 7.0, device="cuda", dtype=torch.bfloat16)
    y2 = hip.conv2d_nhwc(x, packed, bias, Cout, 5, 5, 2, 2, 1, relu=True, out=buf[..., 64:64 + Cout])
    np.testing.assert_allclose(y2.float().cpu().numpy(), (ref + bias).clamp_min(0).cpu().numpy(), rtol=1e-2, atol=1e-2)
    assert float((buf[..., :64] - 7).abs().max()) == 0 and float((buf[..., 64 + Cout:] - 7).abs().max()) == 0
    assert torch.equal(y, hip.conv2d_nhwc(x, packed, None, Cout, 5, 5, 2, 2, 1))
    # a strided input view (channel slice of a wider tensor) is read in place
    wide = torch.randn(B, H, W, Cin + 32, device="cuda").bfloat16()
    y3 = hip.conv2d_nhwc(wide[..., 32:], packed, None, Cout, 5, 5, 2, 2, 1)
    np.testing.assert_allclose(y3.float().cpu().numpy(), ref_conv(wide[..., 32:], w, None, 2, 2, 1).cpu().numpy(), rtol=1e-2, atol=1e-2)


def test_conv_small_cout_f32_logits_and_residual():
    from openess_amd import hip
    torch.manual_seed(7)
    B, H, W, Cin, Cout = 2, 23, 29, 64, 11
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / 8
    y = hip.conv2d_nhwc(x, hip.pack_conv_weight(w), None, Cout, 1, 1, out_f32=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref_conv(x, w, None, 1, 0, 1).cpu().numpy(), rtol=2e-3, atol=2e-3)
    # residual add + relu (INSResBlock / Bottleneck tail), 3x3
    w3 = torch.randn(64, 64, 3, 3, device="cuda") / 24
    res = torch.randn(B, H, W, 64, device="cuda").bfloat16()
    y = hip.conv2d_nhwc(x, hip.pack_conv_weight(w3), None, 64, 3, 3, 1, 1, 1, relu=True, residual=res)
    ref = (ref_conv(x, w3, None, 1, 1, 1).bfloat16().float() + res.float()).clamp_min(0)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, R, pad: one geometry per kernel that ends in conv_epilogue
    (2, 23, 29, 64, 76, 1, 0),          # 128-row tile, ragged channel tail (Cout % 8 == 4) + residual: per-element tail loads
    (8, 55, 80, 512, 128, 1, 0),        # tile-quantisation rule -> 64-row tiles (550 of them), 4 residual pieces per thread
    (2, 110, 160, 64, 256, 1, 0),       # K <= 256 -> BK = 32 ring kernel, 8 pieces per thread
    (2, 110, 160, 256, 256, 3, 1),      # row-halo 3x3 kernel
    (8, 110, 160, 256, 512, 1, 0),      # >= 400 tiles of 256 x 256: 16 pieces per thread
])
def test_conv_residual_epilogue_on_every_tile_shape(case):
    """out = relu(conv + bias + residual): the residual rows are requested as one batch per thread before the barrier of the
    epilogue (conv_epilogue); every tile shape, a ragged channel tail, rows beyond M, a strided residual view."""
    from openess_amd import hip
    B, H, W, Cin, Cout, R, pad = case
    torch.manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, R, R, device="cuda") / (Cin * R * R) ** 0.5
    bias = torch.randn(Cout, device="cuda")
    Cw = (Cout + 7) // 8 * 8 + 8
    res = torch.randn(B, H, W, Cw, device="cuda").bfloat16()[..., :Cout]         # views: pixel stride (multiple of 8) != Cout
    out = torch.zeros(B, H, W, Cw, device="cuda", dtype=torch.bfloat16)
    y = hip.conv2d_nhwc(x, hip.pack_conv_weight(w), bias, Cout, R, R, 1, pad, 1, relu=True, residual=res, out=out[..., :Cout])
    assert float(out[..., Cout:].abs().max()) == 0.0                               # nothing written past the last channel
    ref = (ref_conv(x, w, bias, 1, pad, 1).bfloat16().float() + res.float()).clamp_min(0)
    err = (y.float() - ref).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.abs().max()) + 2e-2
    assert float(err.mean()) <= 4e-3 * float(ref.abs().mean()) + 1e-4
    y2 = hip.conv2d_nhwc(x, hip.pack_conv_weight(w), bias, Cout, R, R, 1, pad, 1, relu=True, residual=res)
    assert torch.equal(y, y2)


def test_conv_channel_slices_of_concat_buffers():
    """Input read from, and output written into, channel slices of wider NHWC buffers (ConvLSTM cat(x,h))."""
    from openess_amd import hip
    torch.manual_seed(9)
    B, H, W = 2, 14, 18
    buf_in = torch.randn(B, H, W, 96, device="cuda").bfloat16()
    buf_out = torch.zeros(B, H, W, 160, device="cuda").bfloat16()
    w = torch.randn(64, 32, 3, 3, device="cuda") / 17
    x = buf_in[..., 32:64]
    out = buf_out[..., 96:160]
    hip.conv2d_nhwc(x, hip.pack_conv_weight(w), None, 64, 3, 3, 1, 1, 1, out=out)
    ref = ref_conv(x, w, None, 1, 1, 1)
    np.testing.assert_allclose(buf_out[..., 96:160].float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    assert float(buf_out[..., :96].abs().max()) == 0.0


def test_conv_dgrad_operator():
    """flip packing: dX of a stride-1 conv computed with the forward kernel == autograd's input gradient."""
    from openess_amd import hip
    torch.manual_seed(11)
    B, H, W, Cin, Cout, R, pad, dil = 2, 16, 21, 32, 48, 3, 2, 2
    x = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Cout, Cin, R, R, device="cuda") / 17).bfloat16().float()
    y = F.conv2d(x, w, padding=pad, dilation=dil)
    gy = torch.randn_like(y).bfloat16().float()
    y.backward(gy)
    gy_nhwc = gy.permute(0, 2, 3, 1).contiguous().bfloat16()
    gx = hip.conv2d_nhwc(gy_nhwc, hip.pack_conv_weight(w, flip=True), None, Cin, R, R, 1, dil * (R - 1) - pad, dil,
                         out_f32=True)
    np.testing.assert_allclose(gx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.cpu().numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, R, stride, pad, dil
    (2, 12, 20, 16, 24, 3, 1, 1, 1),
    (2, 9, 70, 64, 32, 3, 1, 1, 1),        # Wo > 64: two pixel chunks per row, Cout < tile
    (1, 11, 13, 256, 256, 3, 1, 1, 1),
    (2, 10, 12, 32, 256, 1, 1, 0, 1),      # 1x1 head
    (2, 16, 24, 64, 64, 3, 2, 1, 1),       # stride 2
    (2, 17, 70, 72, 136, 3, 2, 1, 1),      # stride 2 on the 128-row LDS-DMA form: ragged Cout / Cin / pixel chunks, two chunks per row
    (8, 55, 80, 256, 256, 3, 1, 1, 1),     # the decoder's 55x80 layers at full size (10 per step)
    (2, 330, 320, 64, 32, 3, 1, 1, 1),     # narrow full-resolution layer: whole-gradient-per-workgroup kernel (X halo), TMV = 32
    (1, 520, 400, 128, 64, 3, 1, 1, 1),    # same kernel, TMV = 64, two 64-channel chunks, ragged last K-step per row (400 = 6 x 64 + 16)
    (3, 300, 250, 64, 24, 3, 1, 1, 1),     # Cout not a multiple of 32 (rows 24..31 of the tile stay zero), W % 64 != 0
    (2, 400, 130, 192, 48, 3, 1, 1, 1),    # three 64-channel chunks, Cout 48 on the 64-row tile, a 2-pixel-wide last strip
    (1, 7, 15000, 64, 32, 3, 1, 1, 1),     # very wide, 7 rows: more strips than workgroup slots, row ranges of one or two rows
    (1, 14, 18, 48, 136, 3, 1, 2, 2),      # dilation, ragged tiles
    (2, 8, 130, 8, 16, 5, 1, 2, 1),
    (1, 220, 320, 64, 32, 3, 1, 1, 1),     # decoder layer at scale: 70 400 pixels reduced by the full split-K fan-out
    (1, 110, 160, 128, 256, 3, 1, 1, 1),   # several (Cout, taps*Cin) tiles x split-K
    (2, 28, 40, 2048, 256, 3, 1, 12, 12),  # ASPP rate 12 weight gradient at the full-size 28x40 map
    (2, 28, 40, 2048, 256, 3, 1, 18, 18),  # ASPP rate 18
])
def test_conv_wgrad_matches_autograd(case):
    from openess_amd import hip
    B, H, W, Cin, Cout, R, stride, pad, dil = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.zeros(Cout, Cin, R, R, device="cuda", requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, stride, pad, dil)
    gy = torch.randn_like(y).bfloat16()
    y.backward(gy.float())
    dw = hip.conv2d_wgrad(x, gy.permute(0, 2, 3, 1).contiguous(), Cout, Cin, R, R, stride, pad, dil)
    ref = w.grad
    err = float((dw - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, err
    # input with zero-padded extra channels (Cin_x > Cin), as the 5->8 / 3->8 padded stems have
    if Cin % 16 == 8:
        xp = torch.cat([x, torch.zeros(B, H, W, 8, device="cuda", dtype=torch.bfloat16)], -1)
        dw2 = hip.conv2d_wgrad(xp, gy.permute(0, 2, 3, 1).contiguous(), Cout, Cin, R, R, stride, pad, dil)
        assert float((dw2 - ref).abs().max() / ref.abs().max()) < 2e-3


def test_conv_256_tile_kernel_stats_residual_and_ragged_rows():
    """The 256 x 256-tile path of large 1x1 layers: BatchNorm partial sums (a 256-row tile fills stats row 2t and zeroes row
    2t + 1, so the per-128-row table sums to the column sums whichever kernel ran), residual + ReLU epilogue, and a pixel count
    that is not a multiple of 256."""
    from openess_amd import hip
    torch.manual_seed(9)
    B, H, W, Cin, Cout = 2, 219, 321, 256, 512          # M = 140 598 = 549 x 256 + 54
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / np.sqrt(Cin)
    packed = hip.pack_conv_weight(w)
    ref = ref_conv(x, w, None, 1, 0, 1).reshape(-1, Cout)
    tiles = (B * H * W + 127) // 128
    part = torch.full((tiles, 2, Cout), float("nan"), device="cuda")
    y = hip.conv2d_nhwc(x, packed, None, Cout, 1, 1, 1, 0, 1, tile_stats=part)
    assert torch.isfinite(part).all()
    np.testing.assert_allclose(part[:, 0].double().sum(0).cpu().numpy(), ref.double().sum(0).cpu().numpy(), rtol=2e-3, atol=2.0)
    np.testing.assert_allclose(part[:, 1].double().sum(0).cpu().numpy(), (ref.double() ** 2).sum(0).cpu().numpy(), rtol=2e-3)
    np.testing.assert_allclose(y.float().reshape(-1, Cout).cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    res = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
    bias = torch.randn(Cout, device="cuda")
    y2 = hip.conv2d_nhwc(x, packed, bias, Cout, 1, 1, 1, 0, 1, relu=True, residual=res)
    ref2 = (ref + bias + res.float().reshape(-1, Cout)).clamp_min(0)
    np.testing.assert_allclose(y2.float().reshape(-1, Cout).cpu().numpy(), ref2.cpu().numpy(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("dil", [6, 12, 18])
def test_conv_splitk_aspp_full_size_equals_one_pass(dil):
    """Split-K form of the small-M / long-K layers (DeepLabv3's ASPP at the BASELINE geometry: B = 8, 28 x 40 x 2048 -> 256,
    3 x 3 dilated; M = 8 960, K = 18 432): fp32 K-slices + fixed-order reduce kernel against the one-pass kernel and the fp32
    reference -- output, BatchNorm tile statistics, residual + ReLU tail, channel-slice destination, fp32 logits form."""
    from openess_amd import _lib, hip
    torch.manual_seed(20 + dil)
    B, H, W, Cin, Cout = 8, 28, 40, 2048, 256
    assert _lib.load().oess_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, dil, dil, 1, 0) > 0      # the rule takes it
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / np.sqrt(Cin * 9)
    packed = hip.pack_conv_weight(w)
    ref = ref_conv(x, w, None, 1, dil, dil).reshape(-1, Cout)
    tiles = (B * H * W + 127) // 128
    p1 = torch.full((tiles, 2, Cout), float("nan"), device="cuda")
    p2 = torch.full((tiles, 2, Cout), float("nan"), device="cuda")
    buf = torch.zeros(B, H, W, 5 * Cout, device="cuda", dtype=torch.bfloat16)          # ASPP concat buffer: branch 2's slice
    y_split = hip.conv2d_nhwc(x, packed, None, Cout, 3, 3, 1, dil, dil, tile_stats=p1, out=buf[..., 2 * Cout:3 * Cout])
    y_one = hip.conv2d_nhwc(x, packed, None, Cout, 3, 3, 1, dil, dil, tile_stats=p2, allow_splitk=False)
    assert float(buf[..., :2 * Cout].abs().max()) == 0 and float(buf[..., 3 * Cout:].abs().max()) == 0
    np.testing.assert_allclose(y_split.float().reshape(-1, Cout).cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    assert float((y_split.float() - y_one.float()).abs().max()) <= 2e-2          # same sums, different fp32 association
    np.testing.assert_allclose(p1.cpu().numpy(), p2.cpu().numpy(), rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(p1[:, 0].double().sum(0).cpu().numpy(), ref.double().sum(0).cpu().numpy(), rtol=2e-3, atol=1.0)
    np.testing.assert_allclose(p1[:, 1].double().sum(0).cpu().numpy(), (ref.double() ** 2).sum(0).cpu().numpy(), rtol=2e-3)
    res = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
    bias = torch.randn(Cout, device="cuda")
    y2 = hip.conv2d_nhwc(x, packed, bias, Cout, 3, 3, 1, dil, dil, relu=True, residual=res)
    ref2 = (ref + bias + res.float().reshape(-1, Cout)).clamp_min(0)
    np.testing.assert_allclose(y2.float().reshape(-1, Cout).cpu().numpy(), ref2.cpu().numpy(), rtol=1e-2, atol=2e-2)
    y3 = hip.conv2d_nhwc(x, packed, bias, Cout, 3, 3, 1, dil, dil, out_f32=True)
    np.testing.assert_allclose(y3.reshape(-1, Cout).cpu().numpy(), (ref + bias).cpu().numpy(), rtol=2e-3, atol=2e-3)
    # bit-repeatable: the slices are added in a fixed order
    y_again = hip.conv2d_nhwc(x, packed, None, Cout, 3, 3, 1, dil, dil)
    assert torch.equal(y_again, hip.conv2d_nhwc(x, packed, None, Cout, 3, 3, 1, dil, dil))


@pytest.mark.parametrize("shape", [(8, 28, 40, 1024, 256, 1), (8, 28, 40, 512, 2048, 1), (8, 55, 80, 128, 512, 1), (2, 13, 17, 64, 64, 3)])
@pytest.mark.parametrize("tail", ["plain", "relu", "res_relu", "slice"])
def test_conv_bn_small_map_one_launch_matches_batchnorm(shape, tail):
    """conv + train-mode BatchNorm [+ residual] [+ ReLU] on a small map: statistics and apply in ONE launch
    (oess_norm_tile_stats_apply_nhwc_bf16) against nn.BatchNorm2d on the fp32 conv of the same operands, incl. running
    statistics, a channel-slice destination (ASPP concat buffer) and bit-repeatability."""
    from openess_amd import hip
    B, H, W, Cin, Cout, k = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
    packed = hip.pack_conv_weight(w)
    bn = torch.nn.BatchNorm2d(Cout).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    refbn = torch.nn.BatchNorm2d(Cout).cuda().train()
    refbn.load_state_dict(bn.state_dict())
    res = torch.randn(B, H, W, Cout, device="cuda").bfloat16() if tail == "res_relu" else None
    out = None
    if tail == "slice":
        buf = torch.zeros(B, H, W, 3 * Cout, device="cuda", dtype=torch.bfloat16)
        out = buf[..., Cout:2 * Cout]
    relu = tail in ("relu", "res_relu", "slice")
    with torch.no_grad():
        y = hip.conv_bn_train_nhwc(x, packed, Cout, k, k, 1, k // 2, 1, bn, relu=relu, residual=res, out=out)
        yr = refbn(ref_conv(x, w, None, 1, k // 2, 1).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        if res is not None:
            yr = yr + res.float()
        if relu:
            yr = yr.clamp_min(0)
    np.testing.assert_allclose(y.float().cpu().numpy(), yr.cpu().numpy(), rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), refbn.running_mean.cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), refbn.running_var.cpu().numpy(), rtol=2e-3, atol=1e-4)
    if tail == "slice":
        assert float(buf[..., :Cout].abs().max()) == 0 and float(buf[..., 2 * Cout:].abs().max()) == 0
    with torch.no_grad():
        y2 = hip.conv_bn_train_nhwc(x, packed, Cout, k, k, 1, k // 2, 1, bn, relu=relu, residual=res)
        y3 = hip.conv_bn_train_nhwc(x, packed, Cout, k, k, 1, k // 2, 1, bn, relu=relu, residual=res)
    assert torch.equal(y2, y3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cx,C", [(2, 9, 11, 32, 32), (1, 13, 10, 64, 64), (3, 8, 8, 96, 32),
                                          (1, 110, 160, 128, 128)])     # one DSEC level at full width: 552 tiles > 512 slots (second round, ragged last m-tile)
def test_convlstm_fused_step_matches_torch(B, H, W, Cx, C):
    """oess_convlstm_fused_bf16 (Gates conv + cell update in one kernel, gate-interleaved weights, transposed MFMA)
    against a plain fp32 restatement of ConvLSTM.forward (e2vid/model/submodules.py:199-214), three recurrent steps
    with ping-pong cat(x, h) buffers; also against the unfused conv + gate-kernel pair."""
    import torch
    import torch.nn.functional as F
    from openess_amd import hip
    torch.manual_seed(B * 100 + C)
    dev = "cuda"
    w = torch.randn(4 * C, Cx + C, 3, 3, device=dev) * 0.05
    bias = torch.randn(4 * C, device=dev) * 0.1
    packed_f = hip.pack_conv_weight(w, flip=2)
    packed_u = hip.pack_conv_weight(w)
    wq = w.bfloat16().float()
    xh = [torch.zeros(B, H, W, Cx + C, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    xh_u = torch.zeros(B, H, W, Cx + C, device=dev, dtype=torch.bfloat16)
    cell = torch.empty(B, H, W, C, device=dev)
    cell_u = torch.empty(B, H, W, C, device=dev)
    h_ref = torch.zeros(B, C, H, W, device=dev)
    c_ref = torch.zeros(B, C, H, W, device=dev)
    cur = 0
    for step in range(3):
        x = torch.randn(B, H, W, Cx, device=dev).bfloat16()
        xh[cur][..., :Cx] = x
        xh_u[..., :Cx] = x
        hip.convlstm_fused(xh[cur], packed_f, bias, cell, xh[1 - cur][..., Cx:], 3, 1, prev_cell_is_zero=(step == 0))
        cur = 1 - cur
        gates_u = hip.conv2d_nhwc(xh_u, packed_u, bias, 4 * C, 3, 3, 1, 1, 1)
        hip.convlstm_gates(gates_u, cell_u, xh_u[..., Cx:], prev_cell_is_zero=(step == 0))
        # reference: bf16-rounded operands (what the kernel multiplies), fp32 everywhere else
        stacked = torch.cat([x.float().permute(0, 3, 1, 2), h_ref.bfloat16().float()], 1)
        g = F.conv2d(stacked, wq, bias, padding=1)
        gi, gr, go, gc = g.chunk(4, 1)
        c_ref = torch.sigmoid(gr) * c_ref + torch.sigmoid(gi) * torch.tanh(gc)
        h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
        h_hip = xh[cur][..., Cx:].float().permute(0, 3, 1, 2)
        c_hip = cell.permute(0, 3, 1, 2)
        # tolerance: h is stored in bf16 (2^-8 relative), c accumulates the bf16 rounding of h_prev over the steps
        assert torch.allclose(c_hip, c_ref, atol=2e-2, rtol=2e-2), (step, float((c_hip - c_ref).abs().max()))
        assert torch.allclose(h_hip, h_ref, atol=2e-2, rtol=2e-2), (step, float((h_hip - h_ref).abs().max()))
        h_ref = h_hip.clone()          # both paths continue from the same (bf16) state: no drift in the comparison
        assert torch.allclose(cell, cell_u, atol=3e-2, rtol=3e-2)
        assert torch.allclose(xh[cur][..., Cx:].float(), xh_u[..., Cx:].float(), atol=3e-2, rtol=3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("geoms", [
    # (B, H, W, Cx = C): the three E2VID levels (scaled down), tile counts 30 x 2 / 8 x 4 / 2 x 8 (not all multiples of 8)
    [(1, 48, 80, 64), (1, 24, 40, 128), (1, 12, 20, 256)],
    [(2, 24, 40, 128), (2, 12, 20, 256)],                        # two problems (the drain stage of the skewed schedule)
    [(1, 13, 11, 64), (1, 7, 9, 64), (3, 5, 6, 128)],            # ragged last tiles, 3 + 1 + 2 tiles
    [(1, 9, 11, 64), (1, 13, 10, 32)],                           # the zero-state problem has Cin = 32: not the row-halo kernel's -> launched one by one
    [(8, 220, 320, 64), (8, 110, 160, 128), (8, 55, 80, 256)],   # the BASELINE size (B = 8, 440 x 640 crop): 256-row grouped tiles == 128-row launches, bit for bit
])
def test_convlstm_fused_group_equals_separate_launches(geoms):
    """oess_convlstm_fused_group_bf16: n independent ConvLSTM steps in one launch give bit-identical hidden and cell states to n
    oess_convlstm_fused_bf16 launches (same tiles, same arithmetic), with zero and non-zero previous cells mixed; outputs of one
    problem overlapping another's buffers are rejected."""
    import torch
    from openess_amd import hip
    torch.manual_seed(len(geoms) * 7 + geoms[0][1])
    dev = "cuda"
    probs_a, probs_b, keep = [], [], []
    for i, (B, H, W, C) in enumerate(geoms):
        fresh = (i == 1)
        w = torch.randn(4 * C, 2 * C, 3, 3, device=dev) * (0.3 / C ** 0.5)
        bias = torch.randn(4 * C, device=dev) * 0.1
        packed = hip.pack_conv_weight(w[:, :C] if fresh else w, flip=2)
        xh = (torch.randn(B, H, W, 2 * C, device=dev) * 0.5).bfloat16()
        cell0 = torch.randn(B, H, W, C, device=dev)
        for probs in (probs_a, probs_b):
            cell = cell0.clone()
            h = torch.full((B, H, W, 2 * C), 7.0, device=dev, dtype=torch.bfloat16)
            probs.append((xh[..., :C] if fresh else xh, packed, bias, cell, h[..., C:], 3, 1, fresh))
            keep.append(h)
    for pr in probs_a:
        hip.convlstm_fused(*pr)
    out = hip.convlstm_fused_group(probs_b)
    assert len(out) == len(geoms)
    for pa, pb in zip(probs_a, probs_b):
        assert torch.equal(pa[3], pb[3])
        assert torch.equal(pa[4], pb[4])
        assert float(pb[4].float().abs().max()) < 1.01           # every hidden value written (|h| <= 1)
    # overlapping outputs: a second problem that updates problem 0's cell (its hidden output is another buffer)
    B, H, W, C = geoms[0]
    clash = [probs_b[0], probs_b[0][:3] + (probs_b[0][3], keep[0][..., C:]) + probs_b[0][5:]]
    with pytest.raises(RuntimeError):
        hip.convlstm_fused_group(clash)
    with pytest.raises(ValueError):
        hip.convlstm_fused_group([])


@pytest.mark.gpu
@pytest.mark.parametrize("geoms", [
    [(1, 48, 80, 64), (1, 24, 40, 128), (1, 16, 20, 256)],      # the three E2VID levels scaled down (a tile must not span more rows than the image has)
    [(2, 24, 40, 128), (2, 16, 20, 256)],
    [(3, 16, 24, 64)],                                          # 4.5 tiles: ragged last tile, tiles that run across image borders
    [(1, 20, 33, 64), (2, 18, 27, 64), (1, 16, 41, 128)],       # odd widths: every tile starts mid-row
    [(8, 220, 320, 64), (8, 110, 160, 128), (8, 55, 80, 256)],  # the BASELINE size (B = 8, 440 x 640 crop)
])
def test_convlstm_w128_group_matches_fused_launches(geoms):
    """oess_convlstm_w128_group_bf16 (persistent 128 x 128-wave-tile kernel, w128-tiled cell state) against oess_convlstm_fused_bf16 on
    [B,H,W,C] cells, three recurrent steps with ping-pong cat(x, h) buffers and the cell updated in place: same MFMA products in the
    same order, the cell update reassociated (bias inside the exponent's FMA) -> fp32-rounding agreement of the cell, one bf16 ulp on
    the hidden state.  Also the relayout round trip."""
    import torch
    from openess_amd import hip
    torch.manual_seed(len(geoms) * 11 + geoms[0][2])
    dev = "cuda"
    st = []
    for i, (B, H, W, C) in enumerate(geoms):
        w = torch.randn(4 * C, 2 * C, 3, 3, device=dev) * (0.3 / C ** 0.5)
        bias = torch.randn(4 * C, device=dev) * 0.1
        st.append(dict(B=B, H=H, W=W, C=C, packed=hip.pack_conv_weight(w, flip=2), packed_x=hip.pack_conv_weight(w[:, :C], flip=2), bias=bias,
                       xh_a=[torch.zeros(B, H, W, 2 * C, device=dev, dtype=torch.bfloat16) for _ in range(2)],
                       xh_b=[torch.zeros(B, H, W, 2 * C, device=dev, dtype=torch.bfloat16) for _ in range(2)],
                       cell_a=torch.empty(B, H, W, C, device=dev),
                       cell_b=torch.full((hip.convlstm_w128_cell_elems(B * H * W, C),), 3.0, device=dev), cur=0))
    for s_ in st:                                              # relayout round trip on random data
        c = torch.randn(s_["B"] * s_["H"] * s_["W"], s_["C"], device=dev)
        t = hip.convlstm_w128_cell_relayout(c, c.shape[0], s_["C"], True)
        assert torch.equal(hip.convlstm_w128_cell_relayout(t, c.shape[0], s_["C"], False).reshape(c.shape), c)
    for step in range(3):
        probs_a, probs_b = [], []
        for i, s_ in enumerate(st):
            C, cur = s_["C"], s_["cur"]
            x = (torch.randn(s_["B"], s_["H"], s_["W"], C, device=dev) * 0.5).bfloat16()
            fresh = step == 0
            for xh in (s_["xh_a"], s_["xh_b"]):
                xh[cur][..., :C] = x
            for xh, cell, probs in ((s_["xh_a"], s_["cell_a"], probs_a), (s_["xh_b"], s_["cell_b"], probs_b)):
                probs.append((xh[cur][..., :C] if fresh else xh[cur], s_["packed_x"] if fresh else s_["packed"], s_["bias"], cell,
                              xh[1 - cur][..., C:], 3, 1, fresh))
        for pr in probs_a:
            hip.convlstm_fused(*pr)
        assert hip.convlstm_w128_group(probs_b)
        for s_ in st:
            C, cur = s_["C"], s_["cur"]
            px = s_["B"] * s_["H"] * s_["W"]
            cb = hip.convlstm_w128_cell_relayout(s_["cell_b"], px, C, False).reshape(s_["cell_a"].shape)
            ha, hb = s_["xh_a"][1 - cur][..., C:].float(), s_["xh_b"][1 - cur][..., C:].float()
            assert torch.allclose(cb, s_["cell_a"], atol=2e-5, rtol=1e-5), (step, float((cb - s_["cell_a"]).abs().max()))
            assert float((ha - hb).abs().max()) <= 2.0 ** -7, (step, float((ha - hb).abs().max()))
            assert float((ha - hb).abs().mean()) < 2e-4
            # both paths continue from the SAME hidden state (a one-ulp difference would otherwise grow through the recurrence)
            s_["xh_b"][1 - cur][..., C:] = s_["xh_a"][1 - cur][..., C:]
            s_["cell_b"].copy_(hip.convlstm_w128_cell_relayout(s_["cell_a"].reshape(px, C), px, C, True))
            s_["cur"] = 1 - cur
    # not taken: hidden size not a multiple of 64, or a cell buffer in the wrong size -> False, nothing launched
    B, H, W, C = 1, 16, 24, 32
    w = torch.randn(4 * C, 2 * C, 3, 3, device=dev)
    xh = torch.zeros(B, H, W, 2 * C, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(B, H, W, 2 * C, device=dev, dtype=torch.bfloat16)
    assert hip.convlstm_w128_group([(xh, hip.pack_conv_weight(w, flip=2), None, torch.empty(B, H, W, C, device=dev), out[..., C:], 3, 1, False)]) is False


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,in_extra,out_extra", [
    (1, 256, 256, 256, 1024, 0, 0),        # 1 024 tiles, 4 K-slabs
    (1, 250, 263, 320, 1024, 64, 0),       # ragged last m-tile, 5 K-slabs (the six-fold unrolled loop leaves early), input a channel slice
    (2, 128, 257, 1024, 1024, 0, 256),     # 16 K-slabs, output a channel slice of a wider buffer
    (8, 110, 160, 512, 2048, 0, 0),        # a teacher layer at the BASELINE size
])
def test_conv1x1_w128_equals_the_256_tile_kernel(B, H, W, Cin, Cout, in_extra, out_extra, monkeypatch):
    """conv1x1_w128_kernel (conv_w128_gemm.h: persistent workgroups, 128 x 128 wave tiles, raw bf16 result + BatchNorm tile statistics)
    against conv_fwd_dma_kernel<256, 256> on the same call (OESS_W128_GEMM=0): the same products in the same order -> identical bf16
    outputs; the statistics are sums of the same stored values in another order -> equal to fp32 rounding, totals to 1e-5."""
    import torch
    from openess_amd import hip
    torch.manual_seed(Cin + Cout)
    dev = "cuda"
    xb = (torch.randn(B, H, W, Cin + in_extra, device=dev) * 0.5).bfloat16()
    x = xb[..., in_extra // 2: in_extra // 2 + Cin] if in_extra else xb
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * (1.0 / Cin ** 0.5)
    packed = hip.pack_conv_weight(w)
    M = B * H * W
    tiles = (M + 127) // 128
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OESS_W128_GEMM", mode)
        ob = torch.full((B, H, W, Cout + out_extra), 3.0, device=dev, dtype=torch.bfloat16)
        out = ob[..., out_extra // 2: out_extra // 2 + Cout] if out_extra else ob
        part = torch.full((tiles, 2, Cout), 7.0, device=dev)
        hip.conv2d_nhwc(x, packed, None, Cout, 1, 1, 1, 0, 1, out=out, tile_stats=part)
        res[mode] = (ob.clone(), part.clone())
    assert torch.equal(res["0"][0], res["1"][0])
    if out_extra:
        assert float((res["1"][0][..., :out_extra // 2].float() - 3.0).abs().max()) == 0           # neighbours of the slice untouched
    t0, t1 = res["0"][1].double().sum(0), res["1"][1].double().sum(0)
    assert torch.allclose(t0, t1, rtol=1e-5, atol=1e-3), float((t0 - t1).abs().max())
    # per-128-row statistics of the new kernel against the stored tensor itself
    y = (res["1"][0][..., out_extra // 2: out_extra // 2 + Cout] if out_extra else res["1"][0]).reshape(M, Cout).float()
    pad = tiles * 128 - M
    yp = torch.cat([y, torch.zeros(pad, Cout, device=dev)]) if pad else y
    s1 = yp.reshape(tiles, 128, Cout).double().sum(1)
    s2 = (yp.reshape(tiles, 128, Cout).double() ** 2).sum(1)
    assert torch.allclose(res["1"][1][:, 0].double(), s1, rtol=1e-4, atol=1e-2)
    assert torch.allclose(res["1"][1][:, 1].double(), s2, rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("relu", [False, True])
def test_conv1x1_w128_bias_relu_epilogue(relu, monkeypatch):
    """The bias (+ ReLU) epilogue of conv1x1_w128_kernel (the teacher's 2048 -> 256 decoder layer) against conv_fwd_dma_kernel<256, 256>:
    bias added to the fp32 accumulators, then the activation, then ONE rounding -- identical bits; and against fp32 PyTorch."""
    import torch
    import torch.nn.functional as F
    from openess_amd import hip
    torch.manual_seed(5)
    dev = "cuda"
    B, H, W, Cin, Cout = 2, 257, 256, 384, 512                    # ragged last m-tile, 6 K-slabs
    x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * (1.0 / Cin ** 0.5)
    bias = torch.randn(Cout, device=dev)
    packed = hip.pack_conv_weight(w)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OESS_W128_GEMM", mode)
        res[mode] = hip.conv2d_nhwc(x, packed, bias, Cout, 1, 1, 1, 0, 1, relu=relu).clone()
    assert torch.equal(res["0"], res["1"])
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), bias)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    assert float((res["1"].float() - ref).abs().max()) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,dil,in_extra,out_extra", [
    (1, 256, 256, 128, 512, 1, 0, 0),       # 512 tiles, 6 macro steps
    (2, 131, 250, 192, 512, 3, 64, 0),      # ragged last tile, tiles that start mid-row and cross image borders, 9 macro steps (odd), dilation 3, input slice
    (1, 200, 331, 64, 512, 2, 0, 256),      # odd width, 3 macro steps, output a channel slice
    (8, 110, 160, 256, 256, 2, 0, 0),       # teacher layer3 conv2 at the BASELINE size
    (8, 110, 160, 512, 512, 4, 0, 0),       # teacher layer4 conv2
])
def test_conv3x3_w128_equals_the_row_halo_kernel(B, H, W, Cin, Cout, dil, in_extra, out_extra, monkeypatch):
    """conv3x3_w128_kernel (conv3x3_w128.h) against conv3x3_halo_kernel<0> on the same call (OESS_W128_CONV3=0): identical bf16 outputs
    (same products, same order), BatchNorm tile statistics equal to fp32 rounding."""
    import torch
    from openess_amd import hip
    torch.manual_seed(Cin + Cout + dil)
    dev = "cuda"
    xb = (torch.randn(B, H, W, Cin + in_extra, device=dev) * 0.5).bfloat16()
    x = xb[..., in_extra // 2: in_extra // 2 + Cin] if in_extra else xb
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (1.0 / (9 * Cin) ** 0.5)
    packed = hip.pack_conv_weight(w)
    M = B * H * W
    tiles = (M + 127) // 128
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OESS_W128_CONV3", mode)
        ob = torch.full((B, H, W, Cout + out_extra), 3.0, device=dev, dtype=torch.bfloat16)
        out = ob[..., out_extra // 2: out_extra // 2 + Cout] if out_extra else ob
        part = torch.full((tiles, 2, Cout), 7.0, device=dev)
        hip.conv2d_nhwc(x, packed, None, Cout, 3, 3, 1, dil, dil, out=out, tile_stats=part)
        res[mode] = (ob.clone(), part.clone())
    assert torch.equal(res["0"][0], res["1"][0]), float((res["0"][0].float() - res["1"][0].float()).abs().max())
    t0, t1 = res["0"][1].double().sum(0), res["1"][1].double().sum(0)
    assert torch.allclose(t0, t1, rtol=1e-5, atol=1e-3), float((t0 - t1).abs().max())
    y = (res["1"][0][..., out_extra // 2: out_extra // 2 + Cout] if out_extra else res["1"][0]).reshape(M, Cout).float()
    pad = tiles * 128 - M
    yp = torch.cat([y, torch.zeros(pad, Cout, device=dev)]) if pad else y
    assert torch.allclose(res["1"][1][:, 0].double(), yp.reshape(tiles, 128, Cout).double().sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(res["1"][1][:, 1].double(), (yp.reshape(tiles, 128, Cout).double() ** 2).sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
def test_convlstm_fused_rejects_aliasing_and_bad_shapes():
    import torch
    from openess_amd import hip
    dev = "cuda"
    C, Cx = 32, 32
    w = torch.randn(4 * C, Cx + C, 3, 3, device=dev) * 0.05
    packed = hip.pack_conv_weight(w, flip=2)
    xh = torch.zeros(1, 8, 8, Cx + C, device=dev, dtype=torch.bfloat16)
    cell = torch.zeros(1, 8, 8, C, device=dev)
    with pytest.raises(RuntimeError):          # hidden output inside the conv input: neighbouring tiles would race
        hip.convlstm_fused(xh, packed, None, cell, xh[..., Cx:], 3, 1)
    C2 = 16                                       # hidden size not a multiple of 32
    w2 = torch.randn(4 * C2, Cx + C2, 3, 3, device=dev) * 0.05
    xh2 = torch.zeros(1, 8, 8, Cx + C2, device=dev, dtype=torch.bfloat16)
    out2 = torch.zeros(1, 8, 8, C2, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        hip.convlstm_fused(xh2, hip.pack_conv_weight(w2, flip=2), None, torch.zeros(1, 8, 8, C2, device=dev), out2, 3, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cout,relu", [(2, 70, 130, 32, True), (1, 9, 65, 12, False), (3, 8, 64, 32, True),
                                               (4, 130, 650, 32, True),      # 748 tiles > 512 persistent workgroups: 1 or 2 tiles each
                                               (3, 440, 640, 32, True)])     # the E2VID head at full DSEC size: 1650 tiles, 3 or 4 each
def test_conv_small_cin_halo_kernel(B, H, W, Cout, relu):
    """Cin = 8, 5x5, stride 1 (E2VID head): the LDS halo-tile kernel (conv_smallcin_kernel), several tiles with ragged
    edges, bias + ReLU in bf16, output written into a channel slice of a wider buffer."""
    from openess_amd import hip
    torch.manual_seed(H * W + Cout)
    x = torch.randn(B, H, W, 8, device="cuda").bfloat16()
    x[..., 5:] = 0                                   # 5 event bins padded to 8 channels
    w = torch.randn(Cout, 8, 5, 5, device="cuda") / np.sqrt(5 * 25)
    b = torch.randn(Cout, device="cuda")
    packed = hip.pack_conv_weight(w)
    wide = torch.full((B, H, W, Cout + 16), 7.0, device="cuda", dtype=torch.bfloat16)
    y = hip.conv2d_nhwc(x, packed, b, Cout, 5, 5, 1, 2, 1, relu=relu, out=wide[..., 8:8 + Cout])
    ref = ref_conv(x, w, b, 1, 2, 1)
    if relu:
        ref = ref.clamp_min(0)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    assert float((wide[..., :8].float() - 7).abs().max()) == 0 and float((wide[..., 8 + Cout:].float() - 7).abs().max()) == 0


def test_conv_gelu_epilogue_matches_exact_erf():
    """relu = 2 selects nn.GELU() (exact erf form) in the epilogue; the kernel evaluates erf by Abramowitz-Stegun 7.1.26
    (|error| <= 1.5e-7): fp32 output within 2e-6 of F.gelu on the reference conv, bf16 output within bf16 rounding -- on the
    128 x 128 and on the 256 x 256 tile path (ViT fc1 class)."""
    from openess_amd import hip
    torch.manual_seed(12)
    for (B, H, W, Cin, Cout) in ((1, 9, 13, 64, 96), (8, 1, 1121, 768, 3072)):
        x = (torch.randn(B, H, W, Cin, device="cuda") * 1.5).bfloat16()
        w = torch.randn(Cout, Cin, 1, 1, device="cuda") / np.sqrt(Cin)
        b = torch.randn(Cout, device="cuda")
        packed = hip.pack_conv_weight(w)
        ref = F.gelu(ref_conv(x, w, b, 1, 0, 1))
        y32 = hip.conv2d_nhwc(x, packed, b, Cout, 1, 1, 1, 0, 1, relu=2, out_f32=True)
        pre = ref_conv(x, w, b, 1, 0, 1)
        mine = hip.conv2d_nhwc(x, packed, b, Cout, 1, 1, 1, 0, 1, out_f32=True)
        # isolate the activation: apply exact GELU to the kernel's own pre-activation
        np.testing.assert_allclose(y32.cpu().numpy(), F.gelu(mine).cpu().numpy(), rtol=0, atol=2e-6 * float(pre.abs().max()) + 2e-6)
        y = hip.conv2d_nhwc(x, packed, b, Cout, 1, 1, 1, 0, 1, relu=2)
        np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("geom", [(8, 440, 640), (3, 37, 51), (1, 9, 7), (2, 64, 96)])
def test_e2vid_head_enc0_fused_equals_two_convs(geom):
    """E2VID head (5x5, 8 -> 32, ReLU) + encoder 0 conv (5x5 stride 2, 32 -> 64, ReLU) in one kernel (the 32-channel head output
    stays in LDS) against the two conv launches -- bit-equal: the same bf16 rounding point between the layers and the same
    k order in both products -- and against the fp32 reference; output into a channel slice of a wider buffer."""
    from openess_amd import hip
    B, H, W = geom
    torch.manual_seed(sum(geom))
    x8 = torch.zeros(B, H, W, 8, device="cuda", dtype=torch.bfloat16)
    x8[..., :5] = torch.randn(B, H, W, 5, device="cuda").bfloat16()
    wh = torch.zeros(32, 8, 5, 5, device="cuda")
    wh[:, :5] = torch.randn(32, 5, 5, 5, device="cuda") / np.sqrt(125)
    bh = torch.randn(32, device="cuda") * 0.1
    we = torch.randn(64, 32, 5, 5, device="cuda") / np.sqrt(800)
    be = torch.randn(64, device="cuda") * 0.1
    ph, pe = hip.pack_conv_weight(wh), hip.pack_conv_weight(we)
    head = hip.conv2d_nhwc(x8, ph, bh, 32, 5, 5, 1, 2, 1, relu=True)
    two = hip.conv2d_nhwc(head, pe, be, 64, 5, 5, 2, 2, 1, relu=True)
    Ho, Wo = two.shape[1], two.shape[2]
    buf = torch.full((B, Ho, Wo, 128), 3.0, device="cuda", dtype=torch.bfloat16)
    one = hip.e2vid_head_enc0(x8, ph, bh, True, pe, be, True, out=buf[..., :64])
    assert float((buf[..., 64:] - 3).abs().max()) == 0
    ref_h = ref_conv(x8, wh, bh, 1, 2, 1).clamp_min(0).bfloat16()
    ref = ref_conv(ref_h, we, be, 2, 2, 1).clamp_min(0)
    np.testing.assert_allclose(one.float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-2, atol=2e-2)
    d = (one.float() - two.float()).abs()
    assert float(d.max()) <= 2e-2 and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))
    assert torch.equal(one, hip.e2vid_head_enc0(x8, ph, bh, True, pe, be, True))
    # no activation / no bias forms
    one2 = hip.e2vid_head_enc0(x8, ph, None, False, pe, None, False)
    ref2 = ref_conv(ref_conv(x8, wh, None, 1, 2, 1).bfloat16(), we, None, 2, 2, 1)
    np.testing.assert_allclose(one2.float().cpu().numpy(), ref2.cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("geom", [(8, 440, 640), (3, 37, 51), (2, 64, 96)])
@pytest.mark.parametrize("normalize", [True, False])
def test_e2vid_events_head_enc0_equals_relayout_plus_fused(geom, normalize):
    """oess_e2vid_events_head_enc0_bf16 (EventPreprocessor apply + NHWC8 packing inside the fused head / encoder-0 kernel) ==
    oess_event_slice_to_nhwc8_bf16 followed by oess_e2vid_head_enc0_bf16, bit for bit, on the third of four 5-bin slices."""
    from openess_amd import hip
    B, H, W = geom
    torch.manual_seed(sum(geom) + int(normalize))
    ev = torch.randn(B, 20, H, W, device="cuda")
    ev[ev.abs() < 0.7] = 0.0
    wh = torch.zeros(32, 8, 5, 5, device="cuda")
    wh[:, :5] = torch.randn(32, 5, 5, 5, device="cuda") / np.sqrt(125)
    bh = torch.randn(32, device="cuda") * 0.1
    we = torch.randn(64, 32, 5, 5, device="cuda") / np.sqrt(800)
    be = torch.randn(64, device="cuda") * 0.1
    ph, pe = hip.pack_conv_weight(wh), hip.pack_conv_weight(we)
    x8 = hip.event_slice_to_nhwc8(ev, 10, 5, normalize=normalize)                  # logical [B, 8, H, W] channels_last
    two = hip.e2vid_head_enc0(x8.permute(0, 2, 3, 1), ph, bh, True, pe, be, True)
    one = hip.e2vid_events_head_enc0(ev, 10, 5, normalize, ph, bh, True, pe, be, True)
    assert torch.equal(one, two)
    # a tensor that is exactly one slice (the reference-contract call form) takes the single-slice statistics
    sl = ev[:, 10:15].contiguous()
    assert torch.equal(hip.e2vid_events_head_enc0(sl, 0, 5, normalize, ph, bh, True, pe, be, True), two)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 44, 64), (1, 23, 37), (8, 220, 320)])
def test_conv5x5s2_group_equals_separate_launches(B, H, W):
    """oess_conv5x5s2_group_bf16: the level-1 and level-2 encoder convs of the skewed schedule in one launch are bit-identical to two
    oess_conv2d_fwd_bf16 calls -- with the real buffer layout (the level-2 conv reads the h half of the cat(x, h) buffer whose x half
    the level-1 conv writes in the same launch); truly overlapping outputs and wrong output shapes are rejected."""
    import torch
    from openess_amd import hip
    torch.manual_seed(B + H)
    dev = "cuda"
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    h0 = (torch.randn(B, H, W, 128, device=dev) * 0.5).bfloat16()                 # cat buffer of level 0: its h half feeds level 1
    w1 = torch.randn(128, 64, 5, 5, device=dev) * 0.03
    w2 = torch.randn(256, 128, 5, 5, device=dev) * 0.02
    b1, b2 = torch.randn(128, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
    p1, p2 = hip.pack_conv_weight(w1), hip.pack_conv_weight(w2)
    res = []
    for grouped in (False, True):
        xh1 = (torch.randn(B, H1, W1, 256, device=dev, generator=torch.Generator(dev).manual_seed(3)) * 0.5).bfloat16()
        xh2 = torch.zeros(B, H2, W2, 512, device=dev, dtype=torch.bfloat16)
        h1_before = xh1[..., 128:].clone()
        probs = [(xh1[..., 128:], p2, b2, 256, False, xh2[..., :256]), (h0[..., 64:], p1, b1, 128, True, xh1[..., :128])]
        if grouped:
            hip.conv5x5s2_group(probs)
        else:
            for x, pk, bias, Co, relu, out in probs:
                hip.conv2d_nhwc(x, pk, bias, Co, 5, 5, 2, 2, 1, relu=relu, out=out)
        assert torch.equal(xh1[..., 128:], h1_before)                              # the h half is only read
        res.append((xh1.clone(), xh2.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[1][1][..., :256].float().abs().max()) > 0
    xh1 = torch.zeros(B, H1, W1, 256, device=dev, dtype=torch.bfloat16)
    xh2 = torch.zeros(B, H2, W2, 512, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):            # level 2 reads what level 1 writes
        hip.conv5x5s2_group([(xh1[..., :128], p2, b2, 256, False, xh2[..., :256]), (h0[..., 64:], p1, b1, 128, True, xh1[..., :128])])
    with pytest.raises(ValueError):
        hip.conv5x5s2_group([(h0[..., 64:], p1, b1, 128, True, xh1[..., :64])])


def test_packed_weight_group_refresh_equals_individual_packing():
    """engine.PackedWeight: after an in-place update of several conv weights (an optimiser step) the first stale operand that is
    asked for repacks ALL stale registered operands -- forward and existing data-gradient operands -- with one launch into their
    existing buffers; the operands equal freshly packed ones bit for bit, frozen and derived weights are left alone, and the
    second step reuses the cached device table."""
    from openess_amd import engine, hip
    torch.manual_seed(0)
    convs = [torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda") * 0.1) for co, ci, k in ((64, 32, 3), (136, 64, 1), (72, 200, 3), (32, 64, 5))]
    frozen = torch.nn.Parameter(torch.randn(16, 8, 3, 3, device="cuda"), requires_grad=False)
    pws = [engine.PackedWeight() for _ in convs]
    pwf = engine.PackedWeight()
    x = [torch.randn(2, c.shape[1], 9, 11, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last) for c in convs]
    for rep in range(3):
        outs = []
        for w, pw, xi in zip(convs, pws, x):
            xi = xi.clone().requires_grad_(True)
            y = engine.conv2d_train(xi, w, None, pw, w.shape[2], 1, w.shape[2] // 2, 1)
            y.float().sum().backward()                       # creates / uses the data-gradient operand
            outs.append(y.detach())
        pwf.get(frozen, None, None, cin_pad=8)
        buffers = [(pw.packed.data_ptr(), pw.packed_flip.data_ptr()) for pw in pws]
        for w, pw, o, xi in zip(convs, pws, outs, x):
            assert torch.equal(pw.packed, hip.pack_conv_weight(w)) and torch.equal(pw.packed_flip, hip.pack_conv_weight(w, flip=True))
            ref = hip.conv2d_nhwc(xi.permute(0, 2, 3, 1), hip.pack_conv_weight(w), None, w.shape[0], w.shape[2], w.shape[2], 1, w.shape[2] // 2, 1)
            assert torch.equal(o.permute(0, 2, 3, 1), ref)
        with torch.no_grad():
            for w in convs:
                w.add_(torch.randn_like(w) * 0.01)             # the "optimiser step": every weight stale at once
                w.grad = None
        if rep > 0:
            assert [(pw.packed.data_ptr(), pw.packed_flip.data_ptr()) for pw in pws] == buffers      # repacked in place
    # one call repacks everything that is stale, and nothing when nothing is
    n = engine.PackedWeight.refresh_stale()
    assert n == 2 * len(convs) and engine.PackedWeight.refresh_stale() == 0
    assert pwf._wref is None                                                                       # frozen weights never register
