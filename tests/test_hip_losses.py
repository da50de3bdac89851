"""GPU parity: fused Dice+CE, superpixel scatter-mean (fwd+bwd) and confusion matrix vs golden
vectors produced by the reference (fp32 tolerance stated per check; integers exact)."""
import numpy as np
import pytest
import torch

from oracle import losses as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,K", [("a", 11), ("b", 6)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_task_loss_golden(golden_losses, tag, K, layout):
    from openess_amd import hip
    g = golden_losses
    lg = torch.from_numpy(g[f"task_{tag}_logits"]).cuda()
    if layout == "nhwc":
        lg = lg.contiguous(memory_format=torch.channels_last)
    lg.requires_grad_(True)
    tgt = torch.from_numpy(g[f"task_{tag}_target"]).cuda()
    total, parts = hip.task_loss(lg, tgt, K)
    (total * 1.0).backward()
    np.testing.assert_allclose(total.item(), g[f"task_{tag}_loss"], rtol=2e-6)
    np.testing.assert_allclose(parts[0].item(), g[f"dice_{tag}_loss"], rtol=2e-6)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g[f"task_{tag}_grad"], rtol=1e-4, atol=1e-8)
    lg2 = torch.from_numpy(g[f"task_{tag}_logits"]).cuda().requires_grad_(True)
    d, _ = hip.task_loss(lg2, tgt, K, losses=("dice",))
    (3.0 * d).backward()            # upstream scale goes through the device scalar
    np.testing.assert_allclose(lg2.grad.cpu().numpy(), 3.0 * g[f"dice_{tag}_grad"], rtol=1e-4, atol=1e-8)


def test_task_loss_full_size_vs_oracle():
    """DSEC size 2 x 11 x 440 x 640 (bf16 NHWC logits as produced by the decoder) vs fp32 oracle."""
    from openess_amd import hip
    torch.manual_seed(0)
    B, K, H, W = 2, 11, 440, 640
    lg = (torch.randn(B, K, H, W) * 3).bfloat16()
    tgt = torch.randint(0, K, (B, H, W))
    tgt[torch.rand(B, H, W) < 0.05] = 255
    ref_in = lg.float().requires_grad_(True)
    ref = ol.task_loss(ref_in, tgt, K)
    ref.backward()
    x = lg.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    total, _ = hip.task_loss(x, tgt.cuda(), K)
    total.backward()
    np.testing.assert_allclose(total.item(), ref.item(), rtol=1e-5)
    # gradient is stored as bf16: 2^-8 relative
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), ref_in.grad.numpy(), rtol=1e-2, atol=1e-9)


@pytest.mark.parametrize("B,K,H,W", [(2, 11, 440, 640), (1, 11, 37, 53), (3, 6, 20, 31), (1, 16, 16, 16)])
def test_task_loss_dense_nhwc_fp32_equals_plane_layout(B, K, H, W):
    """fp32 channels_last logits (what the decoder's classifier conv writes) take the LDS-staged kernels; same loss and gradient as
    the NCHW layout through the generic kernels (identical per-pixel arithmetic; only the grouping of the partial sums differs),
    with ignored pixels, pixel counts that are not a multiple of the 256-pixel tile and an upstream scale."""
    from openess_amd import hip
    torch.manual_seed(K + H)
    lg = (torch.randn(B, K, H, W, device="cuda") * 3)
    tgt = torch.randint(0, K, (B, H, W), device="cuda")
    tgt[torch.rand(B, H, W, device="cuda") < 0.1] = 255
    a = lg.clone().requires_grad_(True)
    b = lg.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    la, pa = hip.task_loss(a, tgt, K)
    lb, pb = hip.task_loss(b, tgt, K)
    (la * 0.7).backward()
    (lb * 0.7).backward()
    np.testing.assert_allclose(lb.item(), la.item(), rtol=2e-6)
    np.testing.assert_allclose(pb[0].item(), pa[0].item(), rtol=2e-6)
    assert b.grad.is_contiguous(memory_format=torch.channels_last) or b.grad.shape == a.grad.shape
    np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=2e-5, atol=1e-10)
    assert float(b.grad[(tgt == 255)[:, None].expand_as(b.grad)].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["a", "b"])
def test_superpixel_pool_golden(golden_losses, tag):
    from openess_amd import hip
    g = golden_losses
    fk = torch.from_numpy(g[f"sp_{tag}_feat_k"]).cuda().requires_grad_(True)
    fq = torch.from_numpy(g[f"sp_{tag}_feat_q"]).cuda().requires_grad_(True)
    ids = torch.from_numpy(g[f"sp_{tag}_ids"]).cuda()
    sps = int(g[f"sp_{tag}_size"])
    k = hip.superpixel_pool(fk, ids, sps)
    q = hip.superpixel_pool(fq, ids, sps)
    assert tuple(k.shape) == g[f"sp_{tag}_k"].shape          # S = max id + 1, data dependent
    np.testing.assert_allclose(k.detach().cpu().numpy(), g[f"sp_{tag}_k"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q.detach().cpu().numpy(), g[f"sp_{tag}_q"], rtol=1e-5, atol=1e-6)
    w = torch.from_numpy(g[f"sp_{tag}_w"]).cuda()
    ((k * w).sum() + (q * w.flip(0)).sum()).backward()
    np.testing.assert_allclose(fk.grad.cpu().numpy(), g[f"sp_{tag}_gk"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(fq.grad.cpu().numpy(), g[f"sp_{tag}_gq"], rtol=1e-5, atol=1e-7)


def test_superpixel_pool_full_size():
    """B=2 x 256 ch x 440 x 640, 10x10 block superpixels + ids > superpixel_size collisions."""
    from openess_amd import hip
    torch.manual_seed(1)
    B, C, H, W, sps = 2, 256, 440, 640, 100
    feat = torch.randn(B, C, H, W)
    yy = (torch.arange(H) * 10 // H)[:, None]
    xx = (torch.arange(W) * 10 // W)[None, :]
    ids = (yy * 10 + xx)[None].repeat(B, 1, 1).long()
    ids[1, :40, :64] = 130                                   # id >= superpixel_size: collides across samples
    ref = ol.superpixel_pool(feat, ids, sps)
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1e-2)):
        f = feat.to(dt).cuda().contiguous(memory_format=torch.channels_last)
        k = hip.superpixel_pool(f, ids.cuda(), sps)
        assert k.shape == ref.shape
        refd = ol.superpixel_pool(feat.to(dt).float(), ids, sps)
        np.testing.assert_allclose(k.cpu().numpy(), refd.numpy(), rtol=tol, atol=tol * 1e-2)


@pytest.mark.parametrize("C,dt", [(64, torch.bfloat16), (128, torch.float32), (256, torch.bfloat16), (40, torch.float32)])
def test_superpixel_pool_random_ids(C, dt):
    """Adversarial ids (no spatial coherence: every pixel ends a run), values up to 255 (> superpixel_size and beyond
    the LDS table of the vectorised kernel), ragged pixel counts; C = 40 takes the generic lane-per-channel kernel."""
    from openess_amd import hip
    torch.manual_seed(C)
    B, H, W, sps = 3, 37, 53, 100
    feat = torch.randn(B, C, H, W)
    ids = torch.randint(0, 256, (B, H, W))
    ids[0, :, :20] = 7                                        # one long run as well
    f = feat.to(dt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    k = hip.superpixel_pool(f, ids.cuda(), sps)
    fr = feat.to(dt).float().requires_grad_(True)
    ref = ol.superpixel_pool(fr, ids, sps)
    assert k.shape == ref.shape
    np.testing.assert_allclose(k.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    w = torch.randn_like(ref)
    (k * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    tol = 1e-2 if dt == torch.bfloat16 else 1e-5
    np.testing.assert_allclose(f.grad.float().cpu().numpy(), fr.grad.numpy(), rtol=tol, atol=tol * 1e-2)


@pytest.mark.parametrize("C,dt", [(256, torch.bfloat16), (256, torch.float32), (40, torch.float32), (96, torch.bfloat16)])
def test_superpixel_pool_bit_repeatable_and_exact(C, dt):
    """K7 forward meets its partial sums as 64-bit fixed-point integers: (a) five calls on the same input give identical bits
    (random ids <= 300: LDS-table path, direct-global path and cross-sample collisions all active), interleaved with other work
    that perturbs the workgroup schedule; (b) features that are multiples of 2^-8 have exactly representable sums, so the
    result equals the float64 quotient rounded to fp32 by the same two roundings -- bit for bit."""
    from openess_amd import hip
    torch.manual_seed(C)
    B, H, W, sps = 4, 120, 160, 100
    q = (torch.randint(-1024, 1024, (B, C, H, W)).float() / 256.0).to(dt).float()   # |x| <= 4, multiples of 2^-8 (also after bf16 rounding)
    ids = torch.randint(0, 300, (B, H // 4, W // 4)).repeat_interleave(4, 1).repeat_interleave(4, 2)
    ids[1] = torch.randint(0, 256, (H, W))                                    # one sample without any spatial coherence
    f = q.to(dt).cuda().contiguous(memory_format=torch.channels_last)
    idc = ids.cuda()
    outs = []
    for rep in range(5):
        outs.append(hip.superpixel_pool(f, idc, sps).clone())
        torch.randn(1 << (18 + rep), device="cuda").sum()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    gid = (ids + torch.arange(B)[:, None, None] * sps).reshape(-1)
    S = int(gid.max()) + 1
    assert outs[0].shape == (S, C)
    pm = q.permute(0, 2, 3, 1).reshape(-1, C).double()
    sums = torch.zeros(S, C, dtype=torch.float64).index_add_(0, gid, pm)
    cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, gid, torch.ones_like(gid, dtype=torch.float64))
    ref = sums.float() / (cnt.float() + 1e-6)[:, None]                        # exact sums -> one rounding; fp32 add of 1e-6; fp32 divide
    assert torch.equal(outs[0].cpu(), ref)


def test_superpixel_pool_baseline_size_conservation_and_repeat():
    """BASELINE size (B = 8 x 256 ch x 440 x 640, bf16, SAM-like irregular uint8 regions that collide across samples): size-
    independent properties of the scatter-mean -- (a) conservation: sum_s k[s] * count[s] equals the per-channel sum of all
    features (every pixel lands in exactly one row), (b) the counts add up to the pixel count, (c) two calls give identical bits,
    (d) the backward of sum(k * w) hands every pixel the row w[id] / (count + 1e-6)."""
    from openess_amd import hip
    torch.manual_seed(3)
    B, C, H, W, sps = 8, 256, 440, 640, 100
    feat = (torch.randint(-512, 512, (B, H, W, C), device="cuda").float() / 128.0).bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    coarse = torch.randint(0, 256, (B, H // 20, W // 20), device="cuda")
    ids = coarse.repeat_interleave(20, 1).repeat_interleave(20, 2)
    ids = torch.roll(ids, shifts=(7, 13), dims=(1, 2)).contiguous()                 # regions not aligned to the pixel chunks
    S = int((ids + torch.arange(B, device="cuda")[:, None, None] * sps).max()) + 1
    k = hip.superpixel_pool(feat, ids, sps, S=S)
    k2 = hip.superpixel_pool(feat.detach(), ids, sps, S=S)
    assert torch.equal(k.detach(), k2)
    gid = (ids + torch.arange(B, device="cuda")[:, None, None] * sps).reshape(-1)
    cnt = torch.bincount(gid, minlength=S).double()
    assert int(cnt.sum()) == B * H * W
    total = feat.detach().permute(0, 2, 3, 1).reshape(-1, C).double().sum(0)       # exact: multiples of 2^-7, |x| <= 4
    back = (k.detach().double() * (cnt.float() + 1e-6).double()[:, None]).sum(0)
    np.testing.assert_allclose(back.cpu().numpy(), total.cpu().numpy(), rtol=0, atol=2e-2)       # 800 rows x fp32 rounding of k
    w = torch.randn(S, C, device="cuda")
    (k * w).sum().backward()
    p = torch.randint(0, B * H * W, (4096,), device="cuda")
    g = feat.grad.permute(0, 2, 3, 1).reshape(-1, C)[p].float()
    want = (w / (cnt.float() + 1e-6)[:, None])[gid[p]]
    np.testing.assert_allclose(g.cpu().numpy(), want.bfloat16().float().cpu().numpy(), rtol=0, atol=0)


def test_superpixel_pool_out_of_range_input_is_loud():
    """Range contract of the fixed-point sums: a non-finite feature or |x| >= 32768 turns the whole output into NaN."""
    from openess_amd import hip
    f = torch.randn(1, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
    ids = torch.zeros(1, 16, 16, dtype=torch.int64, device="cuda")
    assert torch.isfinite(hip.superpixel_pool(f, ids, 100)).all()
    for bad in (float("inf"), float("nan"), 4.0e4):
        g = f.clone()
        g[0, 3, 5, 7] = bad
        assert torch.isnan(hip.superpixel_pool(g, ids, 100)).all()
    g = f.clone()
    g[0, 3, 5, 7] = 32000.0
    assert torch.isfinite(hip.superpixel_pool(g, ids, 100)).all()


def test_confusion_matrix(golden_losses):
    from openess_amd import hip
    g = golden_losses
    conf = torch.zeros(121, dtype=torch.int64, device="cuda")
    for p, t in zip(g["met_pred"], g["met_gt"]):
        hip.confusion_accumulate(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), 11, 255, conf)
    cm = conf.view(11, 11).cpu().numpy()
    assert np.array_equal(cm, g["met_cm"])
    miou, _, acc = ol.miou_acc(cm)
    assert miou == pytest.approx(float(g["met_miou"]), rel=1e-12)
    # full-size: 8 x 440 x 640
    pred = torch.randint(0, 11, (8, 440, 640), device="cuda")
    gt = torch.randint(0, 11, (8, 440, 640), device="cuda")
    gt[torch.rand(8, 440, 640, device="cuda") < 0.05] = 255
    conf.zero_()
    hip.confusion_accumulate(pred, gt, 11, 255, conf)
    assert np.array_equal(conf.view(11, 11).cpu().numpy(), ol.confusion_matrix(pred.cpu().numpy(), gt.cpu().numpy(), 11))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["feat", "logit"])
@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_consistency_losses_golden(tag, layout):
    """a16 (openess_trainer.py:497-503): HIP L1-mean and cosine-mean losses + input gradients vs the golden torch
    values (fp32: 1e-6 relative on the loss, 1e-6 absolute on the gradients = summation order only), and bf16 operands
    against the same formulas evaluated on the bf16-rounded inputs."""
    import os
    import torch
    import torch.nn.functional as F
    from openess_amd import hip
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consistency.npz"))

    def prep(x, dtype=torch.float32):
        t = torch.from_numpy(x).cuda().to(dtype)
        if layout == "channels_last":
            t = t.contiguous(memory_format=torch.channels_last)
        return t.requires_grad_(True)

    a, b = prep(g[f"{tag}_a"]), prep(g[f"{tag}_b"])
    l1 = hip.l1_mean(a, b)
    ga, gb = torch.autograd.grad(l1, (a, b))
    np.testing.assert_allclose(l1.item(), g[f"{tag}_l1"], rtol=1e-6)
    np.testing.assert_allclose(ga.cpu().numpy(), g[f"{tag}_l1_ga"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(gb.cpu().numpy(), g[f"{tag}_l1_gb"], rtol=0, atol=1e-8)
    lc = hip.cosine_mean_loss(a, b)
    ga, gb = torch.autograd.grad(lc, (a, b))
    np.testing.assert_allclose(lc.item(), g[f"{tag}_cos"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(ga.cpu().numpy(), g[f"{tag}_cos_ga"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(gb.cpu().numpy(), g[f"{tag}_cos_gb"], rtol=1e-4, atol=1e-7)
    # bf16 operands
    a16, b16 = prep(g[f"{tag}_a"], torch.bfloat16), prep(g[f"{tag}_b"], torch.bfloat16)
    ar, br = a16.detach().float().requires_grad_(True), b16.detach().float().requires_grad_(True)
    ref1 = F.l1_loss(ar, br)
    refc = torch.mean(1 - F.cosine_similarity(ar, br, dim=1))
    rga, rgb = torch.autograd.grad(refc, (ar, br))
    l1 = hip.l1_mean(a16, b16)
    lc = hip.cosine_mean_loss(a16, b16)
    ga, gb = torch.autograd.grad(lc, (a16, b16))
    np.testing.assert_allclose(l1.item(), ref1.item(), rtol=1e-5)
    np.testing.assert_allclose(lc.item(), refc.item(), rtol=1e-5, atol=1e-6)
    assert ga.dtype == torch.bfloat16
    np.testing.assert_allclose(ga.float().cpu().numpy(), rga.cpu().numpy(), rtol=1e-2, atol=1e-6)
    np.testing.assert_allclose(gb.float().cpu().numpy(), rgb.cpu().numpy(), rtol=1e-2, atol=1e-6)


@pytest.mark.gpu
def test_nce_loss_golden(golden_losses):
    """a14 NCELoss (utils/loss_functions.py:140-154) on the HIP kernels vs the reference's golden loss and both input
    gradients (fp32 throughout: 2e-6 relative on the loss, gradients to summation order), plus a full-size S = 800 case
    against the oracle."""
    from openess_amd.utils.loss_functions import NCELoss
    g = golden_losses
    k = torch.from_numpy(g["nce_k"]).cuda().requires_grad_(True)
    q = torch.from_numpy(g["nce_q"]).cuda().requires_grad_(True)
    loss = NCELoss(temperature=0.07)(k, q)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["nce_loss"], rtol=2e-6)
    # gradients are O(1) sums of ~S terms of mixed sign: fp32 summation order shows up at 1e-5 absolute
    np.testing.assert_allclose(k.grad.cpu().numpy(), g["nce_gk"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(q.grad.cpu().numpy(), g["nce_gq"], rtol=2e-5, atol=1e-5)
    torch.manual_seed(2)
    for S, C in ((800, 256), (37, 64), (9, 6)):              # BASELINE size; odd S (ragged last row pair); C % 4 != 0: generic kernel
        kk = torch.nn.functional.normalize(torch.randn(S, C), dim=1)
        qq = torch.nn.functional.normalize(torch.randn(S, C), dim=1)
        kk[5] = 0; qq[7] = 0                                     # empty superpixels: zero rows still enter the loss
        kr, qr = kk.clone().requires_grad_(True), qq.clone().requires_grad_(True)
        ref = ol.nce_loss(kr, qr, 0.07)
        ref.backward()
        kd, qd = kk.cuda().requires_grad_(True), qq.cuda().requires_grad_(True)
        out = NCELoss(temperature=0.07)(kd, qd)
        (out * 3.0).backward()
        np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-5)
        np.testing.assert_allclose(kd.grad.cpu().numpy(), 3.0 * kr.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(qd.grad.cpu().numpy(), 3.0 * qr.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_adamw_multi_tensor_matches_torch():
    """a18: openess_amd.utils.optim.AdamW (one multi-tensor HIP launch) vs torch.optim.AdamW over 4 steps on a ragged
    parameter list (a parameter whose grad is None at first joins later; two groups with different lr / weight decay),
    and state_dict interchange in both directions."""
    from openess_amd.utils.optim import AdamW
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (7,), (300, 300), (1,), (70000,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    mk = lambda cls, ps: cls([{'params': ps[:3], 'lr': 5e-4}, {'params': ps[3:], 'lr': 5e-6, 'weight_decay': 0.05}])
    ref, mine = mk(torch.optim.AdamW, ref_p), mk(AdamW, my_p)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref_p, my_p)):
            if i == 2 and step < 2:
                a.grad = b.grad = None                                   # joins at step 2
                continue
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        ref.step(); mine.step()
        for a, b in zip(ref_p, my_p):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)   # 1 ulp of the update
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    assert sd_ref['state'].keys() == sd_mine['state'].keys()
    for k in sd_ref['state']:
        assert float(sd_ref['state'][k]['step']) == float(sd_mine['state'][k]['step'])
        np.testing.assert_allclose(sd_mine['state'][k]['exp_avg_sq'].cpu().numpy(), sd_ref['state'][k]['exp_avg_sq'].cpu().numpy(), rtol=2e-6, atol=1e-12)
    mine.load_state_dict(sd_ref); ref.load_state_dict(sd_mine)         # interchangeable


@pytest.mark.gpu
def test_pointwise_feature_pool_equals_pooling_the_convolved_map():
    """hip.PointwiseFeature.pool: superpixel mean of a 1x1 convolution = the convolution of the superpixel mean (plus the bias
    times n / (n + 1e-6)), reference order training/pretrain_trainer.py:445-465 on models/style_networks.py:166's map.
    Against float64 of the reference order: forward 2e-3 of the largest value (x itself is bf16; the materialised form adds a
    second bf16 rounding and is checked at 1e-2), weight / bias / input gradients 1e-2; empty superpixels give exact zeros."""
    from openess_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    B, Cin, Cout, H, W, sps = 2, 32, 256, 40, 64, 100
    conv = torch.nn.Conv2d(Cin, Cout, 1).to(dev)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sp = torch.randint(0, 60, (B, H, W), generator=g).to(dev)            # ids 60..99 stay empty
    S = B * sps
    gk = torch.randn(S, Cout, generator=g).to(dev)
    k = hip.superpixel_pool(hip.PointwiseFeature(x, conv), sp, sps, S)
    k.backward(gk)
    got = (k.detach().clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), x.grad.float().clone())
    conv.zero_grad(); x.grad = None
    # reference order in float64
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xd, wd, bd).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    ids = (sp + torch.arange(B, device=dev)[:, None, None] * sps).reshape(-1)
    one = torch.zeros(S, B * H * W, dtype=torch.float64, device=dev)
    one[ids, torch.arange(B * H * W, device=dev)] = 1.0
    kd = (one @ y) / (one.sum(1, keepdim=True) + 1e-6)
    kd.backward(gk.double())

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max())
    assert rel(got[0], kd.detach()) < 2e-3
    empty = one.sum(1) == 0
    assert bool(empty.any()) and float(got[0][empty].abs().max()) == 0.0
    assert rel(got[1], wd.grad) < 1e-2 and rel(got[2], bd.grad) < 1e-2 and rel(got[3], xd.grad) < 1e-2
    # the materialised path on the same inputs
    k2 = hip.superpixel_pool(conv(x.detach().float()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last), sp, sps, S)
    assert rel(k2.detach(), kd.detach()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(2, 256, 11, 16), (1, 64, 7, 9), (2, 128, 5, 40)])
def test_upsampled_normalized_feature_pool(B, C, H, W):
    """hip.UpsampledNormalizedFeature.pool = scatter_mean(F.normalize(nn.Upsample(x4, bilinear, align_corners=True)(x))) with a
    one-pass backward (models/image_model.py:121-143 + training/pretrain_trainer.py:445-465).  The forward uses the same two
    kernels as the composed path (bit-equal); the input gradient is compared with float64 autograd of the reference ops at 1.5e-2
    of its largest value (bf16 input, bf16 saved map; the composed path, which rounds two more tensors to bf16, at 3e-2) and is
    bit-identical between two runs."""
    from openess_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + W)
    sps = 50
    x = torch.randn(B, C, H, W, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sp = torch.randint(0, 37, (B, 4 * H, 4 * W), generator=g).to(dev)
    sp[0, :2] = 49
    S = B * sps
    gk = torch.randn(S, C, generator=g).to(dev)
    k = hip.superpixel_pool(hip.UpsampledNormalizedFeature(x, 4), sp, sps, S)
    k.backward(gk)
    gx = x.grad.float().clone(); x.grad = None
    k2 = hip.superpixel_pool(hip.UpsampledNormalizedFeature(x, 4), sp, sps, S)
    k2.backward(gk)
    assert torch.equal(k, k2) and torch.equal(x.grad.float(), gx)
    x.grad = None
    kc = hip.superpixel_pool(hip.bilinear_l2norm_train(x, 4), sp, sps, S)
    kc.backward(gk)
    gx_c = x.grad.float().clone()
    assert torch.equal(kc, k)
    xd = x.detach().double().requires_grad_(True)
    up = torch.nn.functional.interpolate(xd, scale_factor=4, mode="bilinear", align_corners=True)
    fn = torch.nn.functional.normalize(up, p=2, dim=1).permute(0, 2, 3, 1).reshape(-1, C)
    ids = (sp + torch.arange(B, device=dev)[:, None, None] * sps).reshape(-1)
    sums = torch.zeros(S, C, dtype=torch.float64, device=dev).index_add_(0, ids, fn)
    cnt = torch.zeros(S, dtype=torch.float64, device=dev).index_add_(0, ids, torch.ones_like(ids, dtype=torch.float64))
    kd = sums / (cnt[:, None] + 1e-6)
    kd.backward(gk.double())
    ref = xd.grad

    def rel(a):
        return float((a.double() - ref).abs().max() / ref.abs().max())
    assert float((k.detach().double() - kd.detach()).abs().max()) < 2e-3
    assert rel(gx) < 1.5e-2, rel(gx)
    assert rel(gx_c) < 3e-2, rel(gx_c)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,h,w,Ho,Wo,dtype", [(2, 256, 7, 10, 110, 160, torch.bfloat16), (1, 64, 5, 6, 37, 53, torch.float32),
                                                  (3, 300, 4, 4, 64, 64, torch.bfloat16)])
def test_upsampled_feature_pool_through_the_pooling_matrix(B, C, h, w, Ho, Wo, dtype):
    """hip.UpsampledFeature.pool = scatter_mean(F.interpolate(y, size, bilinear, align_corners=False)) (models/deeplabv3.py:184 +
    training/pretrain_trainer.py:445-465) as a product with the pooling matrix.  Against float64 autograd of the reference ops:
    forward 1e-5 (fp32 input) / 1e-5 (bf16 input: the input is exact in bf16, the arithmetic is fp32) of the largest value,
    gradient 1e-5 (fp32) / 1e-2 (bf16 output rounding); counts exact; ids beyond superpixel_size land in the next sample's rows,
    rows beyond S are dropped; bit-identical between two runs (integer weight sums, fixed-order products)."""
    from openess_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + Wo)
    sps = 40
    y = torch.randn(B, C, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sp = torch.randint(0, 30, (B, Ho, Wo), generator=g).to(dev)
    sp[:, : Ho // 4, : Wo // 3] = 3                       # one large superpixel
    sp[0, -2:, :] = 45                                    # id >= superpixel_size: row 45 (sample 1's range, or dropped when B == 1)
    S = B * sps
    gk = torch.randn(S, C, generator=g).to(dev)
    k = hip.superpixel_pool(hip.UpsampledFeature(y, (Ho, Wo)), sp, sps, S)
    k.backward(gk)
    gy = y.grad.float().clone(); y.grad = None
    k2 = hip.superpixel_pool(hip.UpsampledFeature(y, (Ho, Wo)), sp, sps, S)
    k2.backward(gk)
    assert torch.equal(k, k2) and torch.equal(y.grad.float(), gy)
    yd = y.detach().double().requires_grad_(True)
    up = torch.nn.functional.interpolate(yd, size=(Ho, Wo), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(-1, C)
    ids = (sp + torch.arange(B, device=dev)[:, None, None] * sps).reshape(-1)
    ok = ids < S
    sums = torch.zeros(S, C, dtype=torch.float64, device=dev).index_add_(0, ids[ok], up[ok])
    cnt = torch.zeros(S, dtype=torch.float64, device=dev).index_add_(0, ids[ok], torch.ones(int(ok.sum()), dtype=torch.float64, device=dev))
    kd = sums / (cnt[:, None] + 1e-6)
    kd.backward(gk.double())
    assert float((k.detach().double() - kd.detach()).abs().max() / kd.detach().abs().max()) < 1e-5
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((gy.double() - yd.grad).abs().max() / yd.grad.abs().max()) < tol
    # the materialised path (full-resolution tensor -> K7) agrees to its own rounding
    km = hip.superpixel_pool(hip.UpsampledFeature(y, (Ho, Wo)).materialize(), sp, sps, S)
    assert float((km.detach().double() - kd.detach()).abs().max() / kd.detach().abs().max()) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.gpu
def test_l1_mean_of_two_upsampled_features_and_dense_cosine():
    """(a) hip.l1_mean on two hip.UpsampledFeature = nn.L1Loss on the two interpolated maps (training/openess_trainer.py:497 on
    models/deeplabv3.py:184's outputs), computed as mean |up(a - b)|: loss 2e-3 relative to float64 of the reference ops (one
    bf16 rounding of the difference and one of the upsampled map), gradients: cosine with the float64 gradient >= 0.995 (sign
    flips where |up(a - b)| is within a bf16 ulp of zero).  (b) vectorised / tail / single-operand L1 against torch.
    (c) cosine consistency on fp32 channels_last logits with K = 11, 19, 6 (dense small-C kernels): loss 1e-6, gradients 1e-5 of
    float64 autograd of mean(1 - cosine_similarity)."""
    from openess_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, C, h, w, Ho, Wo = 2, 64, 6, 9, 90, 140
    a = torch.randn(B, C, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(B, C, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    l = hip.l1_mean(hip.UpsampledFeature(a, (Ho, Wo)), hip.UpsampledFeature(b, (Ho, Wo)))
    l.backward()
    ad, bd = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    up = lambda t: torch.nn.functional.interpolate(t, size=(Ho, Wo), mode="bilinear", align_corners=False)
    ld = (up(ad) - up(bd)).abs().mean()
    ld.backward()
    assert abs(float(l.detach()) - float(ld.detach())) / float(ld.detach()) < 2e-3
    cos = torch.nn.functional.cosine_similarity
    assert float(cos(a.grad.double().flatten(), ad.grad.flatten(), dim=0)) > 0.995
    assert float(cos(b.grad.double().flatten(), bd.grad.flatten(), dim=0)) > 0.995
    # (b)
    for n, dt in ((8 * 1000 + 5, torch.bfloat16), (4 * 777 + 3, torch.float32), (64, torch.bfloat16)):
        x = torch.randn(n, generator=g).to(dev).to(dt).requires_grad_(True)
        y = torch.randn(n, generator=g).to(dev).to(dt).requires_grad_(True)
        l = hip.l1_mean(x.view(1, 1, 1, n), y.view(1, 1, 1, n))
        l.backward()
        ref = (x.detach().double() - y.detach().double()).abs().mean()
        assert abs(float(l.detach()) - float(ref)) < 1e-6 * max(1.0, float(ref))
        sg = torch.sign(x.detach().float() - y.detach().float()) / n
        assert torch.allclose(x.grad.float(), sg.to(dt).float(), atol=0, rtol=0) and torch.allclose(y.grad.float(), (-sg).to(dt).float(), atol=0, rtol=0)
    # (c)
    for K in (11, 19, 6):
        P = (2, 37, 53)
        la = torch.randn(P[0], K, P[1], P[2], generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        lb = torch.randn(P[0], K, P[1], P[2], generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        l = hip.cosine_mean_loss(la, lb)
        l.backward()
        xd, yd = la.detach().double().requires_grad_(True), lb.detach().double().requires_grad_(True)
        ld = (1 - torch.nn.functional.cosine_similarity(xd, yd, dim=1)).mean()
        ld.backward()
        assert abs(float(l.detach()) - float(ld.detach())) < 1e-6
        for got, ref in ((la.grad, xd.grad), (lb.grad, yd.grad)):
            assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5
