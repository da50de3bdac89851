"""GPU parity of the host-side model mirrors (HIP MFMA convs, bf16 activations, fp32 accumulate) against
the fp32 CPU oracle (itself pinned to the reference by tests/test_oracle_nets_golden.py) and directly
against the reference's golden outputs.  Tolerances are bf16-class: error is measured relative to the
tensor's max magnitude (activations are rounded to bf16 = 2^-8 after every layer)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nets as on
from oracle.step import E2VID_LIGHTWEIGHT_CONFIG, OracleStep
from tests.synth import compact, damp_residual, fill_by_name

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "nets.npz")))


@pytest.fixture(scope="module")
def keys():
    return json.load(open(os.path.join(GOLDEN, "nets_keys.json")))


def test_e2vid_recurrent_latents(g, keys):
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    m = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    assert sorted(m.state_dict().keys()) == keys["e2vid"]          # checkpoint-compatible with the reference
    fill_by_name(m, 11)
    m.cuda()
    ev = torch.from_numpy(g["e2vid_events"]).cuda()
    rec = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
    for i in range(3):
        _, states, latent = rec.update_reconstruction(ev[:, 5 * i:5 * i + 5])
    ref = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(ref, 11, keys["e2vid"])
    st = None
    with torch.no_grad():
        for i in range(3):
            _, st, lat_ref = ref(on.event_preprocess(ev[:, 5 * i:5 * i + 5].cpu()), st)
    for k in (1, 2, 4, 8):
        got = latent[k].float().cpu().numpy()
        assert got.shape == lat_ref[k].shape
        assert relerr(got, lat_ref[k].numpy()) < 3e-2, k
        # and directly against the reference's own output (compact golden form)
        sub, _, _ = compact(got)
        key = f"e2vid_latent{k}"
        refsub = g[key + "__sub"] if key + "__sub" in g else compact(g[key])[0]
        assert relerr(sub, refsub) < 3e-2, k
    # fused slice path == reference-contract path
    rec2 = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
    for i in range(3):
        _, _, latent2 = rec2.update_reconstruction(ev, channel_slice=(5 * i, 5))
    assert torch.equal(latent2[8], latent[8])
    # sub-windows whose latents the caller drops run head + encoder-0 conv as ONE kernel (need_latents=False): the states advance
    # identically (same products, same bf16 rounding point between the two layers), so the last sub-window's latents -- computed
    # the ordinary way -- match the all-unfused loop and the reference's golden
    rec3 = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
    for i in range(3):
        _, _, latent3 = rec3.update_reconstruction(ev, channel_slice=(5 * i, 5), need_latents=(i == 2))
    assert torch.equal(latent3[1], latent[1])
    for k in (2, 4, 8):
        assert relerr(latent3[k].float().cpu().numpy(), latent[k].float().cpu().numpy()) < 5e-3, k
        assert relerr(latent3[k].float().cpu().numpy(), lat_ref[k].numpy()) < 3e-2, k
    rec4 = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
    rec4.skew = False
    _, _, lat_none = rec4.update_reconstruction(ev, channel_slice=(0, 5), need_latents=False)
    assert lat_none[1] is None and lat_none[2].shape == latent[2].shape
    # skewed schedule (default for need_latents=False calls: level l works on sub-window s - l, the ConvLSTM steps of a call are one
    # launch) == plain order, bit for bit, over a sequence longer than the skew; the states after the draining call are complete
    ev5 = torch.cat([ev, ev.flip(1)[:, :10] * 0.5], 1).contiguous()           # 5 sub-windows
    lat = {}
    for skew in (False, True):
        r = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
        r.skew = skew
        for i in range(5):
            _, states, l5 = r.update_reconstruction(ev5, channel_slice=(5 * i, 5), need_latents=(i == 4))
        lat[skew] = ({k: v.clone() for k, v in l5.items()}, [s_['cell'].clone() for s_ in states], [s_['cur'] for s_ in states])
        if skew:
            assert not states.pending()
            # a sequence that stops without the draining call is drained by the next plain call (here: reconstruct=True)
            for i in range(3):
                r.update_reconstruction(ev5, channel_slice=(5 * i, 5), need_latents=False)
            assert r.last_states_for_each_channel['grayscale'].pending()
            img, _, _ = r.update_reconstruction(ev5, channel_slice=(15, 5), reconstruct=True)
            assert img.shape[-2:] == (32, 48) and bool(torch.isfinite(img).all())
    for k in (1, 2, 4, 8):
        assert torch.equal(lat[True][0][k], lat[False][0][k]), k
    for a_, b_ in zip(lat[True][1], lat[False][1]):
        assert torch.equal(a_, b_)
    assert lat[True][2] == lat[False][2]


@pytest.mark.gpu
def test_e2vid_recurrent_baseline_size_skew_equals_plain():
    """BASELINE size (B = 8, 440 x 640, 5 bins per sub-window): the skewed schedule -- grouped 256 x 128-tile ConvLSTM launches,
    grouped stride-2 encoder convs, head + encoder 0 as one kernel -- ends in the same bits as the plain order, which launches
    every level's 128 x 128-tile kernels one by one.  A size-independent property: the CPU oracle cannot run this size."""
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    m = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(m, 11)
    m.cuda()
    gen = torch.Generator(device="cuda").manual_seed(5)
    nwin = 5
    ev = torch.randn((8, 5 * nwin, 440, 640), device="cuda", generator=gen) * (torch.rand((8, 5 * nwin, 440, 640), device="cuda", generator=gen) < 0.3)
    out = {}
    for skew in (False, True):
        r = ImageReconstructor(m, 440, 640, 5, torch.device("cuda"))
        r.skew = skew
        for i in range(nwin):
            _, states, lat = r.update_reconstruction(ev, channel_slice=(5 * i, 5), need_latents=(i == nwin - 1))
        out[skew] = ({k: v.clone() for k, v in lat.items() if v is not None}, [s_['cell'].clone() for s_ in states])
        del r
    assert set(out[True][0]) == set(out[False][0])
    for k in out[True][0]:
        assert torch.equal(out[True][0][k], out[False][0][k]), k
        assert bool(torch.isfinite(out[True][0][k].float()).all())
    for a_, b_ in zip(out[True][1], out[False][1]):
        assert torch.equal(a_, b_)


@pytest.fixture(scope="module")
def gpre():
    return dict(np.load(os.path.join(GOLDEN, "e2vid_pre.npz")))


@pytest.mark.parametrize("case", ["sparse", "dense", "zeros", "single"])
def test_event_preprocessor_matches_reference_golden(gpre, case):
    """EventPreprocessor.__call__ on the HIP kernels vs the REFERENCE's own output
    (e2vid/utils/inference_utils.py:70-87, tests/golden/gen_golden_e2vid_pre.py): fp32 formula in the reference's operation
    order; statistics are double sums here vs torch's float32 tree sums there -> 2e-6 relative; the all-zero tensor passes
    through untouched and the single-non-zero tensor is NaN everywhere exactly like the reference (unguarded 0/0)."""
    from openess_amd.e2vid.utils.inference_utils import EventPreprocessor
    x = torch.from_numpy(gpre[f"pre_in_{case}"]).cuda()
    got = EventPreprocessor()(x).cpu().numpy()
    want = gpre[f"pre_out_{case}"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=2e-5, atol=2e-6)
    # the fused slice form feeds the encoder the same values (bf16, padded to 8 channels)
    from openess_amd import hip
    y = hip.event_slice_to_nhwc8(x.contiguous(), 0, 5).float().cpu().numpy()
    if case != "single":
        np.testing.assert_allclose(y[:, :5], want, rtol=1e-2, atol=1e-2)
        assert np.abs(y[:, 5:]).max() == 0


def test_update_reconstruction_with_padding_matches_reference_golden(gpre, keys):
    """three recurrent `ImageReconstructor.update_reconstruction` steps at 30x44 (CropParameters pads to
    32x48 by reflection) vs the outputs of the reference's own ImageReconstructor (e2vid/image_reconstructor.py:80-123)."""
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    m = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(m, 11)
    m.cuda()
    ev = torch.from_numpy(gpre["rec_events"]).cuda()
    rec = ImageReconstructor(m, 30, 44, 5, torch.device("cuda"))
    assert rec.crop.needs_pad
    for i in range(3):
        img, states, latent = rec.update_reconstruction(ev[:, 5 * i:5 * i + 5], reconstruct=True)
    for k in (1, 2, 4, 8):
        got = latent[k].float().cpu().numpy()
        assert got.shape == gpre[f"rec_latent_{k}"].shape
        assert relerr(got, gpre[f"rec_latent_{k}"]) < 3e-2, k
    from openess_amd.e2vid.model.submodules import ConvLSTM
    c2 = ConvLSTM.cell_nhwc(states[2]).permute(0, 3, 1, 2).float().cpu().numpy()   # deepest ConvLSTM cell state (reference order)
    assert c2.shape == gpre["rec_state_c_2"].shape and relerr(c2, gpre["rec_state_c_2"]) < 3e-2
    want = gpre["rec_img"]                                                     # the reference returns the padded 32x48 image
    got = img.float().cpu().numpy()
    if got.shape != want.shape:
        cp = rec.crop
        want = want[:, :, cp.iy0:cp.iy1, cp.ix0:cp.ix1]
    assert got.shape == want.shape and np.abs(got - want).max() < 3e-2


def test_e2vid_offline_reconstruction_image(g, keys):
    """the full UNetRecurrent forward (residual blocks, transposed-conv decoders on the dgrad kernels, pred + sigmoid)
    after 3 recurrent steps vs the reference's own image (golden) and the fp32 oracle."""
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    m = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(m, 11)
    m.cuda()
    ev = torch.from_numpy(g["e2vid_events"]).cuda()
    rec = ImageReconstructor(m, 32, 48, 5, torch.device("cuda"))
    for i in range(3):
        img, _, latent = rec.update_reconstruction(ev[:, 5 * i:5 * i + 5], reconstruct=True)
    assert img.shape == (2, 1, 32, 48) and img.dtype == torch.float32
    got = img.cpu().numpy()
    assert np.abs(got - g["e2vid_img"]).max() < 2e-2            # sigmoid output in [0, 1]; bf16 activations through 14 more layers
    ref = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG, full=True).eval()
    fill_by_name(ref, 11, keys["e2vid"])
    st = None
    with torch.no_grad():
        for i in range(3):
            img_ref, st, _ = ref(on.event_preprocess(ev[:, 5 * i:5 * i + 5].cpu()), st)
    assert np.abs(got - img_ref.numpy()).max() < 2e-2
    assert cos(got - got.mean(), img_ref.numpy() - img_ref.numpy().mean()) > 0.995
    # the transposed convolution alone against nn.ConvTranspose2d on the same bf16 operands
    torch.manual_seed(3)
    dec = m.unetrecurrent.decoders[0]
    x = torch.randn(2, 256, 7, 9, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = dec(x).float()
    t, bn = dec.transposed_conv2d, dec.norm_layer
    with torch.no_grad():
        yr = torch.nn.functional.conv_transpose2d(x.float(), t.weight.bfloat16().float(), None, 2, 2, 1)
        yr = torch.relu(torch.nn.functional.batch_norm(yr, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    assert y.shape == yr.shape == (2, 128, 14, 18)
    assert relerr(y.cpu().numpy(), yr.cpu().numpy()) < 2e-2


def test_semseg_e2vid_forward_backward(g, keys):
    from openess_amd import hip
    from openess_amd.models.style_networks import SemSegE2VID
    net = SemSegE2VID(256, 11, skip_connect=True, skip_type='concat', text_embeddings_path=None)
    assert sorted(net.state_dict().keys()) == keys["semseg"]
    fill_by_name(net, 12)
    net.cuda().train()
    lat = {k: torch.from_numpy(g[f"semseg_lat{k}"]).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
           for k in (1, 2, 4, 8)}
    tgt = torch.from_numpy(g["semseg_target"]).cuda()
    pred, x256 = net(lat)
    loss, _ = hip.task_loss(pred[1], tgt, 11)
    loss.backward()
    ref = on.SemSegE2VID(256, 11)
    fill_by_name(ref, 12, keys["semseg"])
    ref.train()
    lat_ref = {k: v.float().cpu() for k, v in lat.items()}
    pred_ref, x256_ref = ref(lat_ref)
    from oracle import losses as ol
    loss_ref = ol.task_loss(pred_ref[1], tgt.cpu(), 11)
    loss_ref.backward()
    assert relerr(pred[1].float().cpu().detach().numpy(), pred_ref[1].detach().numpy()) < 4e-2
    assert relerr(x256.float().cpu().detach().numpy(), x256_ref.detach().numpy()) < 4e-2
    assert loss.item() == pytest.approx(loss_ref.item(), rel=2e-2)
    # SURVEY 8d argmax agreement: >= 99.9 % wherever the oracle's top-2 margin exceeds the measured logit error bound
    lg, lr = pred[1].float().cpu().detach().numpy(), pred_ref[1].detach().numpy()
    srt = np.sort(lr, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 2 * np.abs(lg - lr).max()
    assert clear.mean() > 0.5 and (lg.argmax(1) == lr.argmax(1))[clear].mean() >= 0.999
    assert (lg.argmax(1) == lr.argmax(1)).mean() >= 0.97
    pr = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        if name.startswith("decoder_scale_5") or name == "text_embeddings":
            continue
        assert p.grad is not None, name
        c = cos(p.grad.cpu().numpy(), pr[name].grad.numpy())
        if name.endswith("model.0.bias") or name.endswith("model.3.bias"):
            continue        # conv bias in front of an affine-free InstanceNorm: true gradient is 0 (pure rounding noise)
        assert c > 0.98, (name, c)
    assert net.decoder_scale_5[0].weight.grad is None          # never used in the skip path (reference too)


def test_teacher_forward(g, keys):
    """Frozen dilated ResNet-50 teacher (BatchNorm in TRAIN mode, as the reference leaves it).  A random-weight
    50-layer net with batch-statistics BN amplifies rounding noise roughly 2x per stage, so correctness is
    checked block by block with the ORACLE's activation as each block's input (teacher forcing), plus an
    end-to-end direction check of the unit-norm output features on a larger image."""
    from openess_amd import engine
    from openess_amd.models.image_model import DilationFeatureExtractor
    t = DilationFeatureExtractor(None)
    assert sorted(k for k in t.encoder.state_dict().keys()) == keys["teacher_encoder"]
    fill_by_name(t.encoder, 13)
    fill_by_name(t.decoder[0], 14)
    t.cuda().train()
    ref = on.DilationFeatureExtractor()
    fill_by_name(ref.encoder, 13, keys["teacher_encoder"])
    fill_by_name(ref.decoder[0], 14)
    ref.train()
    torch.manual_seed(5)
    img = torch.rand(2, 3, 96, 128)
    with torch.no_grad():
        e, r = t.encoder, ref.encoder
        rr = r.maxpool(torch.relu(r.bn1(r.conv1(img))))
        x = e.stem(engine.to_cl_bf16(img.cuda()))
        assert relerr(x.float().cpu().numpy(), rr.numpy()) < 2e-2
        for ln in ("layer1", "layer2", "layer3", "layer4"):
            for blk, rblk in zip(getattr(e, ln), getattr(r, ln)):
                x_in = rr.cuda().bfloat16().contiguous(memory_format=torch.channels_last)
                rr_next = rblk(rr)
                y = blk(x_in)
                assert relerr(y.float().cpu().numpy(), rr_next.numpy()) < 3e-2, ln
                rr = rr_next
        feat = t(img.cuda())
        fr = ref(img)
    c = (feat.float().cpu().numpy() * fr.numpy()).sum(1)
    assert float(c.mean()) > 0.85       # chaotic amplification of bf16 rounding in a RANDOM-weight BN-train net
    # reference golden: running-stat update of the stem BN (momentum 0.1) on the golden image
    t2 = DilationFeatureExtractor(None)
    fill_by_name(t2.encoder, 13)
    t2.cuda().train()
    with torch.no_grad():
        t2(torch.from_numpy(g["teacher_img"]).cuda())
    np.testing.assert_allclose(t2.encoder.bn1.running_mean.cpu().numpy(), g["teacher_bn1_running_mean_after"], rtol=2e-2, atol=2e-3)


def test_deeplab_eval_and_train(g, keys):
    from openess_amd import hip
    from openess_amd.models.deeplabv3 import deeplabv3_resnet50
    net = deeplabv3_resnet50(num_classes=11, text_embeddings_path=None, output_stride=32, pretrained_backbone='')
    assert sorted(net.state_dict().keys()) == keys["deeplab"]
    fill_by_name(net, 15)
    net.cuda().eval()
    img = torch.from_numpy(g["deeplab_img"]).cuda()
    with torch.no_grad():
        lg, ft = net(img)
    ref = on.DeepLabV3(11, 32)
    fill_by_name(ref, 15, keys["deeplab"])
    ref.eval()
    with torch.no_grad():
        lr, fr = ref(img.cpu())
    assert relerr(lg.float().cpu().numpy(), lr.numpy()) < 5e-2
    assert relerr(ft.float().cpu().numpy(), fr.numpy()) < 5e-2
    net.train()
    net.classifier.ASPP.project[3].p = 0.0
    lg, _ = net(img)
    tgt = torch.from_numpy(g["deeplab_target"]).cuda()
    loss, _ = hip.task_loss(lg, tgt, 11)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["deeplab_train_loss"]), rel=5e-2)
    # Gradients: the golden case has a 2x3 feature map (12 samples per BatchNorm channel); the train-mode BN
    # backward then cancels all but a sliver of the incoming gradient and bf16 activation rounding dominates
    # the conv-weight gradients (a pure-PyTorch bf16 run of the same net shows the same, tools/debug_deeplab2.py).
    # What is well conditioned: every parameter receives a finite gradient, and the BN affine gradient matches.
    ref.train()
    ref.classifier.ASPP.project[3].p = 0.0
    from oracle import losses as ol
    lr2, _ = ref(img.cpu())
    ol.task_loss(lr2, tgt.cpu(), 11).backward()
    pr = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        if "pixel_feature" in name:
            assert p.grad is None
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    assert cos(net.classifier.classifier[1].weight.grad.cpu().numpy(), pr["classifier.classifier.1.weight"].grad.numpy()) > 0.9


def test_teacher_well_conditioned_end_to_end(g, keys):
    """The teacher on well-conditioned weights (residual branches damped x0.25, tests/synth.py:damp_residual): here bf16
    storage does not get amplified (the fp32 oracle with bf16 rounding points keeps cosine > 0.999 against itself,
    tests/test_oracle_nets_golden.py), so END-TO-END parity is held tight -- a kernel bug cannot hide behind "chaos"."""
    from openess_amd.models.image_model import DilationFeatureExtractor
    from tests.synth import damp_residual
    t = DilationFeatureExtractor(None)
    fill_by_name(t.encoder, 13)
    fill_by_name(t.decoder[0], 14)
    damp_residual(t.encoder)
    t.cuda().train()
    ref = on.DilationFeatureExtractor()
    fill_by_name(ref.encoder, 13, keys["teacher_encoder"])
    fill_by_name(ref.decoder[0], 14)
    damp_residual(ref.encoder)
    ref.train()
    img = torch.from_numpy(g["teacherwc_img"])
    with torch.no_grad():
        feat = t(img.cuda()).float().cpu().numpy()
        fr = ref(img).numpy()
    c = (feat * fr).sum(1)
    assert float(c.mean()) >= 0.999, float(c.mean())
    assert float(c.min()) >= 0.99, float(c.min())
    # and against the REFERENCE module's own output (compact golden form)
    sub, _, _ = compact(feat)
    assert cos(sub, g["teacherwc_feat__sub"]) >= 0.999


def test_deeplab_well_conditioned_forward_backward(g, keys):
    """DeepLabv3-R50 train step at 4x3x224x320 (OS16 map 14x20 = 1120 samples per BatchNorm channel; ASPP rates 6/12/18 all
    have in-range off-centre taps; models/deeplabv3.py:137-142,295-348) on well-conditioned weights: logits, loss, argmax
    and EVERY parameter gradient against the fp32 oracle, plus 16 gradients against the reference's own golden."""
    from openess_amd import hip
    from openess_amd.models.deeplabv3 import deeplabv3_resnet50
    from oracle import losses as ol
    from tests.synth import damp_residual, wc_image
    net = deeplabv3_resnet50(num_classes=11, text_embeddings_path=None, output_stride=32, pretrained_backbone='')
    fill_by_name(net, 15)
    damp_residual(net)
    net.cuda().train()
    net.classifier.ASPP.project[3].p = 0.0
    img = torch.from_numpy(wc_image())
    tgt = torch.from_numpy(g["deeplabwc_target"]).long()
    lg, ft = net(img.cuda())
    loss, _ = hip.task_loss(lg, tgt.cuda(), 11)
    loss.backward()
    ref = on.DeepLabV3(11, 32)
    fill_by_name(ref, 15, keys["deeplab"])
    damp_residual(ref)
    ref.train()
    ref.classifier.ASPP.project[3].p = 0.0
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    lr, fr = ref(img)
    loss_ref = ol.task_loss(lr, tgt, 11)
    loss_ref.backward()
    lgc, lrc = lg.float().detach().cpu().numpy(), lr.detach().numpy()
    # Tolerance CALIBRATED by the oracle itself: the fp32 oracle with nothing changed but bf16 rounding of every stored conv /
    # BatchNorm output (the storage points of this pipeline) moves THIS far from the plain fp32 oracle on this net -- post-ReLU
    # activations carry a common mode that the next train-mode BatchNorm removes, so rounding relative to the raw magnitude is
    # several times larger relative to the signal.  The HIP path must stay within 1.5x of that rounding-only distance.
    emu = on.DeepLabV3(11, 32)
    fill_by_name(emu, 15, keys["deeplab"])
    damp_residual(emu)
    emu.train()
    emu.classifier.ASPP.project[3].p = 0.0
    on.emulate_bf16_storage(emu)                      # bf16 activations, activation gradients and conv-weight operands; fp32 maths
    le, fe = emu(img.bfloat16().float())
    ol.task_loss(le, tgt, 11).backward()
    le, fe = le.detach(), fe.detach()
    rms = lambda a, b: float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).mean() / (np.asarray(b, np.float64) ** 2).mean()))  # noqa: E731
    e_emu, e_gpu = rms(le.numpy(), lrc), rms(lgc, lrc)
    f_emu, f_gpu = rms(fe.numpy(), fr.detach().numpy()), rms(ft.float().detach().cpu().numpy(), fr.detach().numpy())
    print(f"deeplab wc: rms rel error logits gpu {e_gpu:.4f} vs rounding-only {e_emu:.4f}; feats gpu {f_gpu:.4f} vs {f_emu:.4f}; "
          f"max-rel gpu {relerr(lgc, lrc):.4f} vs {relerr(le.numpy(), lrc):.4f}")
    assert e_gpu <= 1.5 * e_emu + 1e-3 and f_gpu <= 1.5 * f_emu + 1e-3
    assert relerr(lgc, lrc) <= 1.5 * relerr(le.numpy(), lrc) + 1e-3
    assert loss.item() == pytest.approx(loss_ref.item(), rel=1e-2)
    assert loss.item() == pytest.approx(float(g["deeplabwc_loss"]), rel=1e-2)            # the reference's own loss
    # argmax agreement.  Pixels whose oracle top-2 margin is above 4 sigma of the logit error must agree >= 99.9 %;
    # over ALL pixels (incl. near-ties of random-weight logits) the floor is what the rounding-only oracle reaches.
    am, ar = lgc.argmax(1), lrc.argmax(1)
    srt = np.sort(lrc, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    clear = margin > 4 * float(np.sqrt(((lgc - lrc) ** 2).mean()))          # top-2 margin above 4 sigma of the logit error
    assert clear.mean() > 0.15
    assert (am == ar)[clear].mean() >= 0.999
    agree_emu = float((le.numpy().argmax(1) == ar).mean())
    assert (am == ar).mean() >= agree_emu - 0.02, ((am == ar).mean(), agree_emu)     # near-ties flip under ANY bf16 storage
    assert (am == g["deeplabwc_argmax"])[clear].mean() >= 0.999
    # identical integer confusion matrices given identical argmax maps (evaluation/metrics.py:4-23)
    conf = torch.zeros(11, 11, dtype=torch.int64, device="cuda")
    hip.confusion_accumulate(torch.from_numpy(ar).cuda(), tgt.cuda(), 11, 255, conf)
    assert np.array_equal(conf.cpu().numpy(), ol.confusion_matrix(ar, tgt.numpy(), 11))
    pr = dict(ref.named_parameters())
    rows, low = [], []
    pe = dict(emu.named_parameters())
    for name, p in net.named_parameters():
        if "pixel_feature" in name:
            assert p.grad is None
            continue
        if name == "classifier.text_embeddings":
            continue
        assert p.grad is not None, name
        if name.endswith(".bias") and not name.endswith(("bn1.bias", "bn2.bias", "bn3.bias", "1.bias", "2.bias")):
            continue
        gg, gr, ge = p.grad.cpu().numpy(), pr[name].grad.numpy(), pe[name].grad.numpy()
        c, c_emu, c_ge = cos(gg, gr), cos(ge, gr), cos(gg, ge)
        rows.append((name, c, c_emu, c_ge))
        # (1) against the fp32 oracle the HIP gradient may be only as far as bf16 storage alone puts the oracle itself
        if c < min(0.98, c_emu - 0.15):
            low.append((name, round(c, 4), round(c_emu, 4)))
    c_emu_of = {r[0]: r[2] for r in rows}
    rows.sort(key=lambda r: r[1])
    print("deeplab wc grads: (name, cos gpu~fp32, cos bf16emu~fp32, cos gpu~bf16emu), 8 worst:", [(n, round(a, 3), round(b, 3), round(c, 3)) for n, a, b, c in rows[:8]])
    print("deeplab wc grads: median cos gpu~fp32 %.4f, bf16emu~fp32 %.4f, gpu~bf16emu %.4f" % tuple(float(np.median([r[k] for r in rows])) for k in (1, 2, 3)))
    assert not low, low[:12]
    # (2) the two bf16-storage implementations are statistically indistinguishable: same median distance from the truth
    assert abs(float(np.median([r[1] for r in rows])) - float(np.median([r[2] for r in rows]))) < 0.05
    from tests.test_oracle_nets_golden import WC_GRADS
    named = dict(net.named_parameters())
    for name in WC_GRADS:
        got = named[name].grad.cpu().numpy()
        key = "deeplabwc_grad_" + name
        refv = g[key] if key in g else g[key + "__sub"]
        got = got if key in g else compact(got)[0]
        # the REFERENCE's own gradient (golden): same calibrated bound as against the oracle
        if not np.any(refv):           # the strided golden sample can land on taps that are out of range for every pixel (rate 18 at 14x20)
            assert not np.any(got), name
            continue
        assert cos(got, refv) >= min(0.98, c_emu_of[name] - 0.15), (name, cos(got, refv), c_emu_of[name])
    print("deeplab wc: argmax agreement", float((am == ar).mean()), "rounding-only", agree_emu, "clear fraction", float(clear.mean()))


def test_conv_train_fn_gradients():
    """conv2d_train (HIP forward, HIP dgrad incl. the zero-insert form for strided convs, HIP wgrad) vs PyTorch autograd of the same
    bf16-rounded operands: every gradient direction must agree to fp32-accumulation noise."""
    import torch.nn.functional as F
    from openess_amd import engine
    torch.manual_seed(0)
    for (Cin, Cout, k, st, pad, dil, H, W) in ((256, 512, 3, 1, 1, 1, 6, 8), (512, 11, 1, 1, 0, 1, 6, 8),
                                               (64, 64, 3, 2, 1, 1, 16, 24), (2048, 256, 3, 1, 6, 6, 6, 8),
                                               (32, 256, 1, 1, 0, 1, 20, 28),
                                               (64, 128, 3, 2, 1, 1, 15, 23),       # stride 2, odd sizes (output_padding 0)
                                               (128, 256, 1, 2, 0, 1, 14, 21),      # 1x1 stride-2 downsample branch
                                               (32, 64, 5, 2, 2, 1, 18, 26),        # 5x5 stride 2
                                               (2048, 256, 3, 1, 12, 12, 28, 40),   # ASPP rate 12, non-degenerate (28x40 map)
                                               (2048, 256, 3, 1, 18, 18, 28, 40)):  # ASPP rate 18
        x = torch.randn(2, Cin, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        wgt = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
        bias = torch.randn(Cout, device="cuda")
        wp, bp, xp = torch.nn.Parameter(wgt.clone()), torch.nn.Parameter(bias.clone()), x.clone().requires_grad_(True)
        y = engine.conv2d_train(xp, wp, bp, engine.PackedWeight(), k, st, pad, dil)
        gy = torch.randn_like(y.float())
        y.backward(gy.to(y.dtype))
        xr, wr, br = x.float().clone().requires_grad_(True), wgt.bfloat16().float().clone().requires_grad_(True), bias.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, st, pad, dil)
        yr.backward(gy.bfloat16().float())
        assert relerr(y.float().detach().cpu().numpy(), yr.detach().cpu().numpy()) < 1e-2
        assert cos(xp.grad.float().cpu().numpy(), xr.grad.cpu().numpy()) > 0.9999
        assert cos(wp.grad.cpu().numpy(), wr.grad.cpu().numpy()) > 0.9999
        assert cos(bp.grad.cpu().numpy(), br.grad.cpu().numpy()) > 0.9999


@pytest.mark.parametrize("option,contr", [("frame2voxel", False), ("frame2voxel", True), ("frame2recon", True)])
def test_pretrain_step_matches_oracle(option, contr):
    """Two optimisation steps of the GPU step vs the CPU oracle step from identical weights and inputs."""
    from openess_amd.training.pretrain_step import PretrainStep
    torch.manual_seed(3)
    B, H, W, nwin = 2, 64, 96, 3
    st = PretrainStep(config_option=option, img_size=(H, W), nr_events_data=nwin, if_spatial_contrastive=contr,
                      superpixel_size=25, lr=1e-4)
    ref = OracleStep(option, 11, nwin, 5, contr, 25, lr=1e-4)
    for name, m in st.models_dict.items():
        fill_by_name(m, 100 + len(name))
        fill_by_name(ref.modules()[name], 100 + len(name), sorted(m.state_dict().keys()))
        damp_residual(m), damp_residual(ref.modules()[name])         # well-conditioned weights: bf16 storage is not amplified
    if option == "frame2recon":
        st.model_recon.classifier.ASPP.project[3].p = 0.0
        ref.model_recon.classifier.ASPP.project[3].p = 0.0
    ev = (torch.randn(B, nwin * 5, H, W) * (torch.rand(B, nwin * 5, H, W) > 0.7)).contiguous()
    frame = torch.rand(B, 3, H, W)
    pl = torch.randint(0, 11, (B, H, W))
    pl[0, :5] = 255
    sp = torch.randint(0, 25, (B, H // 8, W // 8)).repeat_interleave(8, 1).repeat_interleave(8, 2)
    first = ev if option == "frame2voxel" else frame
    S = int((sp + torch.arange(B)[:, None, None] * 25).max()) + 1
    for it in range(2):
        losses, _, tl = st.train_step((first.cuda(), None, frame.cuda(), pl.cuda(), sp.cuda(), S))
        lref, tref = ref.train_step((first, None, frame, pl, sp))
        for k in lref:
            # InfoNCE at T = 0.07 multiplies feature error by ~14: 5 % on the L2-normalised frame2voxel features; 10 % on
            # frame2recon's UN-normalised ASPP features, whose rounding-only error is already 7-9 % rms on this random-weight
            # net (measured against the fp32 oracle with bf16 rounding points in test_deeplab_well_conditioned_*); others 2 %.
            # Round 2 needed 15 % / 30 % here because the BatchNorm statistics were order-dependent fp32 atomics (run-to-run
            # spread 1-17 % on the second step); they are fixed-order double sums now (tests/test_hip_determinism.py) and the
            # numbers repeat: 10 % on the first step, and on the SECOND step of frame2recon 15 % (measured 13.8 %, the same value
            # every run: AdamW's first update moves every weight by +-lr whatever its gradient's size, so the weights whose
            # gradient is rounding noise on either side take different signs in the two pipelines, and the un-normalised ASPP
            # features carry that into the logits of the 25-way InfoNCE at T = 0.07).
            nce = (1.5e-1 if it == 1 else 1e-1) if option == 'frame2recon' else 5e-2
            rel = nce if k == 'contrastive_nce_loss' else 2e-2
            assert float(losses[k]) == pytest.approx(float(lref[k]), rel=rel), (it, k, float(losses[k]), float(lref[k]))


@pytest.mark.gpu
def test_pretrain_step_full_size_matches_oracle():
    """ONE sample x 4 sub-windows at the BASELINE geometry (440 x 640 crop, 100 k events per sub-window): HIP pre-training step vs
    the CPU oracle from identical weights -- Dice + CE within 2 % (measured 4e-5), InfoNCE within 5 % (measured 6e-4), student logits
    within 0.2 x their spread rms (measured 0.12: bf16 storage through 4 recurrent steps + decoder on random-init weights, whose
    logits are near-ties: the median top-2 margin is 1.7 x that error), per-pixel argmax equal on >= 99 % of the pixels whose oracle
    top-2 margin exceeds 4 x the rms logit error (measured 99.9 % on 23 % of the pixels; 82 % over all pixels).  bench.py reports
    the same comparison on the full 20-window sample as `parity_full_size`."""
    from tests import full_size_parity as fp
    ev, frame, pl, sp = fp.synthetic_sample(4, 100000)
    r = fp.compare(ev, frame, pl, sp, nwin=4)
    assert r["rel"] <= 2e-2, r
    assert r["nce_rel"] <= 5e-2, r
    assert r["logit_rel_rms_err"] <= 0.2, r
    assert r["argmax_agree_clear_margin"] >= 0.99 and r["clear_margin_pixels"] > 0.05, r
    assert r["argmax_agree"] >= 0.7, r


def test_run_reconstruction_cli_end_to_end(tmp_path):
    """e2vid/run_reconstruction.py mirror: events text file -> fixed-size windows -> HIP voxel grid -> recurrent E2VID with the
    decoder path -> PNG frames; the frames equal a hand-driven loop over the same windows with the fp32 oracle (image in [0, 255])."""
    from openess_amd.e2vid import run_reconstruction as rr
    from oracle import events as oe
    rng = np.random.default_rng(5)
    W, H, n = 48, 32, 6000
    t = np.sort(rng.uniform(0.0, 0.2, n))
    ev = np.stack([t, rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, 2, n)], 1)
    path = str(tmp_path / "events.txt")
    with open(path, "w") as f:
        f.write(f"{W} {H}\n")
        for r in ev:
            f.write("%.9f %d %d %d\n" % (r[0], r[1], r[2], r[3]))
    model = rr.load_model('random')
    fill_by_name(model, 11)
    frames = rr.reconstruct(path, model, str(tmp_path / "out"), window_size=2000)
    assert len(frames) == 3 and frames[0].shape == (H, W) and frames[0].dtype == np.uint8
    assert sorted(os.listdir(str(tmp_path / "out"))) == ['frame_0000000000.png', 'frame_0000000001.png', 'frame_0000000002.png', 'timestamps.txt']
    ref = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG, full=True).eval()
    fill_by_name(ref, 11, sorted(model.state_dict().keys()))
    st = None
    ev_read = np.loadtxt(path, skiprows=1)
    with torch.no_grad():
        for k in range(3):
            win = ev_read[k * 2000:(k + 1) * 2000]
            grid = torch.from_numpy(oe.e2vid_voxel_grid(win.copy(), 5, W, H))[None]
            img, st, _ = ref(on.event_preprocess(grid), st)
            want = (img[0, 0].clamp(0, 1) * 255.0).numpy()
            assert np.abs(frames[k].astype(np.float64) - want).max() <= 6.0, k          # 2e-2 of the [0, 1] image, in grey levels


def test_wavefront_schedule_equals_single_stream_order():
    """e2vid/wavefront.py: one HIP stream per ConvLSTM level, ordered by events, vs the single-stream order: same kernels on the
    same buffers -> the latents of 7 recurrent sub-windows are BIT-identical and a whole contrastive PretrainStep gives the same
    loss bits after two optimiser steps (no order-dependent arithmetic is left on the path: a difference here is a cross-stream
    hazard, not rounding)."""
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    from openess_amd.e2vid.wavefront import EncoderWavefront
    from openess_amd.training.pretrain_step import PretrainStep
    torch.manual_seed(1)
    m = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(m, 11)
    m.cuda()
    B, H, W, nwin = 2, 64, 96, 7
    ev = (torch.randn(B, nwin * 5, H, W) * (torch.rand(B, nwin * 5, H, W) > 0.7)).contiguous().cuda()
    outs = []
    for use in (False, True, True):
        rec = ImageReconstructor(m, H, W, 5, torch.device("cuda"))
        wf = EncoderWavefront("cuda", 3) if use else None
        if wf:
            wf.begin()
        for i in range(nwin):
            _, _, latent = rec.update_reconstruction(ev, channel_slice=(5 * i, 5), wavefront=wf)
        if wf:
            wf.end()
        torch.cuda.synchronize()
        outs.append({k: v.float().clone() for k, v in latent.items()})
    for k in (1, 2, 4, 8):
        for o in outs[1:]:
            assert torch.equal(o[k], outs[0][k]), k
    losses = []
    for use in (False, True):
        torch.manual_seed(3)
        st = PretrainStep(config_option="frame2voxel", img_size=(H, W), nr_events_data=nwin, if_spatial_contrastive=True,
                          superpixel_size=25, lr=1e-4, wavefront=use)
        for name, mod in st.models_dict.items():
            fill_by_name(mod, 100 + len(name))
            damp_residual(mod)
        frame = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(9)).cuda()
        pl = torch.randint(0, 11, (B, H, W), generator=torch.Generator().manual_seed(9)).cuda()
        sp = torch.randint(0, 25, (B, H // 8, W // 8), generator=torch.Generator().manual_seed(9)).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda()
        for _ in range(2):
            ls, _, _ = st.train_step((ev, None, frame, pl, sp, B * 25))
        losses.append({k: float(v) for k, v in ls.items()})
    assert losses[1] == losses[0], losses


def test_skewed_schedule_with_four_encoders_equals_plain_order():
    """num_encoders = 4 (the full-size E2VID of e2vid/model/model.py): a call of the skewed schedule has four independent ConvLSTM
    steps -- one grouped launch of three plus one single launch -- and three deeper encoder convs (launched one by one); latents
    and cell states are bit-identical to the plain order over a sequence longer than the skew."""
    from openess_amd.e2vid.image_reconstructor import ImageReconstructor
    from openess_amd.e2vid.model.model import E2VIDRecurrent
    cfg = dict(E2VID_LIGHTWEIGHT_CONFIG, num_encoders=4)
    m = E2VIDRecurrent(cfg).eval()
    fill_by_name(m, 5)
    m.cuda()
    torch.manual_seed(3)
    ev = (torch.randn(1, 35, 64, 96, device="cuda") * (torch.rand(1, 35, 64, 96, device="cuda") > 0.7)).contiguous()
    out = {}
    for skew in (False, True):
        r = ImageReconstructor(m, 64, 96, 5, torch.device("cuda"))
        r.skew = skew
        for i in range(7):
            _, states, lat = r.update_reconstruction(ev, channel_slice=(5 * i, 5), need_latents=(i == 6))
        out[skew] = ({k: v.clone() for k, v in lat.items()}, [s_['cell'].clone() for s_ in states])
    assert sorted(out[True][0]) == [1, 2, 4, 8, 16]
    for k in out[True][0]:
        assert torch.equal(out[True][0][k], out[False][0][k]), k
    for a_, b_ in zip(out[True][1], out[False][1]):
        assert torch.equal(a_, b_)
