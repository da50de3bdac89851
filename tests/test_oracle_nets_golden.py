"""Pin oracle/nets.py against golden vectors produced by the reference's own modules (seeded weights by
parameter name).  CPU only, fp32 on both sides."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle import nets as on
from oracle.step import E2VID_LIGHTWEIGHT_CONFIG
from tests.synth import check_compact, fill_by_name

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "nets.npz")))


@pytest.fixture(scope="module")
def keys():
    return json.load(open(os.path.join(GOLDEN, "nets_keys.json")))


def test_e2vid_recurrent_latents(g, keys):
    torch.set_num_threads(4)
    m = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
    fill_by_name(m, 11, keys["e2vid"])
    ev = torch.from_numpy(g["e2vid_events"])
    states = None
    with torch.no_grad():
        for i in range(3):
            _, states, latent = m(on.event_preprocess(ev[:, 5 * i:5 * i + 5]), states)
    for k, v in latent.items():
        check_compact(g, f"e2vid_latent{k}", v.numpy(), rtol=1e-4, atol=1e-5)


def test_e2vid_full_reconstruction_image(g, keys):
    """Offline reconstruction path: residual blocks, ConvTranspose2d decoders, skip sums, pred + sigmoid
    (e2vid/model/unet.py:160-170) -- the oracle's image after 3 recurrent steps equals the reference's."""
    torch.set_num_threads(4)
    m = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG, full=True).eval()
    assert sorted(m.state_dict().keys()) == keys["e2vid"]              # every key of the reference module, nothing else
    fill_by_name(m, 11, keys["e2vid"])
    ev = torch.from_numpy(g["e2vid_events"])
    states = None
    with torch.no_grad():
        for i in range(3):
            img, states, _ = m(on.event_preprocess(ev[:, 5 * i:5 * i + 5]), states)
    np.testing.assert_allclose(img.numpy(), g["e2vid_img"], rtol=1e-4, atol=1e-5)


def test_semseg_e2vid_forward_and_grads(g, keys):
    torch.set_num_threads(4)
    net = on.SemSegE2VID(256, 11)
    fill_by_name(net, 12, keys["semseg"])
    net.train()
    lat = {k: torch.from_numpy(g[f"semseg_lat{k}"]) for k in (1, 2, 4, 8)}
    tgt = torch.from_numpy(g["semseg_target"])
    pred, x256 = net(lat)
    loss = ol.task_loss(pred[1], tgt, 11)
    loss.backward()
    check_compact(g, "semseg_logits", pred[1].detach().numpy(), 1e-4, 1e-5)
    check_compact(g, "semseg_x256", x256.detach().numpy(), 1e-4, 1e-5)
    check_compact(g, "semseg_out4", pred[4].detach().numpy(), 1e-4, 1e-5)
    assert loss.item() == pytest.approx(float(g["semseg_loss"]), rel=1e-5)
    params = dict(net.named_parameters())
    for name in ("decoder_ch512.0.weight", "decoder_ch256.0.bias", "decoder_scale_1.0.model.0.weight",
                 "decoder_scale_2.1.model.0.weight", "decoder_scale_4.0.model.0.weight"):
        check_compact(g, "semseg_grad_" + name, params[name].grad.numpy(), 2e-3, 1e-7)


def test_teacher(g, keys):
    torch.set_num_threads(4)
    t = on.DilationFeatureExtractor()
    fill_by_name(t.encoder, 13, keys["teacher_encoder"])
    fill_by_name(t.decoder[0], 14)
    t.train()
    with torch.no_grad():
        feat = t(torch.from_numpy(g["teacher_img"]))
    check_compact(g, "teacher_feat", feat.numpy(), 1e-3, 1e-5)
    np.testing.assert_allclose(t.encoder.bn1.running_mean.numpy(), g["teacher_bn1_running_mean_after"], rtol=1e-5, atol=1e-6)


def test_deeplab(g, keys):
    torch.set_num_threads(4)
    net = on.DeepLabV3(11, 32)
    fill_by_name(net, 15, keys["deeplab"])
    img = torch.from_numpy(g["deeplab_img"])
    net.eval()
    with torch.no_grad():
        lg, ft = net(img)
    check_compact(g, "deeplab_eval_logits", lg.numpy(), 1e-4, 1e-5)
    check_compact(g, "deeplab_eval_feats", ft.numpy(), 1e-4, 1e-5)
    net.train()
    net.classifier.ASPP.project[3].p = 0.0
    lg, _ = net(img)
    loss = ol.task_loss(lg, torch.from_numpy(g["deeplab_target"]), 11)
    loss.backward()
    check_compact(g, "deeplab_train_logits", lg.detach().numpy(), 1e-3, 1e-4)
    assert loss.item() == pytest.approx(float(g["deeplab_train_loss"]), rel=1e-4)
    check_compact(g, "deeplab_grad_classifier.classifier.0.weight", net.classifier.classifier[0].weight.grad.numpy(), 5e-3, 1e-6)


WC_GRADS = ("backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer2.3.conv2.weight",
            "backbone.layer3.5.conv3.weight", "backbone.layer4.0.downsample.0.weight", "backbone.layer4.2.conv2.weight",
            "backbone.layer4.2.bn3.weight", "classifier.ASPP.convs.0.0.weight", "classifier.ASPP.convs.1.0.weight",
            "classifier.ASPP.convs.2.0.weight", "classifier.ASPP.convs.3.0.weight", "classifier.ASPP.convs.4.1.weight",
            "classifier.ASPP.project.0.weight", "classifier.ASPP.project.1.bias", "classifier.classifier.0.weight",
            "classifier.classifier.1.weight")


def test_teacher_well_conditioned(g, keys):
    """Same teacher, residual branches damped (tests/synth.py:damp_residual): the oracle reproduces the reference."""
    from tests.synth import damp_residual
    torch.set_num_threads(4)
    t = on.DilationFeatureExtractor()
    fill_by_name(t.encoder, 13, keys["teacher_encoder"])
    fill_by_name(t.decoder[0], 14)
    damp_residual(t.encoder)
    t.train()
    with torch.no_grad():
        feat = t(torch.from_numpy(g["teacherwc_img"]))
    check_compact(g, "teacherwc_feat", feat.numpy(), 1e-3, 1e-5)


def test_deeplab_well_conditioned_224x320(g, keys):
    """DeepLabv3 at 4x3x224x320 (14x20 OS16 map, ASPP rates 6/12/18 non-degenerate), train forward + backward of 16 tensors."""
    from tests.synth import damp_residual, wc_image
    torch.set_num_threads(8)
    net = on.DeepLabV3(11, 32)
    fill_by_name(net, 15, keys["deeplab"])
    damp_residual(net)
    net.train()
    net.classifier.ASPP.project[3].p = 0.0
    lg, ft = net(torch.from_numpy(wc_image()))
    tgt = torch.from_numpy(g["deeplabwc_target"]).long()
    loss = ol.task_loss(lg, tgt, 11)
    loss.backward()
    check_compact(g, "deeplabwc_logits", lg.detach().numpy(), 1e-3, 1e-4)
    check_compact(g, "deeplabwc_feats", ft.detach().numpy(), 1e-3, 1e-4)
    assert loss.item() == pytest.approx(float(g["deeplabwc_loss"]), rel=1e-5)
    assert float((lg.argmax(1).numpy() == g["deeplabwc_argmax"]).mean()) > 0.9999
    named = dict(net.named_parameters())
    for name in WC_GRADS:
        check_compact(g, "deeplabwc_grad_" + name, named[name].grad.numpy(), 5e-3, 1e-6)


def _bf16_rounding_points(module):
    """Round the output of every conv and every BatchNorm to bf16 -- the storage points of the MI355X pipeline --
    while all arithmetic stays the oracle's fp32."""
    return [m.register_forward_hook(lambda mod, i, o: o.bfloat16().float())
            for m in module.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d))]


def test_bf16_storage_sensitivity_of_the_oracle_itself(keys):
    """Separates rounding from bugs WITHOUT the GPU: the fp32 oracle, with nothing changed but bf16 rounding of the
    stored activations, moves as far from itself as the HIP pipeline does (tests/test_hip_nets.py:test_teacher_forward)
    on the chaotic random-weight teacher -- and stays within 1e-3 on the well-conditioned one."""
    from tests.synth import damp_residual
    torch.set_num_threads(8)
    torch.manual_seed(5)
    img = torch.rand(2, 3, 96, 128)
    got = {}
    for damp in (None, 0.25):
        t = on.DilationFeatureExtractor()
        fill_by_name(t.encoder, 13, keys["teacher_encoder"])
        fill_by_name(t.decoder[0], 14)
        if damp:
            damp_residual(t.encoder, damp)
        t.train()
        with torch.no_grad():
            a = t(img)
            _bf16_rounding_points(t)
            b = t(img.bfloat16().float())
        got[damp] = float((a * b).sum(1).mean())
    assert 0.85 < got[None] < 0.97, got            # rounding alone explains the end-to-end cosine of the random-weight net
    assert got[0.25] > 0.999, got
