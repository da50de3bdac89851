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
