#!/usr/bin/env python3
"""Golden vectors for the NETWORK pieces, produced by running the reference modules (imported from
/root/reference) with seeded weights keyed by parameter name (tests/synth.py: fill_by_name).  Only
inputs, expected outputs and the sorted key lists are stored -- no weights, no code.
Run through gen_golden.py:  python tests/golden/gen_golden.py nets"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from tests.synth import damp_residual, fill_by_name, wc_image  # noqa: E402


def _ref_models():
    if "models" not in sys.modules or not hasattr(sys.modules["models"], "__path__") or \
            sys.modules["models"].__path__ != [os.path.join(REF, "models")]:
        pkg = types.ModuleType("models")
        pkg.__path__ = [os.path.join(REF, "models")]
        sys.modules["models"] = pkg
    for stub in ("torchvision", "torchvision.models"):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    import models.style_networks as sn
    import models.deeplabv3 as dl
    import models._resnet as rn
    return sn, dl, rn


def gen_nets():
    import importlib.util
    torch.set_num_threads(4)
    torch.manual_seed(1205)          # every torch.randn / torch.rand input below is reproducible from this recipe
    out, keys = {}, {}
    sn, dl, rn = _ref_models()
    rng = np.random.default_rng(77)
    spec = importlib.util.spec_from_file_location("ref_data_util", os.path.join(REF, "datasets/data_util.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)

    # ---- G7: E2VIDRecurrent, 3 recurrent steps (eval mode), EventPreprocessor maths = normalize_voxel_grid
    from e2vid.model.model import E2VIDRecurrent
    cfg = {'num_bins': 5, 'skip_type': 'sum', 'recurrent_block_type': 'convlstm', 'num_encoders': 3,
           'base_num_channels': 32, 'num_residual_blocks': 2, 'use_upsample_conv': False, 'norm': 'BN'}
    m = E2VIDRecurrent(cfg).eval()
    keys["e2vid"] = sorted(m.state_dict().keys())
    fill_by_name(m, 11)
    B, H, W = 2, 32, 48
    ev = (rng.normal(0, 1, (B, 15, H, W)) * (rng.uniform(0, 1, (B, 15, H, W)) > 0.7)).astype(np.float32)
    states = None
    with torch.no_grad():
        for i in range(3):
            x = du.normalize_voxel_grid(torch.from_numpy(ev[:, 5 * i:5 * i + 5].copy()))
            img, states, latent = m(x, states)
    out["e2vid_img"] = img.numpy()                   # full UNetRecurrent forward: resblocks, transposed-conv decoders, pred, sigmoid
    out["e2vid_events"] = ev
    for k, v in latent.items():
        out[f"e2vid_latent{k}"] = v.numpy()

    # ---- G8: SemSegE2VID forward + TaskLoss + selected gradients
    from utils.loss_functions import TaskLoss
    K = 11
    net = sn.SemSegE2VID(256, K, skip_connect=True, skip_type='concat', text_embeddings_path=None)
    keys["semseg"] = sorted(net.state_dict().keys())
    fill_by_name(net, 12)
    net.train()
    lat = {1: torch.randn(B, 32, H, W), 2: torch.randn(B, 64, H // 2, W // 2), 4: torch.randn(B, 128, H // 4, W // 4),
           8: torch.randn(B, 256, H // 8, W // 8)}
    tgt = torch.from_numpy(rng.integers(0, K, (B, H, W))).long()
    tgt[0, :3] = 255
    pred, x256 = net(lat)
    loss = TaskLoss(losses=['dice', 'cross_entropy'], num_classes=K, ignore_index=255)(pred[1], tgt)
    loss.backward()
    for k, v in lat.items():
        out[f"semseg_lat{k}"] = v.numpy()
    out["semseg_target"], out["semseg_logits"], out["semseg_x256"] = tgt.numpy(), pred[1].detach().numpy(), x256.detach().numpy()
    out["semseg_out4"], out["semseg_loss"] = pred[4].detach().numpy(), loss.detach().numpy()
    for name in ("decoder_ch512.0.weight", "decoder_ch256.0.bias", "decoder_scale_1.0.model.0.weight",
                 "decoder_scale_2.1.model.0.weight", "decoder_scale_4.0.model.0.weight"):
        out["semseg_grad_" + name] = dict(net.named_parameters())[name].grad.numpy()

    # ---- G10: dilated ResNet-50 teacher (reference _resnet.ResNet == torchvision architecture) + decoder
    enc = rn.ResNet(rn.Bottleneck, [3, 4, 6, 3], replace_stride_with_dilation=[True, True, True])
    del enc.fc
    keys["teacher_encoder"] = sorted(enc.state_dict().keys())
    fill_by_name(enc, 13)
    dec = torch.nn.Conv2d(2048, 256, 1)
    fill_by_name(dec, 14)
    img = torch.rand(B, 3, 32, 48)
    enc.train()
    with torch.no_grad():
        x = enc.maxpool(enc.relu(enc.bn1(enc.conv1(img))))
        x = enc.layer4(enc.layer3(enc.layer2(enc.layer1(x))))
        feat = torch.nn.functional.normalize(
            torch.nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True)(dec(x)), p=2, dim=1)
    out["teacher_img"], out["teacher_feat"], out["teacher_enc_out"] = img.numpy(), feat.numpy(), x.numpy()
    out["teacher_bn1_running_mean_after"] = enc.bn1.running_mean.numpy().copy()

    # ---- G9: deeplabv3_resnet50 (output_stride 32 -> else-branch), eval forward and train forward + grads
    net = dl.deeplabv3_resnet50(num_classes=K, text_embeddings_path=None, output_stride=32, pretrained_backbone='')
    keys["deeplab"] = sorted(net.state_dict().keys())
    fill_by_name(net, 15)
    img = torch.rand(B, 3, 64, 96)
    net.eval()
    with torch.no_grad():
        lg, ft = net(img)
    out["deeplab_img"], out["deeplab_eval_logits"], out["deeplab_eval_feats"] = img.numpy(), lg.numpy(), ft.numpy()
    net.train()
    net.classifier.ASPP.project[3].p = 0.0            # dropout RNG streams differ across implementations
    lg, ft = net(img)
    tgt = torch.from_numpy(rng.integers(0, K, (B, 64, 96))).long()
    loss = TaskLoss(losses=['dice', 'cross_entropy'], num_classes=K, ignore_index=255)(lg, tgt)
    loss.backward()
    out["deeplab_target"], out["deeplab_train_logits"], out["deeplab_train_loss"] = tgt.numpy(), lg.detach().numpy(), loss.detach().numpy()
    out["deeplab_grad_classifier.classifier.0.weight"] = net.classifier.classifier[0].weight.grad.numpy()
    out["deeplab_grad_backbone.layer4.2.conv3.weight"] = dict(net.named_parameters())["backbone.layer4.2.conv3.weight"].grad.numpy()

    # ---- G10b: the same teacher on WELL-CONDITIONED weights (residual branches damped, tests/synth.py:damp_residual)
    enc = rn.ResNet(rn.Bottleneck, [3, 4, 6, 3], replace_stride_with_dilation=[True, True, True])
    del enc.fc
    fill_by_name(enc, 13)
    damp_residual(enc)
    enc.train()
    img = torch.rand(2, 3, 96, 128)
    with torch.no_grad():
        x = enc.maxpool(enc.relu(enc.bn1(enc.conv1(img))))
        x = enc.layer4(enc.layer3(enc.layer2(enc.layer1(x))))
        feat = torch.nn.functional.normalize(
            torch.nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True)(dec(x)), p=2, dim=1)
    out["teacherwc_img"], out["teacherwc_feat"], out["teacherwc_enc_out"] = img.numpy(), feat.numpy(), x.numpy()

    # ---- G9b: deeplabv3_resnet50 at 4x3x224x320 (OS16 map 14x20: 1120 samples per BatchNorm channel, ASPP rates
    #      6/12/18 all reach in-range off-centre taps), well-conditioned weights, train forward + backward
    net = dl.deeplabv3_resnet50(num_classes=K, text_embeddings_path=None, output_stride=32, pretrained_backbone='')
    fill_by_name(net, 15)
    damp_residual(net)
    net.train()
    net.classifier.ASPP.project[3].p = 0.0
    img = torch.from_numpy(wc_image())                 # tests/synth.py: regenerated by the consumers, not stored
    tgt = torch.from_numpy(rng.integers(0, K, (4, 28, 40))).long().repeat_interleave(8, 1).repeat_interleave(8, 2)
    tgt[1, :9] = 255
    out["deeplabwc_target"] = tgt.numpy().astype(np.uint8)
    lg, ft = net(img)
    loss = TaskLoss(losses=['dice', 'cross_entropy'], num_classes=K, ignore_index=255)(lg, tgt)
    loss.backward()
    out["deeplabwc_logits"], out["deeplabwc_feats"], out["deeplabwc_loss"] = lg.detach().numpy(), ft.detach().numpy(), loss.detach().numpy()
    out["deeplabwc_argmax"] = lg.argmax(1).numpy().astype(np.uint8)
    named = dict(net.named_parameters())
    for name in ("backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer2.3.conv2.weight",
                 "backbone.layer3.5.conv3.weight", "backbone.layer4.0.downsample.0.weight", "backbone.layer4.2.conv2.weight",
                 "backbone.layer4.2.bn3.weight", "classifier.ASPP.convs.0.0.weight", "classifier.ASPP.convs.1.0.weight",
                 "classifier.ASPP.convs.2.0.weight", "classifier.ASPP.convs.3.0.weight", "classifier.ASPP.convs.4.1.weight",
                 "classifier.ASPP.project.0.weight", "classifier.ASPP.project.1.bias", "classifier.classifier.0.weight",
                 "classifier.classifier.1.weight"):
        out["deeplabwc_grad_" + name] = named[name].grad.numpy()
    from tests.synth import compact
    small = {}
    for k, v in out.items():
        v = np.asarray(v)
        if v.size > 20000 and not k.endswith(("_events", "_img", "_target", "_argmax")) and not k.startswith("semseg_lat"):
            sub, ssum, sabs = compact(v)
            small[k + "__sub"], small[k + "__sum"], small[k + "__abs"] = sub, ssum, sabs
            small[k + "__shape"] = np.array(v.shape)
        else:
            small[k] = v
    out = small
    np.savez_compressed(os.path.join(HERE, "nets.npz"), **out)
    json.dump(keys, open(os.path.join(HERE, "nets_keys.json"), "w"))
    print("nets.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "nets.npz")) // 1024, "KiB")


GROUPS = {"nets": gen_nets}
