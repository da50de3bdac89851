"""Oracle (test infrastructure): plain PyTorch fp32 CPU restatement of the networks on the hot path.

Each class keeps the reference's state_dict keys so that one seeded state_dict fills the reference
module (in tests/golden/gen_golden_nets.py), this oracle and the product module alike.  Pinned by
tests/test_oracle_golden.py against golden vectors produced by the reference modules themselves.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# E2VID front end: e2vid/model/submodules.py:7-31 (ConvLayer), :96-115 (RecurrentConvLayer),
# :175-214 (ConvLSTM); e2vid/model/unet.py:118-170 (UNetRecurrent); e2vid/model/model.py:69-100.
# Only the layers that produce `latent` (unet.py:163) are restated; the rest of the reference's
# state_dict (resblocks / decoders / pred) is ignored with strict=False.
# ------------------------------------------------------------------------------------------------
class ConvLayer(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        self.conv2d = nn.Conv2d(cin, cout, k, stride, padding, bias=(norm != 'BN'))
        self.activation = activation
        self.norm = norm
        if norm == 'BN':
            self.norm_layer = nn.BatchNorm2d(cout)

    def forward(self, x):
        out = self.conv2d(x)
        if self.norm == 'BN':
            out = self.norm_layer(out)
        return torch.relu(out) if self.activation == 'relu' else out


class ConvLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2)

    def forward(self, x, prev_state=None):
        if prev_state is None:
            z = torch.zeros(x.shape[0], self.hidden_size, *x.shape[2:], dtype=x.dtype)
            prev_state = (z, z)
        prev_hidden, prev_cell = prev_state
        gates = self.Gates(torch.cat((x, prev_hidden), 1))
        in_gate, remember_gate, out_gate, cell_gate = gates.chunk(4, 1)        # submodules.py:205
        cell = torch.sigmoid(remember_gate) * prev_cell + torch.sigmoid(in_gate) * torch.tanh(cell_gate)
        hidden = torch.sigmoid(out_gate) * torch.tanh(cell)
        return hidden, cell


class RecurrentConvLayer(nn.Module):
    def __init__(self, cin, cout, norm):
        super().__init__()
        self.conv = ConvLayer(cin, cout, 5, 2, 2, 'relu', norm)
        self.recurrent_block = ConvLSTM(cout, cout, 3)

    def forward(self, x, prev_state):
        x = self.conv(x)
        state = self.recurrent_block(x, prev_state)
        return state[0], state


class ResidualBlock(nn.Module):
    """e2vid/model/submodules.py:140-172."""

    def __init__(self, c, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=(norm != 'BN'))
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=(norm != 'BN'))
        self.norm = norm
        if norm == 'BN':
            self.bn1, self.bn2 = nn.BatchNorm2d(c), nn.BatchNorm2d(c)

    def forward(self, x):
        out = self.conv1(x)
        out = torch.relu(self.bn1(out) if self.norm == 'BN' else out)
        out = self.conv2(out)
        out = self.bn2(out) if self.norm == 'BN' else out
        return torch.relu(out + x)


class TransposedConvLayer(nn.Module):
    """e2vid/model/submodules.py:34-62: ConvTranspose2d(k, stride 2, padding, output_padding 1) -> BN -> relu."""

    def __init__(self, cin, cout, k, padding, norm):
        super().__init__()
        self.transposed_conv2d = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=padding, output_padding=1, bias=(norm != 'BN'))
        self.norm = norm
        if norm == 'BN':
            self.norm_layer = nn.BatchNorm2d(cout)

    def forward(self, x):
        out = self.transposed_conv2d(x)
        return torch.relu(self.norm_layer(out) if self.norm == 'BN' else out)


class UNetRecurrentEncoder(nn.Module):
    """`full=True` adds the residual blocks, decoders and prediction layer (unet.py:160-170, skip_type 'sum',
    use_upsample_conv False): the offline reconstruction path.  The training path only needs the latents."""

    def __init__(self, num_bins=5, num_encoders=3, base=32, norm='BN', full=False, num_residual_blocks=2):
        super().__init__()
        self.num_encoders, self.full = num_encoders, full
        self.head = ConvLayer(num_bins, base, 5, 1, 2)
        self.encoders = nn.ModuleList([RecurrentConvLayer(base * 2 ** i, base * 2 ** (i + 1), norm) for i in range(num_encoders)])
        if full:
            cmax = base * 2 ** num_encoders
            self.resblocks = nn.ModuleList([ResidualBlock(cmax, norm) for _ in range(num_residual_blocks)])
            self.decoders = nn.ModuleList([TransposedConvLayer(base * 2 ** (i + 1), base * 2 ** i, 5, 2, norm)
                                           for i in reversed(range(num_encoders))])
            self.pred = ConvLayer(base, 1, 1, activation=None, norm=norm)

    def forward(self, x, prev_states):
        x = self.head(x)
        head = x
        if prev_states is None:
            prev_states = [None] * self.num_encoders
        blocks, states = [], []
        for i, enc in enumerate(self.encoders):
            x, st = enc(x, prev_states[i])
            blocks.append(x)
            states.append(st)
        latent = {1: head}
        for i, b in enumerate(blocks):
            latent[2 ** (i + 1)] = b
        img = None
        if self.full:
            for rb in self.resblocks:
                x = rb(x)
            for i, dec in enumerate(self.decoders):
                x = dec(x + blocks[self.num_encoders - i - 1])               # skip_sum
            img = torch.sigmoid(self.pred(x + head))
        return img, states, latent


class E2VIDRecurrent(nn.Module):
    def __init__(self, config, full=False):
        super().__init__()
        self.num_encoders = int(config.get('num_encoders', 4))
        assert not full or (config.get('skip_type', 'sum') == 'sum' and not config.get('use_upsample_conv', True))
        self.unetrecurrent = UNetRecurrentEncoder(int(config['num_bins']), self.num_encoders,
                                                  int(config.get('base_num_channels', 32)), config.get('norm'), full=full,
                                                  num_residual_blocks=int(config.get('num_residual_blocks', 2)))

    def forward(self, x, prev_states):
        return self.unetrecurrent(x, prev_states)


def event_preprocess(events):
    """EventPreprocessor.__call__, e2vid/utils/inference_utils.py:78-85 (torch ops exactly as written)."""
    nonzero_ev = (events != 0)
    num_nonzeros = nonzero_ev.sum()
    if num_nonzeros > 0:
        mean = events.sum() / num_nonzeros
        stddev = torch.sqrt((events ** 2).sum() / num_nonzeros - mean ** 2)
        events = nonzero_ev.float() * (events - mean) / stddev
    return events


# ------------------------------------------------------------------------------------------------
# SemSegE2VID: models/style_networks.py:9-198, ReLUINSConv2d :252-263, INSResBlock :266-289
# ------------------------------------------------------------------------------------------------
class ReLUINSConv2d(nn.Module):
    def __init__(self, n_in, n_out, k, stride, padding=0):
        super().__init__()
        self.model = nn.Sequential(nn.Conv2d(n_in, n_out, k, stride, padding, bias=True), nn.InstanceNorm2d(n_out, affine=False),
                                   nn.ReLU(inplace=False))

    def forward(self, x):
        return self.model(x)


class INSResBlock(nn.Module):
    def __init__(self, inplanes, planes):
        super().__init__()
        self.model = nn.Sequential(nn.Conv2d(inplanes, planes, 3, 1, 1), nn.InstanceNorm2d(planes), nn.ReLU(inplace=False),
                                   nn.Conv2d(planes, planes, 3, 1, 1), nn.InstanceNorm2d(planes))

    def forward(self, x):
        return self.model(x) + x


class SemSegE2VID(nn.Module):
    def __init__(self, input_c, output_c):
        super().__init__()
        tch = input_c
        self.register_buffer('text_embeddings', torch.randn(output_c, 512))
        self.decoder_scale_1 = nn.Sequential(*([INSResBlock(tch, tch) for _ in range(5)] + [ReLUINSConv2d(tch, tch // 2, 3, 1, 1)]))
        self.decoder_scale_2 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1), ReLUINSConv2d(tch // 2, tch // 4, 3, 1, 1))
        tch //= 2
        self.decoder_scale_3 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1), ReLUINSConv2d(tch // 2, tch // 2, 3, 1, 1))
        tch //= 2
        self.decoder_scale_4 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1))
        tch //= 2
        self.decoder_scale_5 = nn.Sequential(nn.Conv2d(tch, output_c, 1))
        self.decoder_ch256 = nn.Sequential(nn.Conv2d(tch, 256, 1))
        self.decoder_ch512 = nn.Sequential(nn.Conv2d(256, 512, 1))

    def forward(self, d):
        sz_in = d[1].shape[3]
        x = d[8]
        out = {8: x}
        x = self.decoder_scale_1(x)
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = torch.cat([x, d[4]], 1)
        x = self.decoder_scale_2(x)
        out[sz_in // x.shape[3]] = x
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = torch.cat([x, d[2]], 1)
        x = self.decoder_scale_3(x)
        out[sz_in // x.shape[3]] = x
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = self.decoder_scale_4(x)
        x_ch256 = self.decoder_ch256(x)
        x = self.decoder_ch512(x_ch256)
        x = F.conv2d(x, self.text_embeddings[:, :, None, None])
        out[sz_in // x.shape[3]] = x
        return out, x_ch256


# ------------------------------------------------------------------------------------------------
# ResNet-50: models/_resnet.py:74-209 ; teacher: models/image_model.py:90-143 ;
# DeepLabv3: models/deeplabv3.py:86-189, 295-348
# ------------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return F.relu(out + identity)


class ResNet50(nn.Module):
    def __init__(self, replace_stride_with_dilation=(False, False, False), with_fc=False):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, 6, 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, 3, 2, replace_stride_with_dilation[2])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class DilationFeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = ResNet50((True, True, True))
        for p in self.encoder.parameters():
            p.requires_grad = False
        self.decoder = nn.Sequential(nn.Conv2d(2048, 256, 1), nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True))

    def forward(self, x):
        return F.normalize(self.decoder(self.encoder(x)), p=2, dim=1)


class ASPP(nn.Module):
    def __init__(self, cin, rates):
        super().__init__()
        def cbr(k, d):
            return nn.Sequential(nn.Conv2d(cin, 256, k, padding=0 if k == 1 else d, dilation=d, bias=False), nn.BatchNorm2d(256), nn.ReLU())
        mods = [cbr(1, 1)] + [cbr(3, r) for r in rates]
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, 256, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU()))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * 256, 256, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout(0.1))

    def forward(self, x):
        res = [c(x) for c in self.convs[:4]]
        res.append(F.interpolate(self.convs[4](x), size=x.shape[-2:], mode='bilinear', align_corners=False))
        return self.project(torch.cat(res, 1))


class DeepLabHead(nn.Module):
    def __init__(self, K, rates):
        super().__init__()
        self.ASPP = ASPP(2048, rates)
        self.pixel_feature = nn.Conv2d(256, 512, 3, padding=1, bias=False)
        self.classifier = nn.Sequential(nn.Conv2d(256, 512, 3, padding=1, bias=False), nn.BatchNorm2d(512), nn.ReLU())
        self.register_buffer('text_embeddings', torch.randn(K, 512))

    def forward(self, feat):
        feature = self.ASPP(feat)
        return F.conv2d(self.classifier(feature), self.text_embeddings[:, :, None, None]), feature


class DeepLabV3(nn.Module):
    def __init__(self, K, output_stride=32):
        super().__init__()
        if output_stride == 8:
            rswd, rates = (False, True, True), (12, 24, 36)
        else:
            rswd, rates = (False, False, True), (6, 12, 18)
        self.backbone = ResNet50(rswd)
        self.classifier = DeepLabHead(K, rates)

    def forward(self, x):
        size = x.shape[-2:]
        logits, feats = self.classifier(self.backbone(x))
        return (F.interpolate(logits, size=size, mode='bilinear', align_corners=False),
                F.interpolate(feats, size=size, mode='bilinear', align_corners=False))


# ------------------------------------------------------------------------------------------------
# bf16-STORAGE emulation (test infrastructure).  The MI355X pipeline keeps fp32 master weights and fp32 accumulation but
# STORES activations, activation gradients and the MFMA weight operand in bf16.  `emulate_bf16_storage(module)` gives the
# fp32 oracle exactly those rounding points -- every conv / BatchNorm / residual-block output, every gradient flowing back
# through them, and the conv weights -- and nothing else (arithmetic stays ATen fp32 on the CPU).  It is what separates
# "rounding" from "bug": a random-weight train-mode-BatchNorm network amplifies bf16 storage noise (the common mode the next
# BatchNorm removes carries most of the magnitude the rounding is relative to), and the HIP path has to track THIS oracle
# closely, and the plain fp32 oracle only as closely as this one does.
# ------------------------------------------------------------------------------------------------
def _r16(x):
    return x.bfloat16().float()


def emulate_bf16_storage(module, weights=True, backward=True):
    with torch.no_grad():
        for m in module.modules():
            if weights and isinstance(m, nn.Conv2d):
                m.weight.copy_(_r16(m.weight))
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.InstanceNorm2d, Bottleneck, INSResBlock)):
            m.register_forward_hook(lambda mod, i, o: _r16(o))
            if backward:
                m.register_full_backward_hook(lambda mod, gi, go: tuple(None if g is None else _r16(g) for g in gi))
    return module
