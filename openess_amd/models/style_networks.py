"""Host-side mirror of models/style_networks.py: SemSegE2VID task decoder (reference :9-198),
ReLUINSConv2d (:252-263), INSResBlock (:266-289).  Same constructor arguments, state_dict keys and
return values; the convolutions run on the HIP MFMA kernel (forward + data gradient).

Deviation that changes no result (documented in DESIGN.md): decoder_ch256 -> decoder_ch512 ->
conv2d(text_embeddings) are three linear maps with nothing in between (style_networks.py:161-163),
so the logits are computed as ONE 1x1 conv with the composed operator T*W512*W256 (and composed
bias); the 512-channel full-resolution tensor (4.6 GB fp32 at DSEC B=8) is never materialised.
Autograd differentiates the composition, so every parameter receives the same gradient.
"""
import torch
import torch.nn as nn
import torch.nn.functional as f

from .. import engine, hip


def gaussian_weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 and classname.find('Conv') == 0:
        m.weight.data.normal_(0.0, 0.02)


class _HipConv2d(nn.Conv2d):
    """nn.Conv2d whose forward runs the MFMA kernel (class name starts with '_Hip' on purpose: the
    reference's gaussian_weights_init only touches classes whose name STARTS with 'Conv')."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._pw = engine.PackedWeight()

    def forward(self, x, out_f32=False):
        return engine.conv2d_train(x, self.weight, self.bias, self._pw, self.kernel_size[0], self.stride[0],
                                   self.padding[0], self.dilation[0], out_f32=out_f32)


def _conv(n_in, n_out, kernel_size, stride=1, padding=0, bias=True):
    m = _HipConv2d(n_in, n_out, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
    m.weight.data.normal_(0.0, 0.02)          # == gaussian_weights_init applied by the reference blocks
    return m


def _instance_norm(x, relu, residual=None):
    # nn.InstanceNorm2d defaults: affine=False, no running stats, eps=1e-5; fused [+ residual] [+ ReLU]
    return hip.instance_norm(x, relu=relu, residual=residual)


class ReLUINSConv2d(nn.Module):
    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super().__init__()
        self.model = nn.Sequential(_conv(n_in, n_out, kernel_size, stride, padding, bias=True),
                                   nn.InstanceNorm2d(n_out, affine=False), nn.ReLU(inplace=True))

    def forward(self, x):
        return _instance_norm(self.model[0](x), relu=True)


class INSResBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dropout=0.0):
        super().__init__()
        model = [_conv(inplanes, planes, 3, stride, 1), nn.InstanceNorm2d(planes), nn.ReLU(inplace=True),
                 _conv(planes, planes, 3, 1, 1), nn.InstanceNorm2d(planes)]
        if dropout > 0:
            model += [nn.Dropout(p=dropout)]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        out = _instance_norm(self.model[0](x), relu=True)
        if len(self.model) > 5:
            return self.model[5](_instance_norm(self.model[3](out), relu=False)) + x
        return _instance_norm(self.model[3](out), relu=False, residual=x)      # out += residual (:287)


def skip_concat(x1, x2):
    return torch.cat([x1, x2], dim=1)


def skip_sum(x1, x2):
    return x1 + x2


class SemSegE2VID(nn.Module):
    def __init__(self, input_c, output_c, skip_connect=False, skip_type='sum', input_index_map=False,
                 text_embeddings_path='', if_linear_probing=False, materialize_ch256=True):
        super().__init__()
        if not skip_connect or input_index_map:
            raise NotImplementedError("every shipped config uses skip_connect_task: True (style_networks.py:34)")
        self.skip_connect = skip_connect
        self.skip_type = skip_type
        self.apply_skip_connection = skip_sum if skip_type == 'sum' else skip_concat
        self.materialize_ch256 = materialize_ch256
        tch = input_c
        text_categories = output_c
        self.text_embeddings_path = text_embeddings_path
        if text_embeddings_path is None:
            self.text_embeddings = nn.Parameter(torch.zeros(text_categories, 512))
            nn.init.normal_(self.text_embeddings, mean=0.0, std=0.01)
        else:
            self.register_buffer('text_embeddings', torch.randn(text_categories, 512))
            if text_embeddings_path:
                # the reference hard-codes map_location='cuda' (style_networks.py:31)
                loaded = torch.load(text_embeddings_path, map_location='cpu')
                self.text_embeddings[:, :] = loaded[:, :]
        layers = [INSResBlock(tch, tch) for _ in range(5)]
        layers += [ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1)]
        self.decoder_scale_1 = nn.Sequential(*layers)
        self.decoder_scale_2 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1), ReLUINSConv2d(tch // 2, tch // 4, 3, 1, 1))
        tch = tch // 2
        self.decoder_scale_3 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1), ReLUINSConv2d(tch // 2, tch // 2, 3, 1, 1))
        tch = tch // 2
        self.decoder_scale_4 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, 3, 1, 1))
        tch = tch // 2
        # decoder_scale_5 exists in the reference's state_dict but is never used in the skip path (:61-63, :167)
        self.decoder_scale_5 = nn.Sequential(nn.Conv2d(tch, output_c, kernel_size=1, stride=1, padding=0))
        self.decoder_ch256 = nn.Sequential(_HipConv2d(tch, 256, kernel_size=1, stride=1, padding=0))
        self.decoder_ch512 = nn.Sequential(nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0))
        self.if_linear_probing = if_linear_probing
        if if_linear_probing:
            for mod in (self.decoder_scale_1, self.decoder_scale_2, self.decoder_scale_3, self.decoder_scale_4,
                        self.decoder_ch256, self.decoder_ch512):
                for p in mod.parameters():
                    p.requires_grad = False
            self.linear_probe = nn.Conv2d(text_categories, text_categories, 1)
        self._pw_head = engine.PackedWeight()

    def update_skip_dict(self, skips, x, sz_in):
        rem, scale = sz_in % x.shape[3], sz_in // x.shape[3]
        assert rem == 0
        skips[scale] = x

    def _composed_head(self):
        """T*W512*W256 and its bias (fp32, differentiable)."""
        w1 = self.decoder_ch256[0].weight.flatten(1)              # 256 x 32
        b1 = self.decoder_ch256[0].bias
        w2 = self.decoder_ch512[0].weight.flatten(1)              # 512 x 256
        b2 = self.decoder_ch512[0].bias
        t = self.text_embeddings.float()                          # K x 512
        tw2 = t @ w2
        return (tw2 @ w1)[:, :, None, None], t @ b2 + tw2 @ b1

    def forward(self, input_dict):
        sz_in = input_dict[1].shape[3]
        x = input_dict[8]
        out = {8: x}
        concat = self.skip_type != 'sum'
        x = self.decoder_scale_1(x)
        # nearest x2 + skip concat written into one NHWC buffer (no separate interpolate / cat tensors)
        x = hip.upsample2x_concat(x, input_dict[4]) if concat else hip.upsample2x_concat(x) + input_dict[4]
        x = self.decoder_scale_2(x)
        self.update_skip_dict(out, x, sz_in)
        x = hip.upsample2x_concat(x, input_dict[2]) if concat else hip.upsample2x_concat(x) + input_dict[2]
        x = self.decoder_scale_3(x)
        self.update_skip_dict(out, x, sz_in)
        x = hip.upsample2x_concat(x)
        x = self.decoder_scale_4(x)
        # 'pooled': the 256-channel map only feeds the superpixel mean, which commutes with the 1x1 convolution (hip.PointwiseFeature)
        if self.materialize_ch256 == 'pooled':
            x_ch256 = hip.PointwiseFeature(x, self.decoder_ch256[0])
        else:
            x_ch256 = self.decoder_ch256[0](x) if self.materialize_ch256 else None
        wf, bf = self._composed_head()
        ver = tuple(p._version for p in (self.decoder_ch256[0].weight, self.decoder_ch256[0].bias,
                                         self.decoder_ch512[0].weight, self.decoder_ch512[0].bias, self.text_embeddings))
        logits = engine.conv2d_train(x, wf, bf, self._pw_head, 1, out_f32=True, ver=ver)
        if self.if_linear_probing:
            logits = hip.linear_probe(logits, self.linear_probe)
        self.update_skip_dict(out, logits, sz_in)
        return out, x_ch256
