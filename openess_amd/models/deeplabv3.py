"""Mirror of models/deeplabv3.py: deeplabv3_resnet50 (:128-189), DeepLabHead (:86-125), ASPP (:319-348),
ASPPConv (:295-302), ASPPPooling (:305-316), IntermediateLayerGetter (:21-83).  Same state_dict keys."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, hip
from . import _resnet as resnet
from ._resnet import HipConv2d, conv_bn


class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError("return_layers are not present in model")
        orig = return_layers
        return_layers = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            if name in return_layers:
                del return_layers[name]
            if not return_layers:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        x = engine.to_cl_bf16(x)
        for name, module in self.named_children():
            if name == 'conv1':
                x = conv_bn(module, self['bn1'], x, relu=True)
                continue
            if name in ('bn1', 'relu'):
                continue
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class _ConvBNReLU(nn.Sequential):
    def forward(self, x, out=None):
        return conv_bn(self[0], self[1], x, relu=True, out=out)


class ASPPConv(_ConvBNReLU):
    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(HipConv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class ASPPPooling(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x, out=None):
        size = x.shape[-2:]
        bn = self[2]
        if bn.training and x.is_cuda and x.dtype == torch.bfloat16 and x.stride(1) == 1 and x.shape[1] % 8 == 0 and \
                x.shape[1] <= 2048 and 2 <= x.shape[0] <= 16 and bn.running_mean is not None:
            # train mode: pooling, B-row GEMV, BatchNorm over the B pooled vectors and ReLU on the HIP kernels, one autograd node
            y = hip.aspp_pool_branch(x, self[1], bn)
            if out is not None:
                out.copy_(y)
                return out
            return y
        if x.is_cuda and x.dtype == torch.bfloat16 and x.stride(1) == 1 and x.shape[1] % 8 == 0 and x.shape[1] <= 2048:
            y = hip.global_avg_pool(x)                                     # AdaptiveAvgPool2d(1) without an fp32 copy of the map
        else:
            y = x.float().mean(dim=(2, 3), keepdim=True)
        y = torch.matmul(y.flatten(1), self[1].weight.flatten(1).t())[:, :, None, None]   # B x C x 1 x 1: a GEMV, not a conv
        y = F.relu(self[2](y))
        y = y.to(x.dtype).expand(-1, -1, size[0], size[1])                 # bilinear from 1x1 == broadcast
        if out is not None:
            out.copy_(y)
            return out
        return y


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates):
        super().__init__()
        out_channels = 256
        modules = [_ConvBNReLU(HipConv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels),
                               nn.ReLU(inplace=True))]
        rate1, rate2, rate3 = tuple(atrous_rates)
        modules += [ASPPConv(in_channels, out_channels, rate1), ASPPConv(in_channels, out_channels, rate2),
                    ASPPConv(in_channels, out_channels, rate3), ASPPPooling(in_channels, out_channels)]
        self.convs = nn.ModuleList(modules)
        self.project = nn.Sequential(HipConv2d(5 * out_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels),
                                     nn.ReLU(inplace=True), nn.Dropout(0.1))

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            res = torch.cat([conv(x) for conv in self.convs], dim=1)
        else:
            # no autograd: every branch writes its channel slice of ONE NHWC concat buffer (conv output, BatchNorm apply in
            # place on the slice) -- no torch.cat pass over the 5 x 256 channels
            B, _, H, W = x.shape
            oc = self.convs[0][0].out_channels
            res = engine.empty_cl(B, oc * len(self.convs), H, W, x.device)
            for i, conv in enumerate(self.convs):
                conv(x, out=res[:, i * oc:(i + 1) * oc])
        y = conv_bn(self.project[0], self.project[1], res, relu=True)
        drop = self.project[3]
        if drop.training and drop.p > 0 and y.is_cuda and y.dtype == torch.bfloat16 and y.stride(1) == 1 and y.shape[1] % 8 == 0:
            return hip.dropout(y, drop.p, True, owner=drop)            # Philox mask recomputed in the backward pass (nn.Dropout(0.1), :343)
        return drop(y)


class DeepLabHead(nn.Module):
    def __init__(self, text_embeddings_path, text_categories, in_channels, num_classes, aspp_dilate=[12, 24, 36]):
        super().__init__()
        self.ASPP = ASPP(in_channels, aspp_dilate)
        self.pixel_feature = nn.Conv2d(256, 512, 3, padding=1, bias=False)     # unused in the reference too (:94)
        self.classifier = nn.Sequential(HipConv2d(256, 512, 3, padding=1, bias=False), nn.BatchNorm2d(512),
                                        nn.ReLU(inplace=True))
        self._init_weight()
        self.text_embeddings_path = text_embeddings_path
        if text_embeddings_path is None:
            self.text_embeddings = nn.Parameter(torch.zeros(text_categories, 512))
            nn.init.normal_(self.text_embeddings, mean=0.0, std=0.01)
        else:
            self.register_buffer('text_embeddings', torch.randn(text_categories, 512))
            if text_embeddings_path:
                loaded = torch.load(text_embeddings_path, map_location='cpu')   # reference: map_location='cuda'
                self.text_embeddings[:, :] = loaded[:, :]
        self._pw_text = engine.PackedWeight()

    def forward(self, feature):
        feature = self.ASPP(feature['out'])
        x = conv_bn(self.classifier[0], self.classifier[1], feature, relu=True)
        logits = engine.conv2d_train(x, self.text_embeddings[:, :, None, None].float(), None, self._pw_text, 1,
                                     ver=self.text_embeddings._version)
        return logits, feature

    def _init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class deeplabv3_resnet50(nn.Module):
    feats_fp32 = False
    lazy_feats = False            # see forward()

    def __init__(self, num_classes, text_embeddings_path, output_stride, pretrained_backbone, if_linear_probing=False,
                 if_finetuning=False, frozen_backbone=False):
        super().__init__()
        if output_stride == 8:
            replace_stride_with_dilation, aspp_dilate = [False, True, True], [12, 24, 36]
        else:
            replace_stride_with_dilation, aspp_dilate = [False, False, True], [6, 12, 18]
        backbone = resnet.resnet50(replace_stride_with_dilation=replace_stride_with_dilation)
        classifier = DeepLabHead(text_embeddings_path, num_classes, 2048, num_classes, aspp_dilate)
        self.backbone = IntermediateLayerGetter(backbone, return_layers={'layer4': 'out'})
        self.classifier = classifier
        if pretrained_backbone != '':
            pretrained = torch.load(pretrained_backbone, map_location='cpu')
            self.load_state_dict(pretrained['model_recon'], strict=True)
        self.if_linear_probing = if_linear_probing
        if if_linear_probing:
            for p in self.backbone.parameters():
                p.requires_grad = False
            for p in self.classifier.parameters():
                p.requires_grad = False
            self.linear_probe = nn.Conv2d(num_classes, num_classes, 1)
        self.if_finetuning = if_finetuning
        if if_finetuning and frozen_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False
            for p in self.classifier.parameters():
                p.requires_grad = True

    def forward(self, x):
        input_shape = x.shape[-2:]
        with engine.defer_bn_counters():              # one multi-tensor add for all BatchNorm step counters
            features = self.backbone(x)
            logist, feats = self.classifier(features)
        logist = hip.bilinear_resize(logist.float(), size=input_shape, align_corners=False)      # deeplabv3.py:183
        # deeplabv3.py:184.  feats_fp32: the full-resolution 256-channel map leaves in fp32 (interpolated from the bf16 OS16 map
        # without a second rounding): what the superpixel pooling / InfoNCE / L1 consistency losses consume
        if self.lazy_feats and torch.is_grad_enabled():
            # the consumer only pools the features over superpixels (PretrainStep, frame2recon + contrastive): hip.UpsampledFeature
            # multiplies the OS16 map with the pooling matrix instead of forming the full-resolution tensor
            feats = hip.UpsampledFeature(feats.float() if self.feats_fp32 else feats, input_shape, align_corners=False)
        else:
            feats = hip.bilinear_resize(feats.float() if self.feats_fp32 else feats, size=input_shape, align_corners=False)
        if self.if_linear_probing:
            logist = hip.linear_probe(logist, self.linear_probe)
        return logist, feats
