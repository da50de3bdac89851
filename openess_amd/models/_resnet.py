"""Mirror of models/_resnet.py (ResNet :117-209, Bottleneck :74-114, resnet50) with the convolutions
on the HIP MFMA kernel.  state_dict keys equal torchvision's / the reference's."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, hip


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (bias-free in ResNet) executed by the MFMA kernel; differentiable when its weight
    requires grad, plain inference otherwise."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._pw = engine.PackedWeight()
        self._pw_folded = engine.PackedWeight()      # eval-mode BatchNorm folded in

    def forward(self, x):
        if x.shape[1] % 8 or x.dtype != torch.bfloat16:
            x = engine.to_cl_bf16(x)
        elif x.stride(1) != 1:                       # e.g. torch.cat of an expanded tensor: make it NHWC
            x = x.contiguous(memory_format=torch.channels_last)
        k, s, p, d = self.kernel_size[0], self.stride[0], self.padding[0], self.dilation[0]
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            return engine.conv2d_train(x, self.weight, self.bias, self._pw, k, s, p, d)
        pw = self._pw.get(self.weight, self.bias, None, cin_pad=x.shape[1])
        return engine.conv2d_infer(x, pw, self.out_channels, k, s, p, d)


class HipMaxPool2d(nn.MaxPool2d):
    """nn.MaxPool2d(3, stride=2, padding=1) of the stem on the HIP kernel (forward and backward; ATen's tie rule)."""

    def forward(self, x):
        if x.is_cuda and (self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode) == (3, 2, 1, 1, False):
            if x.dtype != torch.bfloat16 or x.stride(1) != 1 or x.shape[1] % 8:
                x = engine.to_cl_bf16(x)
            return hip.max_pool_3x3s2(x)
        return super().forward(x)


def conv_bn(conv, bn, x, relu=False, residual=None, out=None, skip_in=None, skip_out=None):
    """conv -> BatchNorm2d [-> + residual] [-> ReLU] with nn.BatchNorm2d semantics for both bn.training states.
    No autograd needed (frozen teacher, validation): train-mode BN takes its batch statistics from the conv
    epilogue (no statistics pass); eval-mode BN is folded into the packed weights and the whole tail is the
    conv epilogue.  With autograd: MFMA conv + library BatchNorm."""
    needs_grad = torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad or bn.weight.requires_grad)
    if needs_grad and conv.bias is None and bn.training and bn.weight is not None and bn.running_mean is not None and \
            conv.out_channels % 8 == 0 and conv.out_channels <= 2048:
        # one autograd node: statistics from the conv epilogue (no statistics pass), fused BatchNorm backward
        if x.shape[1] % 8 or x.dtype != torch.bfloat16:
            x = engine.to_cl_bf16(x)
        elif x.stride(1) != 1:
            x = x.contiguous(memory_format=torch.channels_last)
        y = engine.conv_bn_train(x, conv, bn, conv._pw, relu=relu, residual=residual, skip_in=skip_in, skip_out=skip_out)
        if out is not None:
            out.copy_(y)
            return out
        return y
    if needs_grad or conv.bias is not None:
        y = engine.batch_norm_act(conv(x), bn, relu=relu, residual=residual)
        if out is not None:
            out.copy_(y)
            return out
        return y
    if x.shape[1] % 8 or x.dtype != torch.bfloat16:
        x = engine.to_cl_bf16(x)
    elif x.stride(1) != 1:
        x = x.contiguous(memory_format=torch.channels_last)
    if residual is not None and (residual.stride(1) != 1 or residual.dtype != torch.bfloat16):
        residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
    if bn.training:
        pw = conv._pw.get(conv.weight, None, None, cin_pad=x.shape[1])
        y = hip.conv_bn_train_nhwc(engine.nhwc(x), pw.packed, conv.out_channels, k, k, s, p, d, bn, relu=relu,
                                   residual=None if residual is None else engine.nhwc(residual),
                                   out=None if out is None else engine.nhwc(out))
        return engine.from_nhwc(y)
    pw = conv._pw_folded.get(conv.weight, None, bn, cin_pad=x.shape[1])
    return engine.conv2d_infer(x, pw, conv.out_channels, k, s, p, d, relu=relu, residual=residual, out=out)


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    assert groups == 1
    return HipConv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    return HipConv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def _one_node_path(self, x):
        """True when conv1 / conv3 of this block both take conv_bn's one-autograd-node path and x needs a gradient: only then
        can the skip connection's gradient ride in conv1's data-gradient epilogue (engine._ConvBNTrainFn skip_in / skip_out)."""
        if not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype == torch.bfloat16 and x.stride(1) == 1 and
                x.shape[1] % 8 == 0):
            return False
        for conv, bn in ((self.conv1, self.bn1), (self.conv3, self.bn3)):
            if conv.bias is not None or not bn.training or bn.weight is None or bn.running_mean is None or conv.out_channels % 8 or \
                    conv.out_channels > 2048 or conv.stride[0] != 1:
                return False
        return True

    def forward(self, x):
        identity = x
        # no downsample: x feeds conv1 AND the residual add -> one shared hand-over slot instead of autograd's gradient sum
        skip = {} if (self.downsample is None and self._one_node_path(x)) else None
        # A gradient parked by conv3's node that conv1's node never collected (a backward over a subset of the graph) would be a
        # silently missing term: found at the block's next forward, it raises.
        last = getattr(self, '_skip_slot', None)
        if last is not None and 'g' in last:
            last.clear()
            raise RuntimeError("Bottleneck: the skip-connection gradient parked by the last backward pass was never consumed "
                               "(backward over a sub-graph that excluded the block's first conv)")
        self._skip_slot = skip
        out = conv_bn(self.conv1, self.bn1, x, relu=True, skip_in=skip)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        if self.downsample is not None:
            identity = conv_bn(self.downsample[0], self.downsample[1], x)
        return conv_bn(self.conv3, self.bn3, out, relu=True, residual=identity, skip_out=skip)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple")
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = HipConv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def stem(self, x):
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        return self.maxpool(x)

    def features(self, x):
        with engine.defer_bn_counters():             # 53 num_batches_tracked bumps -> one multi-tensor add
            x = self.stem(x)
            x = self.layer1(x)
            x = self.layer2(x)
            x = self.layer3(x)
            return self.layer4(x)

    def forward(self, x):
        x = self.features(x)
        x = torch.flatten(self.avgpool(x.float()), 1)
        return self.fc(x)


def resnet50(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise NotImplementedError("no network access: load weights with load_state_dict")
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)
