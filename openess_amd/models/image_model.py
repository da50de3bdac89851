"""Mirror of models/image_model.py:DilationFeatureExtractor (:90-143) and
models/modules/resnet_encoder.py:ResNetEncoder (:8-38): frozen dilated ResNet-50 (output stride 4),
trainable 1x1 decoder, x4 bilinear (align_corners=True), L2 normalisation over channels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, hip
from ._resnet import Bottleneck, HipConv2d, ResNet


class ResNetEncoder(ResNet):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        del self.fc
        del self.avgpool

    def forward(self, x):
        return self.features(x)

    def load_state_dict(self, state_dict, **kwargs):
        state_dict.pop("fc.bias", None)
        state_dict.pop("fc.weight", None)
        return super().load_state_dict(state_dict, **kwargs)


class DilationFeatureExtractor(nn.Module):
    def __init__(self, image_weights=None, preprocessing=None):
        super().__init__()
        if image_weights not in (None, '', 'none', 'random'):
            # the reference downloads SSL checkpoints over HTTP (image_model.py:38-42); no network here
            print(f"[openess_amd] image_weights='{image_weights}' not loaded (no network): load a state_dict instead")
        self.encoder = ResNetEncoder(block=Bottleneck, layers=[3, 4, 6, 3], replace_stride_with_dilation=[True, True, True])
        for p in self.encoder.parameters():
            p.requires_grad = False
        self.decoder = nn.Sequential(HipConv2d(2048, 256, 1), nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True))
        self.preprocessing = preprocessing
        self.normalize_feature = True
        # True: the differentiable path returns hip.UpsampledNormalizedFeature (x, 4) for a consumer that only pools the features
        # over superpixels (PretrainStep with the contrastive loss); its .materialize() is the tensor this module returns otherwise
        self.lazy_features = False

    def forward(self, x):
        return self.head(self.encode(x))

    def encode(self, x):
        """The frozen part (preprocessing + dilated ResNet-50, train-mode BatchNorm side effects included): depends on no trainable
        weight, so a training loop may run it for the NEXT batch while the current one is still in its backward pass."""
        if self.preprocessing:
            x = self.preprocessing(x)
        x = engine.to_cl_bf16(x)
        with torch.no_grad():                      # encoder params are frozen (image_model.py:113-114)
            return self.encoder(x)

    def head(self, x):
        """The trainable 1x1 decoder + x4 bilinear + L2 normalisation on the encoder's output."""
        x = self.decoder[0](x)
        if torch.is_grad_enabled() and x.requires_grad:
            # differentiable path (contrastive loss active)
            if self.normalize_feature and x.dtype == torch.bfloat16 and x.shape[1] % 64 == 0 and x.shape[1] <= 512:
                if self.lazy_features:
                    return hip.UpsampledNormalizedFeature(x, 4)
                return hip.bilinear_l2norm_train(x, 4)            # one fused forward kernel + (L2 adjoint, bilinear adjoint)
            x = hip.bilinear_resize(x, scale_factor=4, align_corners=True)
            return hip.l2_normalize(x) if self.normalize_feature else x
        return hip.bilinear_l2norm(x, 4, self.normalize_feature)
