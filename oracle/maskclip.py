"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU fp32 restatement of the MaskCLIP ViT-B/16 teacher
(SURVEY.md 8a row a19 / 8f rank 1), written from the cited reference lines in plain PyTorch:

  models/maskclip_model.py:52-222   MaskClipHead (vit=True branch: proj -> L2 normalise -> text-embedding classifier)
  models/maskclip_model.py:448-541  TransformerEncoderLayer incl. the `return_qkv` value path of the last layer
  models/maskclip_model.py:545-851  VisionTransformer (corner padding, cls token, bicubic pos-embed resize, pre/final LN)
  models/maskclip_model.py:854-915  maskClipFeatureExtractor.forward (logits resized to the image, bilinear, align False)

PARITY UNPINNED: the reference module imports mmcv / mmseg, which are absent from this image, so the
reference itself cannot be run here and no golden vectors from it exist.  Third-party pieces restated from their
documented behaviour: mmcv 1.x `MultiheadAttention` (a residual wrapper around `nn.MultiheadAttention`, batch_first
handled by transposes), mmcv `FFN` (Linear-GELU-Linear + identity), `build_norm_layer(LN, eps=1e-6)`, mmcv
`AdaptivePadding('corner')` (zero pad bottom / right to a multiple of the patch size), mmseg `resize` (= F.interpolate).
Parameter names are the mmcv ones (`layers.N.ln1`, `layers.N.attn.attn.in_proj_weight`, `layers.N.ffn.layers.0.0.weight`,
`ln0`, `ln1`, `patch_embed.projection.weight`) so that a MaskCLIP checkpoint's `backbone.*` keys map 1:1."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attn(nn.Module):                      # mmcv MultiheadAttention: .attn is the torch module (maskclip_model.py:496)
    def __init__(self, dims, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(dims, heads, bias=True)

    def forward(self, x, identity):
        y = x.transpose(0, 1)
        y = self.attn(y, y, y, need_weights=False)[0].transpose(0, 1)
        return identity + y


class _FFN(nn.Module):                       # mmcv FFN(num_fcs=2): layers = Sequential(Sequential(Linear, act, drop), Linear, drop)
    def __init__(self, dims, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dims, hidden), nn.GELU(), nn.Dropout(0.0)),
                                    nn.Linear(hidden, dims), nn.Dropout(0.0))

    def forward(self, x, identity):
        return identity + self.layers(x)


class TransformerEncoderLayer(nn.Module):    # maskclip_model.py:448-541
    def __init__(self, dims=768, heads=12, hidden=3072):
        super().__init__()
        self.ln1 = nn.LayerNorm(dims, eps=1e-6)
        self.attn = _Attn(dims, heads)
        self.ln2 = nn.LayerNorm(dims, eps=1e-6)
        self.ffn = _FFN(dims, hidden)

    def forward(self, x, return_qkv=False):
        v = None
        if return_qkv:                       # :518-536  value path: v = out_proj(v_in) ; v += x ; v = ffn(ln2(v))
            y = F.linear(self.ln1(x), self.attn.attn.in_proj_weight, self.attn.attn.in_proj_bias)
            N, L, C3 = y.shape
            y = y.view(N, L, 3, C3 // 3).permute(2, 0, 1, 3).reshape(3 * N, L, C3 // 3)
            y = F.linear(y, self.attn.attn.out_proj.weight, self.attn.attn.out_proj.bias)
            v = y[2 * N:]
            v = v + x
            v = self.ffn(self.ln2(v), identity=v)
        x = self.attn(self.ln1(x), identity=x)
        x = self.ffn(self.ln2(x), identity=x)
        return x, v


class VisionTransformer(nn.Module):          # maskclip_model.py:545-851 with the constructor defaults (:596-625)
    def __init__(self, img_size=(224, 224), patch_size=16, dims=768, layers=12, heads=12):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.patch_embed = nn.Module()
        self.patch_embed.projection = nn.Conv2d(3, dims, patch_size, patch_size, bias=False)
        n_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dims))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, dims))
        self.layers = nn.ModuleList([TransformerEncoderLayer(dims, heads, 4 * dims) for _ in range(layers)])
        self.ln0 = nn.LayerNorm(dims, eps=1e-6)
        self.ln1 = nn.LayerNorm(dims, eps=1e-6)

    def resized_pos_embed(self, hw):          # :770-797
        ph, pw = self.img_size[0] // self.patch_size, self.img_size[1] // self.patch_size
        if hw == (ph, pw):
            return self.pos_embed
        cls_w = self.pos_embed[:, 0:1]
        grid = self.pos_embed[:, -ph * pw:].reshape(1, ph, pw, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=hw, mode='bicubic', align_corners=False)
        return torch.cat((cls_w, grid.flatten(2).transpose(1, 2)), dim=1)

    def forward(self, img):                   # :799-851; returns (x_map, v_map)
        B, _, H, W = img.shape
        p = self.patch_size
        img = F.pad(img, (0, (-W) % p, 0, (-H) % p))             # AdaptivePadding('corner')
        x = self.patch_embed.projection(img)
        hw = (x.shape[2], x.shape[3])
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        x = x + self.resized_pos_embed(hw)
        x = self.ln0(x)
        v = None
        for i, layer in enumerate(self.layers):
            x, vv = layer(x, return_qkv=(i == len(self.layers) - 1))
            if vv is not None:
                v = vv
        x, v = self.ln1(x), self.ln1(v)
        to_map = lambda t: t[:, 1:].reshape(B, hw[0], hw[1], -1).permute(0, 3, 1, 2).contiguous()
        return to_map(x), to_map(v)


class MaskClipHead(nn.Module):                # maskclip_model.py:52-222, vit=True
    def __init__(self, text_categories, text_channels=512, in_channels=768):
        super().__init__()
        self.register_buffer('text_embeddings', torch.randn(text_categories, text_channels))
        self.proj = nn.Conv2d(in_channels, text_channels, 1, bias=False)
        self.image_mapping_local = nn.Conv2d(in_channels, 512, 1)       # constructed, never used (:125)

    def forward(self, v_map):
        feat = self.proj(v_map)
        feat = feat / feat.norm(dim=1, keepdim=True)                     # cls_seg :216-219
        return v_map, F.conv2d(feat, self.text_embeddings[:, :, None, None])


class maskClipFeatureExtractor(nn.Module):    # maskclip_model.py:854-915
    def __init__(self, text_categories, img_size=(224, 224)):
        super().__init__()
        self.encoder = VisionTransformer(img_size=img_size)
        self.decoder = MaskClipHead(text_categories)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, img):
        _, v_map = self.encoder(img)
        _, logits = self.decoder(v_map)
        return F.interpolate(logits, size=img.shape[2:], mode='bilinear', align_corners=False)
