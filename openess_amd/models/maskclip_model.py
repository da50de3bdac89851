"""Mirror of the MaskCLIP ViT-B/16 image tower of models/maskclip_model.py: same module tree and
state_dict keys as the reference (`encoder.layers.N.ln1`, `encoder.layers.N.attn.attn.in_proj_weight`,
`encoder.layers.N.ffn.layers.0.0.weight`, `encoder.ln0/ln1`, `encoder.patch_embed.projection.weight`, `decoder.proj`,
`decoder.text_embeddings`), so `load_checkpoint1`-style loading of a MaskCLIP checkpoint's `backbone.*` keys works
unchanged; the forward runs on liboess kernels:

  patch embedding 16x16/16  -> MFMA conv (oess_conv2d_fwd_bf16)            maskclip_model.py:317-446, 812-813
  LayerNorm (eps 1e-6)      -> oess_layernorm_bf16                         :486-501, 706-718
  in_proj / out_proj / FFN  -> MFMA 1x1 convs over the token axis, bias / residual / GELU fused   :496-512
  attention                 -> oess_attention_d64_bf16                     :538
  last layer                -> value path only (:518-536): v = out_proj(W_v ln1(x)) + x ; v += ffn(ln2(v)) ; the
                               attention branch of that layer is dead for MaskClipHead (feat = proj(v), :181-183)
  head                      -> proj (1x1 conv) -> oess_l2norm -> text-embedding classifier (1x1 conv, fp32 logits)
                               -> bilinear resize to the image, align_corners=False            :157-222, 903-908

Frozen (the reference sets requires_grad=False on encoder and decoder, :888-891): inference only.
The reference trainers construct this model but never call it; here it is callable as an online
teacher: `maskClipFeatureExtractor(img) -> logits [B, K, H, W]`."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, hip


def _linear_tokens(x2d, weight, bias, pw, act=0, residual=None, rows=None, out_f32=False):
    """y = act(x @ W^T + b [+ residual]) for a bf16 token matrix [rows, Cin] on the MFMA conv kernel (1x1 conv, NHWC
    view [1, 1, rows, Cin]).  `weight` may be a row slice of a parameter (its version is taken from the base tensor)."""
    base = weight._base if weight._base is not None else weight
    bver = None if bias is None else (bias._base if bias._base is not None else bias)._version
    pw.get(weight.detach()[:, :, None, None], None if bias is None else bias.detach(), None, cin_pad=x2d.shape[1],
           ver=(base._version, bver, weight.data_ptr()))
    x4 = x2d.view(1, 1, x2d.shape[0], x2d.shape[1])
    r4 = None if residual is None else residual.view(1, 1, residual.shape[0], residual.shape[1])
    y = hip.conv2d_nhwc(x4, pw.packed, pw.bias, weight.shape[0], 1, 1, 1, 0, 1, relu=act, residual=r4, out_f32=out_f32)
    return y.view(x2d.shape[0], weight.shape[0])


class _Attn(nn.Module):
    def __init__(self, dims, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(dims, heads, bias=True)        # parameter container (mmcv wrapper layout)


class _FFN(nn.Module):
    def __init__(self, dims, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dims, hidden), nn.GELU(), nn.Dropout(0.0)),
                                    nn.Linear(hidden, dims), nn.Dropout(0.0))


class TransformerEncoderLayer(nn.Module):
    """maskclip_model.py:448-541."""

    def __init__(self, embed_dims=768, num_heads=12, feedforward_channels=3072):
        super().__init__()
        self.num_heads = num_heads
        self.ln1 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.attn = _Attn(embed_dims, num_heads)
        self.ln2 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.ffn = _FFN(embed_dims, feedforward_channels)
        self._pw = {k: engine.PackedWeight() for k in ('qkv', 'v', 'out', 'fc1', 'fc2')}

    def _ffn(self, x):
        y = hip.layer_norm_tokens(x, self.ln2.weight, self.ln2.bias, self.ln2.eps)
        fc1, fc2 = self.ffn.layers[0][0], self.ffn.layers[1]
        h = _linear_tokens(y, fc1.weight, fc1.bias, self._pw['fc1'], act=2)                  # Linear + GELU
        return _linear_tokens(h, fc2.weight, fc2.bias, self._pw['fc2'], residual=x)          # identity + Linear

    def forward(self, x, B, L):
        """x: bf16 tokens [B*L, C] -> same shape (full block)."""
        a = self.attn.attn
        y = hip.layer_norm_tokens(x, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        qkv = _linear_tokens(y, a.in_proj_weight, a.in_proj_bias, self._pw['qkv'])
        o = hip.attention_d64(qkv, B, L, self.num_heads)
        x = _linear_tokens(o, a.out_proj.weight, a.out_proj.bias, self._pw['out'], residual=x)
        return self._ffn(x)

    def forward_value_path(self, x):
        """:518-536 with return_qkv: v = out_proj(in_proj_v(ln1(x))) + x ; v = v + ffn(ln2(v))."""
        a = self.attn.attn
        C = x.shape[1]
        y = hip.layer_norm_tokens(x, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        v = _linear_tokens(y, a.in_proj_weight[2 * C:], a.in_proj_bias[2 * C:], self._pw['v'])
        v = _linear_tokens(v, a.out_proj.weight, a.out_proj.bias, self._pw['out'], residual=x)
        return self._ffn(v)


class VisionTransformer(nn.Module):
    """maskclip_model.py:545-851 with its constructor defaults (ViT-B/16, pre_norm, final_norm, return_qkv on the last
    layer).  forward(img) -> v_map [B, 768, H/16, W/16] (the value-path feature MaskClipHead consumes)."""

    def __init__(self, img_size=(224, 224), patch_size=16, in_channels=3, embed_dims=768, num_layers=12, num_heads=12,
                 mlp_ratio=4):
        super().__init__()
        if num_heads * 64 != embed_dims:
            raise ValueError("the attention kernel is specialised for head dimension 64")
        self.img_size, self.patch_size, self.embed_dims = tuple(img_size), patch_size, embed_dims
        self.patch_embed = nn.Module()
        self.patch_embed.projection = nn.Conv2d(in_channels, embed_dims, patch_size, patch_size, bias=False)
        n_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, embed_dims))
        self.layers = nn.ModuleList([TransformerEncoderLayer(embed_dims, num_heads, mlp_ratio * embed_dims)
                                     for _ in range(num_layers)])
        self.ln0 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.ln1 = nn.LayerNorm(embed_dims, eps=1e-6)
        self._pw_patch = engine.PackedWeight()
        self._pos_cache = {}

    def _pos(self, hw):
        """:770-797 resize_pos_embed: bicubic, align_corners=False, class-token weight kept (cached per grid size)."""
        key = (hw, self.pos_embed._version)
        if key not in self._pos_cache:
            ph, pw = self.img_size[0] // self.patch_size, self.img_size[1] // self.patch_size
            pe = self.pos_embed.detach().float()
            if hw != (ph, pw):
                grid = pe[:, -ph * pw:].reshape(1, ph, pw, -1).permute(0, 3, 1, 2)
                grid = F.interpolate(grid, size=hw, mode='bicubic', align_corners=False)
                pe = torch.cat((pe[:, 0:1], grid.flatten(2).transpose(1, 2)), dim=1)
            self._pos_cache = {key: pe}
        return self._pos_cache[key]

    @torch.no_grad()
    def forward(self, img):
        B, Cin, H, W = img.shape
        p, C = self.patch_size, self.embed_dims
        Hp, Wp = (H + p - 1) // p * p, (W + p - 1) // p * p
        x8 = torch.zeros((B, Hp, Wp, 8), dtype=torch.bfloat16, device=img.device)      # corner padding + 3 -> 8 channels
        x8[:, :H, :W, :Cin] = img.permute(0, 2, 3, 1)
        pw = self._pw_patch.get(self.patch_embed.projection.weight, None, None, cin_pad=8)
        patches = hip.conv2d_nhwc(x8, pw.packed, None, C, p, p, p, 0, 1)                # [B, hp, wp, C]
        hp, wp = patches.shape[1], patches.shape[2]
        L = hp * wp + 1
        tok = torch.empty((B, L, C), dtype=torch.float32, device=img.device)
        tok[:, 0] = self.cls_token.detach()[0, 0]
        tok[:, 1:] = patches.view(B, hp * wp, C)
        tok += self._pos((hp, wp))
        x = hip.layer_norm_tokens(tok.to(torch.bfloat16).view(B * L, C), self.ln0.weight, self.ln0.bias, self.ln0.eps)
        for layer in self.layers[:-1]:
            x = layer(x, B, L)
        v = self.layers[-1].forward_value_path(x)
        v = hip.layer_norm_tokens(v, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        return v.view(B, L, C)[:, 1:].reshape(B, hp, wp, C).permute(0, 3, 1, 2)         # logical NCHW, NHWC memory


class MaskClipHead(nn.Module):
    """maskclip_model.py:52-222 (vit=True): logits = conv2d(normalize(proj(v)), text_embeddings)."""

    def __init__(self, text_categories=16, text_channels=512, in_channels=768):
        super().__init__()
        self.align_corners = False
        self.num_classes = text_categories
        self.register_buffer('text_embeddings', torch.randn(text_categories, text_channels))
        self.proj = nn.Conv2d(in_channels, text_channels, 1, bias=False)
        self.image_mapping_local = nn.Conv2d(in_channels, 512, 1)           # constructed by the reference, never used (:125)
        self._pw_proj, self._pw_text = engine.PackedWeight(), engine.PackedWeight()

    @torch.no_grad()
    def forward(self, v_map):
        B, C, hp, wp = v_map.shape
        v2 = engine.nhwc(v_map).reshape(B * hp * wp, C)
        feat = _linear_tokens(v2, self.proj.weight[:, :, 0, 0], None, self._pw_proj)
        feat = hip.l2_normalize(feat.view(B, hp, wp, -1).permute(0, 3, 1, 2), eps=1e-30)    # feat / feat.norm(dim=1) (:217)
        f2 = engine.nhwc(feat).reshape(B * hp * wp, -1)
        logits = _linear_tokens(f2, self.text_embeddings, None, self._pw_text, out_f32=True)
        return v_map, logits.view(B, hp, wp, -1).permute(0, 3, 1, 2)


class maskClipFeatureExtractor(nn.Module):
    """maskclip_model.py:854-915.  Checkpoint / text-embedding / projection files are optional here (the reference
    requires them): pass paths to load them with the reference's key mapping, or load a state_dict afterwards."""

    def __init__(self, text_embeddings_path=None, visual_projs_path=None, text_categories=16, maskclip_checkpoint=None,
                 preprocessing=None, test_cfg=dict(mode='whole'), img_size=(224, 224)):
        super().__init__()
        self.encoder = VisionTransformer(img_size=img_size)
        self.decoder = MaskClipHead(text_categories=text_categories)
        self.align_corners = self.decoder.align_corners
        self.num_classes = self.decoder.num_classes
        self.test_cfg = test_cfg
        self.checkpoint = maskclip_checkpoint
        if text_embeddings_path:
            self.decoder.text_embeddings[:, :] = torch.load(text_embeddings_path, map_location='cpu')[:, :]
        if visual_projs_path:                                               # load_visual_projs (:132-143)
            sd = torch.load(visual_projs_path, map_location='cpu')['proj']
            self.decoder.proj.load_state_dict({k: (v[:, :, None, None] if 'weight' in k and v.ndim == 2 else v) for k, v in sd.items()})
        if maskclip_checkpoint:                                             # load_checkpoint1 (:20-50)
            pre = torch.load(maskclip_checkpoint, map_location='cpu')['state_dict']
            pre = {(k[len('backbone.'):] if k.startswith('backbone.') else k): v for k, v in pre.items()}
            mine = self.encoder.state_dict()
            mine.update({k: v for k, v in pre.items() if k in mine and mine[k].shape == v.shape})
            self.encoder.load_state_dict(mine)
        for p in self.parameters():
            p.requires_grad = False

    # (Two half batches on two HIP streams -- the tower couples no samples, and its token GEMMs leave 15-20 % of their last round of
    #  tiles empty -- were measured in round 5: 4.17 vs 4.22 ms, bit-identical; not kept.)
    @torch.no_grad()
    def forward(self, img):
        v_map = self.encoder(img)
        _, logits = self.decoder(v_map)
        return hip.bilinear_resize(logits.float(), size=(img.shape[2], img.shape[3]), align_corners=self.align_corners)
