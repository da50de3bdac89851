"""CPU checks of the MaskCLIP restatement (oracle/maskclip.py): mmcv checkpoint key layout, the last layer's value
path against an independent expansion of nn.MultiheadAttention's packed projection, pos-embed resize and padding."""
import torch
import torch.nn.functional as F

from oracle.maskclip import TransformerEncoderLayer, VisionTransformer, maskClipFeatureExtractor


def test_state_dict_uses_mmcv_names():
    keys = set(maskClipFeatureExtractor(11, img_size=(32, 32)).state_dict().keys())
    for k in ("encoder.patch_embed.projection.weight", "encoder.cls_token", "encoder.pos_embed", "encoder.ln0.weight",
              "encoder.ln1.bias", "encoder.layers.11.ln2.weight", "encoder.layers.0.attn.attn.in_proj_weight",
              "encoder.layers.0.attn.attn.out_proj.bias", "encoder.layers.3.ffn.layers.0.0.weight",
              "encoder.layers.3.ffn.layers.1.bias", "decoder.proj.weight", "decoder.text_embeddings"):
        assert k in keys, k
    assert len(keys) == 155


def test_value_path_is_out_proj_of_value_projection():
    torch.manual_seed(0)
    lyr = TransformerEncoderLayer(128, 2, 256).eval()
    x = torch.randn(2, 7, 128)
    _, v = lyr(x, return_qkv=True)
    a = lyr.attn.attn
    wv, bv = a.in_proj_weight[256:], a.in_proj_bias[256:]
    t = F.linear(F.linear(lyr.ln1(x), wv, bv), a.out_proj.weight, a.out_proj.bias) + x
    ref = t + lyr.ffn.layers(lyr.ln2(t))
    assert torch.allclose(v, ref, atol=1e-5)


def test_padding_and_pos_embed_resize():
    torch.manual_seed(1)
    vit = VisionTransformer(img_size=(32, 32), layers=1).eval()
    with torch.no_grad():
        vit.pos_embed.normal_()
        x_map, v_map = vit(torch.rand(1, 3, 40, 70))          # 40x70 -> 48x80 corner padded -> 3x5 patches
    assert x_map.shape == v_map.shape == (1, 768, 3, 5)
    pe = vit.resized_pos_embed((3, 5))
    assert pe.shape == (1, 16, 768) and torch.equal(pe[:, 0], vit.pos_embed[:, 0])
    assert vit.resized_pos_embed((2, 2)) is vit.pos_embed


def test_encoder_block_pinned_to_pytorch_primitives():
    """Pin of the block ARITHMETIC: the reference block (maskclip_model.py:448-541) is mmcv's wrapper around
    torch.nn.MultiheadAttention + LayerNorm(eps 1e-6) + Linear-GELU-Linear, i.e. PyTorch primitives (mmcv 1.x, absent here).
    On shared weights the restated layer must equal (a) a FRESH batch_first nn.MultiheadAttention / F.layer_norm / F.gelu
    composition and (b) the explicit per-head softmax(q k^T / sqrt(d)) v expansion.  What stays restated (not pinned) is only
    the value-path wiring of the last layer, covered by test_value_path_is_out_proj_of_value_projection."""
    torch.manual_seed(3)
    dims, heads, hidden = 768, 12, 3072
    lyr = TransformerEncoderLayer(dims, heads, hidden).eval()
    with torch.no_grad():
        for n, p in lyr.named_parameters():
            if 'ln' in n and n.endswith('weight'):
                p.uniform_(0.7, 1.3)
            elif n.endswith('bias'):
                p.normal_(0, 0.1)
    x = torch.randn(2, 37, dims)
    with torch.no_grad():
        got, _ = lyr(x)
        sd = lyr.state_dict()
        # (a) fresh PyTorch modules, batch_first, same weights
        mha = torch.nn.MultiheadAttention(dims, heads, bias=True, batch_first=True).eval()
        mha.load_state_dict({k[len('attn.attn.'):]: v for k, v in sd.items() if k.startswith('attn.attn.')})
        y = F.layer_norm(x, (dims,), sd['ln1.weight'], sd['ln1.bias'], 1e-6)
        h = x + mha(y, y, y, need_weights=False)[0]
        z = F.layer_norm(h, (dims,), sd['ln2.weight'], sd['ln2.bias'], 1e-6)
        ref_a = h + F.linear(F.gelu(F.linear(z, sd['ffn.layers.0.0.weight'], sd['ffn.layers.0.0.bias'])),
                             sd['ffn.layers.1.weight'], sd['ffn.layers.1.bias'])
        # (b) explicit attention
        qkv = F.linear(y, sd['attn.attn.in_proj_weight'], sd['attn.attn.in_proj_bias'])
        q, k, v = (t.view(2, 37, heads, dims // heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        att = torch.softmax(q @ k.transpose(-1, -2) / (dims // heads) ** 0.5, dim=-1) @ v
        att = F.linear(att.transpose(1, 2).reshape(2, 37, dims), sd['attn.attn.out_proj.weight'], sd['attn.attn.out_proj.bias'])
        hb = x + att
        zb = F.layer_norm(hb, (dims,), sd['ln2.weight'], sd['ln2.bias'], 1e-6)
        ref_b = hb + F.linear(F.gelu(F.linear(zb, sd['ffn.layers.0.0.weight'], sd['ffn.layers.0.0.bias'])),
                              sd['ffn.layers.1.weight'], sd['ffn.layers.1.bias'])
    assert torch.allclose(got, ref_a, atol=1e-5, rtol=1e-5), float((got - ref_a).abs().max())
    assert torch.allclose(got, ref_b, atol=2e-5, rtol=1e-5), float((got - ref_b).abs().max())


def test_pos_embed_resize_pinned_to_f_interpolate():
    """maskclip_model.py:770-797: the patch part of pos_embed, laid out [1, C, ph, pw], resized by F.interpolate(bicubic,
    align_corners=False), cls weight kept -- recomputed here channel by channel on a [C, 1, ph, pw] batch (an independent layout)."""
    torch.manual_seed(4)
    vit = VisionTransformer(img_size=(224, 224), layers=1).eval()
    with torch.no_grad():
        vit.pos_embed.normal_()
        pe = vit.resized_pos_embed((28, 40))
        grid = vit.pos_embed[0, 1:].reshape(14, 14, 768).permute(2, 0, 1)[:, None]              # [C, 1, 14, 14]
        ref = F.interpolate(grid, size=(28, 40), mode='bicubic', align_corners=False)[:, 0]   # [C, 28, 40]
    assert pe.shape == (1, 1 + 28 * 40, 768)
    assert torch.equal(pe[0, 0], vit.pos_embed[0, 0])
    assert torch.allclose(pe[0, 1:], ref.reshape(768, -1).t(), atol=1e-6)
