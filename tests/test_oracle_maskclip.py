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
