"""GPU parity of the MaskCLIP ViT-B/16 tower vs the CPU fp32 restatement in oracle/maskclip.py.
The reference module needs mmcv / mmseg (absent here), so this row is pinned to the restatement only ("parity
unpinned" in DESIGN.md); state_dict keys are checked against the mmcv naming a MaskCLIP checkpoint uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(img_size, K=11, seed=0):
    from oracle.maskclip import maskClipFeatureExtractor as Oracle
    from openess_amd.models.maskclip_model import maskClipFeatureExtractor as Mirror
    torch.manual_seed(seed)
    o = Oracle(K, img_size=img_size)
    with torch.no_grad():
        for n, p in o.named_parameters():
            if n.endswith('cls_token') or n.endswith('pos_embed'):
                p.normal_(0, 0.3)
            elif 'ln' in n and n.endswith('weight'):
                p.uniform_(0.7, 1.3)
            elif n.endswith('bias'):
                p.normal_(0, 0.1)
        o.decoder.text_embeddings.copy_(torch.nn.functional.normalize(torch.randn(K, 512), dim=1))
    m = Mirror(text_categories=K, img_size=img_size)
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    m.load_state_dict(o.state_dict())
    return o.eval(), m.cuda().eval()


def test_layernorm_and_attention_kernels():
    from openess_amd import hip
    torch.manual_seed(1)
    B, L, heads = 2, 77, 12
    C = heads * 64
    x = torch.randn(B * L, C, device="cuda").bfloat16()
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    y = hip.layer_norm_tokens(x, g, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (C,), g, b, 1e-6)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)
    for B, L in ((2, 77), (1, 32), (3, 130), (1, 1121)):          # ragged key / query tiles, the full-size token count
        qkv = (torch.randn(B * L, 3 * C, device="cuda") * 1.5).bfloat16()
        o = hip.attention_d64(qkv, B, L, heads)
        q, k, v = (t.view(B, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(C, dim=1))
        att = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
        ref = att.permute(0, 2, 1, 3).reshape(B * L, C)
        # P is rounded to bf16 before the P V product (MFMA operand): 2^-8 relative on each weight
        np.testing.assert_allclose(o.float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("img_size,hw", [((32, 32), (48, 80)), ((32, 48), (40, 70))])
def test_maskclip_tower_matches_restatement(img_size, hw):
    """pos-embed resize (bicubic), corner padding (40x70 -> 48x80), 11 full blocks + the value path of the last block,
    head and final bilinear resize.  bf16 activations through 12 blocks: logits within 4e-2 of their range, argmax
    agreement >= 97 % (near-ties flip under bf16 rounding)."""
    o, m = _pair(img_size)
    torch.manual_seed(5)
    img = torch.rand(2, 3, *hw)
    with torch.no_grad():
        ref = o(img)
        _, v_ref = o.encoder(img)
    out = m(img.cuda())
    v = m.encoder(img.cuda())
    assert out.shape == ref.shape == (2, 11, hw[0], hw[1]) and out.dtype == torch.float32
    vr = v_ref.numpy()
    assert np.abs(v.float().cpu().numpy() - vr).max() < 4e-2 * np.abs(vr).max()
    r = ref.numpy()
    assert np.abs(out.cpu().numpy() - r).max() < 4e-2 * (r.max() - r.min())
    agree = (out.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
    assert agree >= 0.97, agree


def test_online_teacher_labels_in_the_step():
    """the frozen tower as ONLINE teacher inside PretrainStep -- the step with `online_teacher` equals the step fed
    with the tower's argmax map as offline pseudo-labels, and that map agrees with the oracle tower's.  The two steps run the same
    kernels on the same labels; what differs from run to run is the arrival order of the fp32 atomics in the normalisation
    statistics and the loss reduction (observed 0 - 2.5e-6 relative between two runs of the SAME step), hence rel 2e-5."""
    from openess_amd.training.pretrain_step import PretrainStep
    from tests.synth import damp_residual, fill_by_name
    o, m = _pair((32, 48))
    torch.manual_seed(2)
    B, H, W, nwin = 2, 64, 96, 2
    ev = (torch.randn(B, nwin * 5, H, W) * (torch.rand(B, nwin * 5, H, W) > 0.7)).contiguous().cuda()
    frame = torch.rand(B, 3, H, W)
    with torch.no_grad():
        labels = m(frame.cuda()).argmax(1)
        ref_labels = o(frame).argmax(1)
    assert (labels.cpu() == ref_labels).float().mean().item() >= 0.97
    losses = []
    for teacher, pl in ((m, torch.zeros_like(labels)), (None, labels)):
        st = PretrainStep(config_option="frame2voxel", img_size=(H, W), nr_events_data=nwin, if_spatial_contrastive=False, lr=1e-4,
                          online_teacher=teacher)
        for name, mod in st.models_dict.items():
            fill_by_name(mod, 100 + len(name))
            damp_residual(mod)
        ls, _, _ = st.train_step((ev, None, frame.cuda(), pl, None, None))
        losses.append(float(ls['dense_clip_loss']))
    assert losses[0] == pytest.approx(losses[1], rel=2e-5)

