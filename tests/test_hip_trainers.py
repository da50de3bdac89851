"""GPU parity of the trainer steps BASELINE configs[0] and configs[4] need and round 1 never compared with an oracle:
fine-tune and linear-probe steps (training/finetune_trainer.py:285-386, linear_probe_trainer.py:276-371) built through
train.py's own dispatch, and a DDD17-shaped pre-training step (200x352, K=6, 2-bin voxels, configs[0] geometry)."""
import os

import numpy as np
import pytest
import torch

from oracle.step import E2VID_LIGHTWEIGHT_CONFIG, OracleStep, OracleSupervisedStep
from tests.synth import damp_residual, fill_by_name

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def _trainer(option, linear_probing, tmp_path):
    import train
    from openess_amd.config.settings import Settings
    train.seed_everything()
    s = Settings(os.path.join(CFG, "finetune_dsec_synthetic.yaml"), generate_log=False)
    s.ckpt_dir = str(tmp_path)
    s.config_option = option
    s.if_finetuning, s.if_linear_probing = (not linear_probing), linear_probing
    trainer, loop = train.build_trainer(s)
    assert loop == 'training'
    assert type(trainer).__name__ == ('OpenESSLinearProbeModel' if linear_probing else 'OpenESSFineTuneModel')
    return trainer, s


@pytest.mark.parametrize("option,linear_probing", [("frame2recon", False), ("frame2voxel", False),
                                                   ("frame2voxel", True), ("frame2recon", True)])
def test_supervised_step_matches_oracle(option, linear_probing, tmp_path):
    trainer, s = _trainer(option, linear_probing, tmp_path)
    K, nwin, (H, W) = s.semseg_num_classes, s.nr_events_data_b, s.img_size_b
    lr = s.lr_voxel if option == "frame2voxel" else s.lr_recon
    ref = OracleSupervisedStep(option, K, nwin, 5, linear_probing, lr=lr)
    for name, m in trainer.models_dict.items():
        fill_by_name(m, 300 + len(name))
        fill_by_name(ref.modules()[name], 300 + len(name), sorted(m.state_dict().keys()))
        damp_residual(m), damp_residual(ref.modules()[name])
    # trainable set: linear probing trains exactly the K->K 1x1 conv (weight + bias)
    n_train = sum(p.numel() for g in trainer.optimizers_dict.values() for grp in g.param_groups for p in grp['params'])
    n_ref = sum(p.numel() for grp in ref.optim.param_groups for p in grp['params'])
    assert n_train == n_ref
    if linear_probing:                 # K->K 1x1 probe (+ SemSegE2VID's unused decoder_scale_5, which the reference forgets to freeze)
        assert n_train == K * K + K + (32 * K + K if option == "frame2voxel" else 0)
    if option == "frame2recon":
        trainer.model_recon.classifier.ASPP.project[3].p = 0.0
        ref.net.classifier.ASPP.project[3].p = 0.0
    torch.manual_seed(4)
    B = 2
    ev = (torch.randn(B, nwin * 5, H, W) * (torch.rand(B, nwin * 5, H, W) > 0.7)).contiguous()
    recon = torch.rand(B, 3, H, W)
    gt = torch.randint(0, K, (B, H // 4, W // 4)).repeat_interleave(4, 1).repeat_interleave(4, 2)
    gt[0, :5] = 255
    key = 'semseg_sensor_b_loss' if option == "frame2voxel" else 'semseg_recon_loss'
    for it in range(2):
        losses, _, total = trainer.train_step((ev.cuda(), gt.cuda(), recon.cuda(), gt.cuda(), gt.cuda(), None))
        lref, tref = ref.train_step((ev, gt, recon))
        assert set(losses) == {key}
        assert float(losses[key]) == pytest.approx(float(lref[key]), rel=2e-2), (it, float(losses[key]), float(lref[key]))
    if linear_probing:                 # the probe's gradient agrees; AdamW moved nothing else (|update| <= lr per step: sign(g) early on)
        net = trainer.models_dict['back_end' if option == "frame2voxel" else 'model_recon']
        a, b = net.linear_probe.weight.grad.cpu().numpy().ravel(), ref.net.linear_probe.weight.grad.numpy().ravel()
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.99
        d = np.abs(net.linear_probe.weight.detach().cpu().numpy() - ref.net.linear_probe.weight.detach().numpy())
        assert d.max() <= 4.5 * lr and np.median(d) <= 0.5 * lr
        assert all(not p.requires_grad for n, p in net.named_parameters()
                   if not n.startswith(('linear_probe', 'decoder_scale_5')))


def test_ddd17_shaped_pretrain_step():
    """BASELINE configs[0] geometry: 200x352 network input, K = 6 classes, 2-bin voxels (E2VID head with 2 input channels)."""
    from openess_amd.training.pretrain_step import PretrainStep
    torch.manual_seed(3)
    B, H, W, nwin, K, bins = 2, 200, 352, 2, 6, 2
    cfg = dict(E2VID_LIGHTWEIGHT_CONFIG, num_bins=bins)
    st = PretrainStep(config_option="frame2voxel", num_classes=K, img_size=(H, W), nr_events_data=nwin, nr_temporal_bins=bins,
                      if_spatial_contrastive=True, superpixel_size=25, lr=1e-4, e2vid_config=cfg)
    ref = OracleStep("frame2voxel", K, nwin, bins, True, 25, lr=1e-4, e2vid_config=cfg)
    for name, m in st.models_dict.items():
        fill_by_name(m, 100 + len(name))
        fill_by_name(ref.modules()[name], 100 + len(name), sorted(m.state_dict().keys()))
        damp_residual(m), damp_residual(ref.modules()[name])
    ev = (torch.randn(B, nwin * bins, H, W) * (torch.rand(B, nwin * bins, H, W) > 0.7)).contiguous()
    frame = torch.rand(B, 3, H, W)
    pl = torch.randint(0, K, (B, H // 8, W // 8)).repeat_interleave(8, 1).repeat_interleave(8, 2)
    pl[0, :5] = 255
    sp = torch.randint(0, 25, (B, H // 8, W // 8)).repeat_interleave(8, 1).repeat_interleave(8, 2)
    S = int((sp + torch.arange(B)[:, None, None] * 25).max()) + 1
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    for it in range(2):
        losses, _, tl = st.train_step((ev.cuda(), None, frame.cuda(), pl.cuda(), sp.cuda(), S))
        lref, tref = ref.train_step((ev, None, frame, pl, sp))
        for k in lref:
            tol = 5e-2 if k == 'contrastive_nce_loss' else 2e-2
            assert float(losses[k]) == pytest.approx(float(lref[k]), rel=tol), (it, k, float(losses[k]), float(lref[k]))
