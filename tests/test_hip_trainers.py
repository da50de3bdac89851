"""GPU parity of the trainer steps BASELINE configs[0] and configs[4] need and round 1 never compared with an oracle:
fine-tune and linear-probe steps (training/finetune_trainer.py:285-386, linear_probe_trainer.py:276-371) built through
train.py's own dispatch, and a DDD17-shaped pre-training step (200x352, K=6, 2-bin voxels, configs[0] geometry)."""
import os

import numpy as np
import pytest
import torch

from oracle.step import E2VID_LIGHTWEIGHT_CONFIG, OracleOpenESSStep, OracleStep, OracleSupervisedStep
from tests.synth import damp_residual, fill_by_name

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def _trainer(option, linear_probing, tmp_path, sup_only=False):
    import train
    from openess_amd.config.settings import Settings
    train.seed_everything()
    s = Settings(os.path.join(CFG, "finetune_dsec_synthetic.yaml"), generate_log=False)
    s.ckpt_dir = str(tmp_path)
    s.config_option = option
    s.if_finetuning, s.if_linear_probing = (not linear_probing and not sup_only), linear_probing
    s.if_supervised_only = sup_only
    trainer, loop = train.build_trainer(s)
    assert loop == 'training'
    want = 'SupOnlyModel' if sup_only else ('OpenESSLinearProbeModel' if linear_probing else 'OpenESSFineTuneModel')
    assert type(trainer).__name__ == want and type(trainer).__module__.endswith(
        {'SupOnlyModel': 'sup_only_trainer', 'OpenESSLinearProbeModel': 'linear_probe_trainer', 'OpenESSFineTuneModel': 'finetune_trainer'}[want])
    return trainer, s


@pytest.mark.parametrize("option,linear_probing,sup_only", [("frame2recon", False, False), ("frame2voxel", False, False),
                                                            ("frame2voxel", True, False), ("frame2recon", True, False),
                                                            ("frame2recon", False, True), ("frame2voxel", False, True)])
def test_supervised_step_matches_oracle(option, linear_probing, sup_only, tmp_path):
    """Fine-tune, linear-probe and supervised-only trainers (the three stage-2/3 modules of train.py:4-8) built through
    train.py's own dispatch: the step of each equals OracleSupervisedStep from identical weights."""
    trainer, s = _trainer(option, linear_probing, tmp_path, sup_only)
    assert trainer.scaler is None
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


@pytest.mark.parametrize("contr", [True, False])
def test_openess_model_step_matches_oracle(contr, tmp_path):
    """OpenESSModel.train_step (train.py's default branch, BASELINE configs[2] "openess_trainer full path",
    training/openess_trainer.py:326-355,478-529) against OracleOpenESSStep from identical weights over two optimiser steps: the
    five loss keys, both optimisers' parameter sets, and the pooling offset hard-coded to 30 (ids up to 44 > 30 collide across
    samples exactly as in the reference; the YAML's superpixel_size = 25 must be ignored)."""
    import train
    from openess_amd.config.settings import Settings
    train.seed_everything()
    s = Settings(os.path.join(CFG, "openess_dsec_synthetic.yaml"), generate_log=False)
    s.ckpt_dir = str(tmp_path)
    s.if_spatial_contrastive = contr
    s.lr_recon = s.lr_frame = 1e-4          # as the pre-training step tests: AdamW's first update is +-lr per weight whatever the gradient
    assert s.superpixel_size == 25
    trainer, loop = train.build_trainer(s)
    assert type(trainer).__name__ == 'OpenESSModel' and loop == 'training'
    K, (H, W) = s.semseg_num_classes, s.img_size_b
    ref = OracleOpenESSStep(K, contr, lr_recon=s.lr_recon, lr_frame=s.lr_frame, weight_task_loss=s.weight_task_loss)
    assert sorted(trainer.optimizers_dict) == ['optimizer_frame', 'optimizer_recon']
    for name in ('model_recon', 'model_frame'):
        m = trainer.models_dict[name]
        fill_by_name(m, 500 + len(name) + (7 if name == 'model_frame' else 0))
        fill_by_name(ref.modules()[name], 500 + len(name) + (7 if name == 'model_frame' else 0), sorted(m.state_dict().keys()))
        damp_residual(m), damp_residual(ref.modules()[name])
        m.classifier.ASPP.project[3].p = 0.0                      # dropout off on both sides (different RNG streams)
        ref.modules()[name].classifier.ASPP.project[3].p = 0.0
    for opt_name, ref_opt in (('optimizer_recon', ref.opt_recon), ('optimizer_frame', ref.opt_frame)):
        mine = [p for grp in trainer.optimizers_dict[opt_name].param_groups for p in grp['params']]
        theirs = [p for grp in ref_opt.param_groups for p in grp['params']]
        assert [tuple(p.shape) for p in mine] == [tuple(p.shape) for p in theirs], opt_name
    torch.manual_seed(8)
    B = 2
    frame, recon = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W)
    pl = torch.randint(0, K, (B, H // 4, W // 4)).repeat_interleave(4, 1).repeat_interleave(4, 2)
    pl[1, -6:] = 255
    sp = torch.randint(0, 45, (B, H // 8, W // 8)).repeat_interleave(8, 1).repeat_interleave(8, 2)
    assert int(sp[0].max()) >= 30                                  # sample 0's ids reach into sample 1's range [30, 60)
    keys = {'semseg_frame_loss', 'semseg_recon_loss', 'cons_feat_loss', 'cons_pred_loss'} | ({'contrastive_nce_loss'} if contr else set())
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    # Tolerances (measured, tools/exp_openess_tol.py -> profiles/r04_exp_openess_tolerances.txt, three seeds x three learning rates):
    #   step 0 (same weights on both sides): TaskLoss / L1 within 0.3 %, cosine logit consistency within 1.3 %, InfoNCE within 5.1 %
    #   (the loss is ~100-150: un-normalised 256-channel ASPP features at T = 0.07, logits in the thousands -- the fp32 oracle with
    #   bf16 rounding points moves by up to 4.7 % against itself there);
    #   step 1 (after one AdamW update of +-lr per weight): InfoNCE is ill-conditioned -- the oracle's OWN bf16-storage emulation
    #   lands 0.1 .. 23 % away from it depending on the seed, this pipeline 5 .. 25 %; fp32 full-resolution features change nothing
    #   (same file) -- so the second step bounds it at 30 % and pins the well-conditioned quantities instead: TaskLoss 2 %,
    #   L1 3 %, cosine consistency 10 %, and the step-0 gradients by direction.
    tol = [{'contrastive_nce_loss': 6e-2, 'cons_feat_loss': 1e-2, 'cons_pred_loss': 3e-2},
           {'contrastive_nce_loss': 3e-1, 'cons_feat_loss': 3e-2, 'cons_pred_loss': 1e-1}]
    grads = {}
    for it in range(2):
        losses, _, total = trainer.train_step((frame.cuda(), None, recon.cuda(), pl.cuda(), sp.cuda(), None))
        lref, tref = ref.train_step((frame, None, recon, pl, sp))
        assert set(losses) == keys == set(lref)
        for k in sorted(keys):
            assert float(losses[k]) == pytest.approx(float(lref[k]), rel=tol[it].get(k, 2e-2)), (it, k, float(losses[k]), float(lref[k]))
        if it == 0:
            assert float(total) == pytest.approx(float(tref), rel=5e-2 if contr else 2e-2)
            for name in ('model_recon', 'model_frame'):
                mine, theirs = dict(trainer.models_dict[name].named_parameters()), dict(ref.modules()[name].named_parameters())
                for pn in ('classifier.ASPP.project.0.weight', 'classifier.classifier.0.weight', 'classifier.ASPP.convs.0.0.weight'):
                    a, b = mine[pn].grad.float().cpu().numpy().ravel(), theirs[pn].grad.numpy().ravel()
                    grads[(name, pn)] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    # The head gradients of both students point the oracle's way as well as bf16 storage allows: the yardstick is the fp32
    # oracle with bf16 rounding points (oracle.nets.emulate_bf16_storage) run from the same weights -- on this random-weight,
    # train-mode-BatchNorm net its own step-0 head gradients have cosine 0.87 .. 0.96 against the plain oracle.
    from oracle import nets as on
    emu = OracleOpenESSStep(K, contr, lr_recon=s.lr_recon, lr_frame=s.lr_frame, weight_task_loss=s.weight_task_loss)
    for name in ('model_recon', 'model_frame'):
        fill_by_name(emu.modules()[name], 500 + len(name) + (7 if name == 'model_frame' else 0), sorted(trainer.models_dict[name].state_dict().keys()))
        damp_residual(emu.modules()[name])
        emu.modules()[name].classifier.ASPP.project[3].p = 0.0
    ref0 = OracleOpenESSStep(K, contr, lr_recon=s.lr_recon, lr_frame=s.lr_frame, weight_task_loss=s.weight_task_loss)
    for name in ('model_recon', 'model_frame'):
        ref0.modules()[name].load_state_dict(emu.modules()[name].state_dict())
        ref0.modules()[name].classifier.ASPP.project[3].p = 0.0
        on.emulate_bf16_storage(emu.modules()[name])
    cos_e = {}
    for st_ in (ref0, emu):
        st_.opt_recon.zero_grad(); st_.opt_frame.zero_grad()
        st_.loss((frame, None, recon, pl, sp))[0].backward()
    for (name, pn) in grads:
        a = dict(emu.modules()[name].named_parameters())[pn].grad.numpy().ravel()
        b = dict(ref0.modules()[name].named_parameters())[pn].grad.numpy().ravel()
        cos_e[(name, pn)] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    print("step-0 head-gradient cosines vs the fp32 oracle: HIP", grads, "bf16-storage emulation", cos_e)
    for key, c in grads.items():
        # measured (gpurun_out r4j): HIP 0.815 .. 0.929, emulation 0.868 .. 0.961, HIP within 0.02 .. 0.10 of the emulation
        assert c > 0.75 and c > cos_e[key] - 0.15, (key, c, cos_e[key])


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


def test_openess_model_two_streams_equal_one_stream(tmp_path):
    """OpenESSModel runs its two students on two HIP streams (forward, and through autograd the backward): the same kernels on the
    same buffers ordered by events, so two optimiser steps must end in the same bits as the one-stream order."""
    import train
    from openess_amd.config.settings import Settings
    from openess_amd.training.openess_trainer import OpenESSModel
    runs = []
    for two in (True, False, True):          # (the default is one stream; the switch must stay exact)
        train.seed_everything()
        s = Settings(os.path.join(CFG, "openess_dsec_synthetic.yaml"), generate_log=False)
        s.ckpt_dir = str(tmp_path)
        s.if_spatial_contrastive = True
        s.lr_recon = s.lr_frame = 1e-4
        trainer, _ = train.build_trainer(s)
        assert isinstance(trainer, OpenESSModel)
        trainer.two_streams = two
        K, (H, W) = s.semseg_num_classes, s.img_size_b
        for name in ('model_recon', 'model_frame'):
            m = trainer.models_dict[name]
            fill_by_name(m, 500 + len(name) + (7 if name == 'model_frame' else 0))
            damp_residual(m)
            m.classifier.ASPP.project[3].p = 0.0
        g = torch.Generator().manual_seed(8)
        B = 2
        frame, recon = torch.rand(B, 3, H, W, generator=g).cuda(), torch.rand(B, 3, H, W, generator=g).cuda()
        pl = torch.randint(0, K, (B, H // 4, W // 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2).cuda()
        sp = torch.randint(0, 45, (B, H // 8, W // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda()
        rec = []
        for _ in range(2):
            losses, _, total = trainer.train_step((frame, None, recon, pl, sp, None))
            rec.append({k: float(v) for k, v in losses.items()})
        w = {f"{k}.{n}": p.detach().clone() for k, m in trainer.models_dict.items() for n, p in m.state_dict().items()}
        runs.append((rec, w))
        torch.randn(1 << 20, device="cuda").sum()
    for other in (1, 2):
        assert runs[0][0] == runs[other][0], (runs[0][0], runs[other][0])
        for n in runs[0][1]:
            assert torch.equal(runs[0][1][n], runs[other][1][n]), (other, n)
