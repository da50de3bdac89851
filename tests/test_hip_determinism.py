"""Run-to-run repeatability of the training path: the normalisation statistics used to be
fp32 atomics (arrival order) with E[x^2] - E[x]^2 in fp32.  They are now per-workgroup partials added in a FIXED order
in double, so the same call twice gives the same bits -- outputs, statistics, every gradient, and the whole pre-training
step's losses and weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cl(x):
    return x.bfloat16().contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("kind,shape", [("bn", (8, 64, 110, 160)), ("bn", (4, 512, 28, 40)), ("in", (8, 64, 110, 160)),
                                        ("in", (2, 256, 55, 80))])
def test_norm_forward_backward_bit_repeatable(kind, shape):
    from openess_amd import hip
    torch.manual_seed(4)
    C = shape[1]
    x0 = cl(torch.randn(*shape, device="cuda") * 3 + 1.5)
    g = cl(torch.randn(*shape, device="cuda"))
    runs = []
    for _ in range(4):
        x = x0.clone().requires_grad_(True)
        if kind == "bn":
            bn = torch.nn.BatchNorm2d(C).cuda().train()
            y = hip.batch_norm_train(x, bn, relu=True)
            y.backward(g)
            runs.append((y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
        else:
            y = hip.instance_norm(x, relu=True)
            y.backward(g)
            runs.append((y.detach(), x.grad))
        torch.cuda.synchronize()
        # different amounts of unrelated work in between change the workgroup arrival order of the next run
        torch.randn(1 << (18 + len(runs)), device="cuda").sum()
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


def test_reduce_finalize_tile_stats_matches_double_sum_and_repeats():
    """oess_norm_reduce_finalize_tile_stats (conv-epilogue partials -> mean / rstd / scale / shift, one launch, last-ticket
    block) against a float64 NumPy sum of the same partials, over random (tiles, C) including the multi-slice sizes, many
    back-to-back calls on one scratch (fence-free cross-workgroup protocol): exact repeatability and
    1-ulp-class agreement with the double reference."""
    from openess_amd import _lib, hip
    lib = _lib.load()
    rng = np.random.default_rng(5)
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    for it in range(60):
        tiles = int(rng.choice([1, 3, 64, 65, 70, 137, 550, 1100, 2200, 2047]))
        C = int(rng.choice([8, 24, 64, 256, 520, 2048]))
        count = float(tiles * 128)
        part = torch.from_numpy(rng.normal(0.5, 1.0, (tiles, 2, C)).astype(np.float32))
        part[:, 1] = part[:, 1].abs() * 128 + part[:, 0] ** 2 / 128 + 1.0           # sum of squares >= (sum)^2 / n
        pd = part.to(dev)
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(dev)
        beta = torch.from_numpy(rng.normal(0, 1, C).astype(np.float32)).to(dev)
        outs = []
        for rep in range(3):
            stt = torch.empty((4, C), dtype=torch.float32, device=dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            sc = hip._stats_scratch(C, dev)
            _lib.check(lib.oess_norm_reduce_finalize_tile_stats(pd.data_ptr(), tiles, C, sc.buf64.data_ptr(), sc.tickets.data_ptr(),
                                                                count, 1e-5, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                                                rv.data_ptr(), 0.1, stt[0].data_ptr(), stt[1].data_ptr(),
                                                                stt[2].data_ptr(), stt[3].data_ptr(), st), "reduce_finalize")
            outs.append((stt.clone(), rm, rv))
            assert int(sc.tickets.abs().sum()) == 0                                  # tickets left zero for the next call
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
        S = part[:, 0].double().numpy().sum(0)
        Q = part[:, 1].double().numpy().sum(0)
        m = S / count
        var = np.maximum(Q / count - m * m, 0)
        r = 1 / np.sqrt(var + 1e-5)
        got = outs[0][0].cpu().numpy()
        np.testing.assert_allclose(got[0], m, rtol=2e-7, atol=1e-9)
        np.testing.assert_allclose(got[1], r, rtol=3e-7)
        np.testing.assert_allclose(got[2], gamma.cpu().numpy() * r.astype(np.float32), rtol=3e-7)
        np.testing.assert_allclose(outs[0][2].cpu().numpy(), 0.9 + 0.1 * var * count / (count - 1), rtol=1e-6)


@pytest.mark.parametrize("option,contr", [("frame2voxel", False), ("frame2recon", False), ("frame2voxel", True), ("frame2recon", True)])
def test_pretrain_step_bit_repeatable(option, contr):
    """The same two optimisation steps from the same weights, twice: identical losses, gradients and weights -- also for the
    contrastive configurations (the superpixel scatter-mean sums in 64-bit fixed point since round 4; EventPreprocessor's
    statistics are fixed-order partial rows)."""
    from openess_amd.training.pretrain_step import PretrainStep
    from tests.synth import damp_residual, fill_by_name
    B, H, W, nwin = 2, 64, 96, 3
    g = torch.Generator().manual_seed(21)
    ev = (torch.randn(B, nwin * 5, H, W, generator=g) * (torch.rand(B, nwin * 5, H, W, generator=g) > 0.7)).contiguous().cuda()
    frame = torch.rand(B, 3, H, W, generator=g).cuda()
    pl = torch.randint(0, 11, (B, H, W), generator=g).cuda()
    sp = torch.randint(0, 25, (B, H // 8, W // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda()
    first = ev if option == "frame2voxel" else frame
    S = int((sp.cpu() + torch.arange(B)[:, None, None] * 25).max()) + 1
    runs = []
    for rep in range(3):
        st = PretrainStep(config_option=option, img_size=(H, W), nr_events_data=nwin, if_spatial_contrastive=contr,
                          superpixel_size=25, lr=1e-4)
        # run 0: the frozen half (teacher encoder, E2VID encoder) on its own HIP streams inside each step (the default); run 1: one
        # stream; run 2: streams AND the software pipeline -- front(step 1) enqueued before the trainable half of step 0
        # (PretrainStep.pipeline_steps).  Same kernels on the same values, ordered by events -> bit-identical.
        st.overlap_teacher = rep != 1
        for name, m in st.models_dict.items():
            fill_by_name(m, 100 + len(name))
            damp_residual(m)
        if option == "frame2recon":
            st.model_recon.classifier.ASPP.project[3].p = 0.0        # dropout off: the Philox stream differs between the two runs
        rec = []
        batch = (first, None, frame, pl, sp, S)
        if rep == 2:
            def back(b, fr):
                for opt in st.optimizers_dict.values():
                    opt.zero_grad()
                t_loss, losses, _ = st.task_train_step(b, front=fr)
                t_loss.backward()
                for opt in st.optimizers_dict.values():
                    opt.step()
                return losses
            for losses in st.pipeline_steps([batch, batch], back):
                rec.append({k: float(v) for k, v in losses.items()})
        else:
            for it in range(2):
                losses, _, tl = st.train_step(batch)
                rec.append({k: float(v) for k, v in losses.items()})
        w = {f"{k}.{n}": p.detach().clone() for k, m in st.models_dict.items() for n, p in m.named_parameters() if p.requires_grad}
        w.update({f"model_frame.buf.{n}": b.detach().clone() for n, b in st.model_frame.named_buffers()})     # BatchNorm running stats
        gr = {f"{k}.{n}": (None if p.grad is None else p.grad.detach().clone()) for k, m in st.models_dict.items()
              for n, p in m.named_parameters() if p.requires_grad}
        runs.append((rec, w, gr))
        torch.randn(1 << (20 + rep), device="cuda").sum()
    for other in (1, 2):
        assert runs[0][0] == runs[other][0], (runs[0][0], runs[other][0])
        for n in runs[0][1]:
            assert torch.equal(runs[0][1][n], runs[other][1][n]), (other, n)
        for n in runs[0][2]:
            a, b = runs[0][2][n], runs[other][2][n]
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), (other, n)


def test_batchnorm_backward_merged_launch_equals_three_launch_path(tmp_path):
    """oess_batchnorm_bwd_nhwc_bf16: for C % 64 == 0 the fixed-order reduction of the partial sums and the apply pass run in one
    launch (bn_bwd_reduce_apply_kernel); d(gamma), d(beta), dx and d(residual) must be bit-identical to the three-launch path,
    which a child process runs with OESS_BN_BWD_THREE_LAUNCHES=1 on the same tensors."""
    import os
    import subprocess
    import sys
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from openess_amd import hip
torch.manual_seed(0)
outs = {}
for tag, (B, C, H, W, relu, res) in {"a": (8, 1024, 28, 40, True, True), "b": (2, 64, 37, 53, True, False), "c": (3, 192, 9, 11, False, False),
                                     "d": (8, 256, 110, 160, True, True)}.items():
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    y = hip.batch_norm_train(x, bn, relu=relu, residual=r)
    g = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y.backward(g)
    outs[tag] = [x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()] + ([r.grad.clone()] if res else [])
torch.save({k: [t.cpu() for t in v] for k, v in outs.items()}, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, env in (("merged", {}), ("three", {"OESS_BN_BWD_THREE_LAUNCHES": "1"})):
        out = str(tmp_path / f"{name}.pt")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=dict(os.environ, **env), timeout=600)
        got[name] = torch.load(out)
    for tag in got["merged"]:
        for a, b in zip(got["merged"][tag], got["three"][tag]):
            assert torch.equal(a, b), tag
