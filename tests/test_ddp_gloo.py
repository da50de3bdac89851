"""N > 1 path on CPU: world_size-2 gloo processes.  Checks (iv) of SURVEY.md section 4: the gradient all-reduce
makes a 2-rank step equal to the single-process step on the concatenated batch (for a loss that is a mean
over samples), identical initial weights after broadcast, disjoint seeded shards, None-grad parameters skipped."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openess_amd.training.ddp import GradAllReduce, broadcast_module_states, shard_indices
    torch.manual_seed(100 + rank)                        # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    unused = torch.nn.Linear(2, 2)                       # never used -> grad stays None
    broadcast_module_states([net, unused])
    w0 = [p.detach().clone() for p in net.parameters()]
    torch.manual_seed(7)
    data = torch.randn(8, 3, 6, 6)
    target = torch.randn(8, 2, 6, 6)
    idx = shard_indices(8, rank, world)
    opt = torch.optim.AdamW(list(net.parameters()) + list(unused.parameters()), lr=1e-2)
    red = GradAllReduce(list(net.parameters()) + list(unused.parameters()), world, bucket_bytes=64)
    assert len(red.buckets) >= 2                          # 64-byte buckets: several messages, launched from the grad hooks
    launched_in_backward = []
    for it in range(2):
        opt.zero_grad()
        red.prepare()                                     # .grad -> views into the flat message buffers
        loss = ((net(data[idx]) - target[idx]) ** 2).mean()
        loss.backward()
        launched_in_backward.append(sum(b.launched for b in red.buckets))
        assert all(p.grad.untyped_storage().data_ptr() == red.buckets[red._where[id(p)][0]].flat.untyped_storage().data_ptr()
                   for p in net.parameters())             # reduced in place: no cat / copy-back
        red()
        if it == 0:
            opt.step()
            w1 = [p.detach().clone() for p in net.parameters()]
    # the un-prepared path (optimizer.zero_grad() only) must give the same averaged gradient
    g_prepared = [p.grad.clone() for p in net.parameters()]
    opt.zero_grad()
    loss = ((net(data[idx]) - target[idx]) ** 2).mean()
    loss.backward()
    red()
    same = all(torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(g_prepared, net.parameters()))
    out[rank] = {"w0": w0, "w1": w1, "idx": idx, "launched_in_backward": launched_in_backward, "unprepared_same": same,
                 "unused_grad_none": all(p.grad is None for p in unused.parameters())}
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for x, y in zip(a["w0"], b["w0"]):
        assert torch.equal(x, y)                          # broadcast made the replicas identical
    for x, y in zip(a["w1"], b["w1"]):
        assert torch.allclose(x, y, atol=1e-7)            # and they stay identical after the step
    assert sorted(a["idx"].tolist() + b["idx"].tolist()) == list(range(8))
    assert a["unused_grad_none"] and b["unused_grad_none"]
    assert a["unprepared_same"] and b["unprepared_same"]
    assert min(a["launched_in_backward"]) >= 1            # at least one bucket's all-reduce went out UNDER backward (overlap)
    # single-process reference on the union of the shards
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    with torch.no_grad():
        for p, w in zip(net.parameters(), a["w0"]):
            p.copy_(w)
    torch.manual_seed(7)
    data = torch.randn(8, 3, 6, 6)
    target = torch.randn(8, 2, 6, 6)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    loss = ((net(data) - target) ** 2).mean()
    loss.backward()
    opt.step()
    for p, w in zip(net.parameters(), a["w1"]):
        assert torch.allclose(p.detach(), w, atol=1e-6)


def _accum_worker(rank, world, port, out):
    """Two backward() calls per step: overlap=True must refuse (the second in-place `grad +=` races with the in-flight
    collective), overlap=False must give the mean over ranks of the ACCUMULATED gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openess_amd.training.ddp import GradAllReduce
    torch.manual_seed(3)
    net = torch.nn.Linear(4, 3)
    torch.manual_seed(50 + rank)
    xa, xb = torch.randn(5, 4), torch.randn(5, 4)
    red = GradAllReduce(net.parameters(), world, bucket_bytes=1 << 20, overlap=True)
    net.zero_grad()
    red.prepare()
    net(xa).sum().backward()
    raised = False
    try:
        net(xb).sum().backward()
    except RuntimeError as e:
        raised = "second gradient accumulation" in str(e)
    for b in red.buckets:                                  # drain what the first backward launched
        if b.handle is not None:
            b.handle.wait()
    red.close()
    red2 = GradAllReduce(net.parameters(), world, bucket_bytes=1 << 20, overlap=False)
    net.zero_grad()
    red2.prepare()
    net(xa).sum().backward()
    net(xb).sum().backward()
    red2()
    out[rank] = {"raised": raised, "grad": [p.grad.clone() for p in net.parameters()],
                 "local": [xa.sum(0) + xb.sum(0), torch.full((3,), 10.0)]}
    dist.destroy_process_group()


def test_second_accumulation_is_refused_with_overlap_and_correct_without():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_accum_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["raised"] and b["raised"]
    want_w = ((a["local"][0] + b["local"][0]) / 2)[None].expand(3, 4)
    assert torch.allclose(a["grad"][0], want_w, atol=1e-5) and torch.allclose(b["grad"][0], want_w, atol=1e-5)
    assert torch.allclose(a["grad"][1], a["local"][1], atol=1e-6)
