"""Data-parallel path on CPU: 2 ranks over gloo exercise the bucketed all-reduce, the global
valid-pixel count exchange and the unused-parameter handling of uresnet_amd.ddp.DataParallel."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.b = nn.Conv2d(8, 5, 1)
        self.unused = nn.Linear(4, 4)          # never called: its grads must stay None

    def forward(self, x):
        return self.b(F.relu(self.a(x)))


def _data(world):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2 * world, 3, 6, 6, generator=g)
    y = torch.randint(0, 5, (2 * world, 6, 6), generator=g)
    y[torch.rand(y.shape, generator=g) < 0.3] = 255     # uneven numbers of valid pixels per rank
    return x, y


def _worker(rank, world, port, bucket_mb, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import uresnet_amd
        from uresnet_amd.ddp import DataParallel, shard_batch
        torch.manual_seed(100 + rank)            # different initial weights: the wrapper must broadcast rank 0's
        net = Net()
        ddp = DataParallel(net, bucket_mb=bucket_mb)
        assert ddp.init_broadcasts == 1          # all parameters (one dtype, one device) travel in one flat buffer
        x, y = _data(world)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        for step in range(2):                    # second step checks that bucket state resets
            net.zero_grad(set_to_none=True)
            out = ddp(xs)
            nll = F.cross_entropy(out, ys, ignore_index=255, reduction="sum")
            valid = (ys != 255).sum().float()
            out3 = torch.stack([nll.detach(), valid, nll.detach() / valid])
            count = ddp.count_hook(out3)
            (nll / count[0]).backward()
            ddp.finish()
        grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}
        if rank == 0:
            ret["grads"] = grads
            ret["weights"] = {k: v.clone() for k, v in net.state_dict().items()}
            ret["count"] = count.item()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("bucket_mb", [32.0, 0.0005])       # one bucket / many small buckets
def test_two_rank_gradients_equal_global_batch_gradients(bucket_mb):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_mb, ret), nprocs=world, join=True)
    # single-process reference: the loss DataParallel computes on gathered outputs (mean over all valid pixels)
    torch.manual_seed(100)
    net = Net()
    net.load_state_dict(ret["weights"])
    x, y = _data(world)
    F.cross_entropy(net(x), y, ignore_index=255).backward()
    assert ret["count"] == float((y != 255).sum())
    for k, p in net.named_parameters():
        got = ret["grads"][k]
        if k.startswith("unused"):
            assert got is None
        else:
            assert torch.allclose(got, p.grad, rtol=1e-5, atol=1e-7), k


class TwoBranch(nn.Module):
    """Two branches registered one after the other but evaluated layer by layer in turns, then merged: the backward
    pass delivers their gradients interleaved, which registration order does not show."""

    def __init__(self):
        super().__init__()
        self.a1, self.a2 = nn.Conv2d(3, 4, 3, padding=1), nn.Conv2d(4, 4, 3, padding=1)
        self.b1, self.b2 = nn.Conv2d(3, 4, 3, padding=1), nn.Conv2d(4, 4, 3, padding=1)
        self.head = nn.Conv2d(8, 5, 1)

    def forward(self, x):
        a = F.relu(self.a1(x))
        b = F.relu(self.b1(x))
        a = F.relu(self.a2(a))
        b = F.relu(self.b2(b))
        return self.head(torch.cat([a, b], 1))


def _order_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import uresnet_amd
        from uresnet_amd.ddp import DataParallel, shard_batch
        torch.manual_seed(7)
        net = TwoBranch()
        names = {p: k for k, p in net.named_parameters()}
        ddp = DataParallel(net, bucket_mb=0.0008)            # a few parameters per bucket
        before = [[names[p] for p in b.params] for b in ddp.buckets]
        arrival = []
        for p in net.parameters():
            p.register_post_accumulate_grad_hook(lambda p: arrival.append(names[p]))
        x, y = _data(world)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        grads = []
        for step in range(3):
            del arrival[:]
            net.zero_grad(set_to_none=True)
            F.cross_entropy(ddp(xs), ys, ignore_index=255).backward()
            ddp.finish()
            grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
        if rank == 0:
            ret["before"] = before
            ret["after"] = [[names[p] for p in b.params] for b in ddp.buckets]
            ret["arrival"] = list(arrival)
            ret["source"] = ddp.bucket_order_source
            ret["same"] = all(torch.equal(grads[0][k], grads[2][k]) for k in grads[0])
    finally:
        dist.destroy_process_group()


def test_buckets_follow_the_observed_gradient_arrival_order():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_order_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    flat_before = [k for b in ret["before"] for k in b]
    flat_after = [k for b in ret["after"] for k in b]
    assert flat_after == ret["arrival"]                      # rebuilt once, in the order backward delivers
    assert flat_after != flat_before                         # which is not reverse registration order here
    assert ret["source"].startswith("gradient arrival")
    branch = [k[0] for k in flat_after if k[0] in "ab"]
    assert "".join(branch).count("ab") + "".join(branch).count("ba") >= 3      # the two branches interleave
    assert ret["same"]                                       # same weights and data: the rebuild changed no gradient


def _mismatch_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import uresnet_amd
        from uresnet_amd.ddp import DataParallel
        torch.manual_seed(0)
        net = Net()
        ddp = DataParallel(net, bucket_mb=32.0)
        x, y = _data(1)
        out = ddp(x)
        if rank == 1:                                        # rank 1 also exercises the layer rank 0 never calls
            out = out + net.unused(torch.ones(1, 4)).sum()
        F.cross_entropy(out, y, ignore_index=255).backward()
        try:
            ddp.finish()
            ret[rank] = "no error"
        except RuntimeError as e:
            ret[rank] = str(e)
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_gradient_sets_raise_instead_of_hanging():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_mismatch_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert "different sets of parameters" in ret[0] and "different sets of parameters" in ret[1]


class TwoBranchOtherOrder(TwoBranch):
    """Same parameters and function, evaluated branch after branch: backward delivers the gradients in another order."""

    def forward(self, x):
        b = F.relu(self.b2(F.relu(self.b1(x))))
        a = F.relu(self.a2(F.relu(self.a1(x))))
        return self.head(torch.cat([a, b], 1))


def _disagree_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import uresnet_amd
        from uresnet_amd.ddp import DataParallel, shard_batch
        torch.manual_seed(7)
        net = TwoBranch() if rank == 0 else TwoBranchOtherOrder()
        ddp = DataParallel(net, bucket_mb=0.0008)
        x, y = _data(world)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        for step in range(3):
            net.zero_grad(set_to_none=True)
            out = ddp(xs)
            nll = F.cross_entropy(out, ys, ignore_index=255, reduction="sum")
            valid = (ys != 255).sum().float()
            count = ddp.count_hook(torch.stack([nll.detach(), valid, nll.detach() / valid]))
            (nll / count[0]).backward()
            ddp.finish()
        ret[rank] = {"layout": ddp.bucket_layout(), "differs": ddp.arrival_order_differs_from_rank0,
                     "grads": {k: p.grad.clone() for k, p in net.named_parameters()},
                     "weights": {k: v.clone() for k, v in net.state_dict().items()}}
    finally:
        dist.destroy_process_group()


def test_a_rank_whose_gradients_arrive_in_another_order_adopts_rank_0s_buckets():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_disagree_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0]["layout"] == ret[1]["layout"]              # same parameter at the same offset of the same bucket
    assert ret[0]["differs"] is False and ret[1]["differs"] is True
    net = TwoBranch()
    net.load_state_dict(ret[0]["weights"])
    x, y = _data(world)
    F.cross_entropy(net(x), y, ignore_index=255).backward()
    for k, p in net.named_parameters():                      # sums of matching gradients, on both ranks
        assert torch.allclose(ret[0]["grads"][k], p.grad, rtol=1e-5, atol=1e-7), k
        assert torch.equal(ret[0]["grads"][k], ret[1]["grads"][k]), k


class _ScaleInto(torch.autograd.Function):
    """y = x * w with the weight gradient written where ops.grad_buffer says, as the backward-weight kernels do."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, g):
        from uresnet_amd import ops
        x, w = ctx.saved_tensors
        dw = ops.grad_buffer(w)
        dw.copy_((g * x).sum().reshape(w.shape))
        return g * w, dw


class AppliedTwice(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.tensor([1.5]))
        self.v = nn.Parameter(torch.tensor([-0.5]))

    def forward(self, x):
        return _ScaleInto.apply(_ScaleInto.apply(x, self.w) + 1.0, self.w) * self.v


def _twice_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import uresnet_amd
        from uresnet_amd.ddp import DataParallel
        net = AppliedTwice()
        ddp = DataParallel(net)
        x = torch.arange(1.0, 5.0) * (rank + 1)
        for step in range(2):
            net.zero_grad(set_to_none=True)
            ddp(x).sum().backward()
            ddp.finish()
        ret[rank] = (net.w.grad.clone(), net.v.grad.clone(), ddp.in_place_elems, ddp.copied_elems)
    finally:
        dist.destroy_process_group()


def test_a_parameter_used_by_two_backward_nodes_gets_the_sum_of_both_gradients():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_twice_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want_w, want_v = torch.zeros(1), torch.zeros(1)
    for rank in range(world):
        w, v = torch.tensor([1.5], requires_grad=True), torch.tensor([-0.5], requires_grad=True)
        x = torch.arange(1.0, 5.0) * (rank + 1)
        ((((x * w) + 1.0) * w) * v).sum().backward()
        want_w += w.grad
        want_v += v.grad
    for rank in range(world):
        assert torch.allclose(ret[rank][0], want_w, rtol=1e-6), (ret[rank][0], want_w)
        assert torch.allclose(ret[rank][1], want_v, rtol=1e-6)


def test_backend_and_device_pairings_that_cannot_work_are_refused():
    from uresnet_amd.ddp import check_backend
    check_backend("nccl", {"cuda"})
    check_backend("gloo", {"cpu"})
    check_backend("cpu:gloo,cuda:nccl", {"cpu", "cuda"})
    with pytest.raises(RuntimeError, match="RCCL"):
        check_backend("gloo", {"cuda"})                      # the pairing that hung in a two-process rehearsal on one GPU
    with pytest.raises(RuntimeError, match="gloo"):
        check_backend("nccl", {"cpu"})
    with pytest.raises(RuntimeError, match="unsupported"):
        check_backend("nccl", {"xpu"})


def test_shard_batch_rejects_uneven_batches():
    from uresnet_amd.ddp import shard_batch
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(5, 3), 0, 2)
    assert shard_batch(torch.arange(8).view(8, 1), 1, 4).flatten().tolist() == [2, 3]
