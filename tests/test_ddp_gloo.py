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


def test_shard_batch_rejects_uneven_batches():
    from uresnet_amd.ddp import shard_batch
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(5, 3), 0, 2)
    assert shard_batch(torch.arange(8).view(8, 1), 1, 4).flatten().tolist() == [2, 3]
