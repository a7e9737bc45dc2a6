"""Data parallelism: one process per GPU, gradient buckets all-reduced over RCCL (xGMI) while the
backward pass is still producing the remaining weight gradients.

Replaces the reference's single-process `torch.nn.DataParallel` wrap
(Segmentation/Cityscapes/train_UResNetAblitionStudy.py:197-198) and keeps its semantics:
  * every replica normalises with its OWN batch statistics (no SyncBN), and
  * the loss is the mean over the valid pixels of the GLOBAL batch (DataParallel computes the
    loss on gathered outputs): `count_hook` sums the per-rank valid-pixel counts (one scalar
    all-reduce) before the backward pass scales d(logits), and gradients are then SUMMED.
Parameters that receive no gradient (the reference's classification nets construct unused
layers) keep `grad is None` on every rank.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU.
"""
import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64                 # keep every slice 256-byte aligned
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.pending = len(params)
        self.work = None
        self.had_grad = [False] * len(params)
        self.events = {}            # stream handle -> event recorded after the last gradient that stream delivered

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)


class DataParallel:
    """Bucketed, overlapped gradient all-reduce for a replicated model.

    usage per step:
        out = ddp.module(x); loss = compute_loss(out, y, ddp.count_hook)
        optimizer.zero_grad(); loss.backward(); ddp.finish(); optimizer.step()

    Buckets start in reverse registration order (head first, stem last).  The first step records the order in which
    gradients really arrive -- the two branches of a U-ResNet interleave in backward, which registration order does
    not show -- and the buckets are rebuilt once in that order, so that each one completes, and its all-reduce
    starts, as early as the backward pass allows.

    The backward-weight kernels write straight into their bucket slice (ops.grad_buffer asks `p._ur_grad_slot`), so
    the large gradients are never copied into the bucket; gradients that arrive as separate tensors (biases,
    BatchNorm parameters, padded layers) are copied by the hook.

    Every rank must produce gradients for the same set of parameters (true for a replicated model with static
    control flow): buckets in which no parameter received a gradient are skipped, and a rank that skipped a bucket
    another rank reduces would leave the collective sequences out of step.  The first step checks this across ranks
    and raises instead of deadlocking later.
    """

    def __init__(self, module, process_group=None, bucket_mb=32.0, average=False, broadcast=True, reorder=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch one process per GPU)")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.average = average                  # True: mean of per-rank gradients; False: sum
        self.bucket_mb = bucket_mb
        params = [p for p in module.parameters() if p.requires_grad]
        if broadcast:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0, group=process_group)
            from . import ops
            ops.bump_weights_epoch()            # packed weight images made before the broadcast are stale now
        self._where = {}
        self._build(list(reversed(params)))
        for p in params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        self._armed = True
        self._arrival = [] if reorder else None  # parameters in the order their gradients arrived (first step only)
        self._checked = False
        self.bucket_order_source = "reverse registration"
        self.in_place_elems = self.copied_elems = 0      # gradient elements written into the buckets by the kernels / copied

    def _build(self, ordered):
        cap = int(self.bucket_mb * (1 << 20) / 4)
        self.buckets, cur, size = [], [], 0
        for p in ordered:
            if cur and (size + p.numel() > cap or cur[0].device != p.device):
                self.buckets.append(_Bucket(cur, cur[0].device))
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device))
        self._where.clear()
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                p._ur_grad_slot = (lambda b=b, i=i: b.view(i))

    # -- loss normalisation over the global batch ------------------------------------------
    def count_hook(self, out3):
        """out3 = [local NLL sum, local valid count, local mean] -> global valid count (1-element tensor)."""
        count = out3[1:2].clone()
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group)
        return count

    # -- gradient buckets ---------------------------------------------------------------------
    def _launch(self, b):
        if not b.flat.is_cuda:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        # A network whose branches run on two streams delivers the gradients of one bucket from both.  The collective is
        # ordered after the stream this hook runs on; that stream first waits for the other stream's LAST DELIVERY into
        # this bucket (an event recorded when it arrived), not for everything the other branch has queued since.
        cur = torch.cuda.current_stream(b.flat.device)
        for handle, ev in b.events.items():
            if handle != cur.cuda_stream:
                cur.wait_event(ev)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        b, i = self._where[p]
        if self._arrival is not None:
            self._arrival.append(p)
        slot = b.view(i)
        if p.grad.data_ptr() != slot.data_ptr():     # not written in place by the backward kernel
            slot.copy_(p.grad)
            self.copied_elems += p.numel()
        else:
            self.in_place_elems += p.numel()
        b.had_grad[i] = True
        b.pending -= 1
        if slot.is_cuda:
            cur = torch.cuda.current_stream(slot.device)
            ev = b.events.get(cur.cuda_stream)
            if ev is None:
                ev = b.events[cur.cuda_stream] = torch.cuda.Event()
            ev.record(cur)
        if b.pending == 0 and self._checked:         # the first step launches from finish(), after the consistency check
            self._launch(b)

    def _check_same_gradient_sets(self):
        flags = torch.tensor([float(h) for b in self.buckets for h in b.had_grad], device=self.buckets[0].flat.device)
        lo, hi = flags.clone(), flags.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError("DataParallel: ranks produced gradients for different sets of parameters")

    def finish(self):
        """Wait for every bucket; parameters then read their reduced gradient straight from the bucket."""
        if not self._checked:
            self._check_same_gradient_sets()
            self._checked = True
        for b in self.buckets:
            if b.work is None:                   # some parameters never produced a gradient
                if not any(b.had_grad):
                    self._reset(b)
                    continue
                for i, had in enumerate(b.had_grad):
                    if not had:
                        b.view(i).zero_()
                self._launch(b)
        for b in self.buckets:
            if b.work is None:
                continue
            b.work.wait()
            if self.average:
                b.flat.mul_(1.0 / self.world)
            self._armed = False
            for i, p in enumerate(b.params):
                if b.had_grad[i]:
                    p.grad = b.view(i)
            self._armed = True
            self._reset(b)
        if self._arrival is not None:
            self._reorder()

    def _reorder(self):
        """Rebuild the buckets once, in the order the first backward pass delivered the gradients (parameters that got
        none keep their place at the end).  The reduced gradients of this step stay where they are: p.grad keeps the old
        bucket alive until the next zero_grad."""
        seen, order = set(), []
        for p in self._arrival:
            if id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        rest = [p for b in self.buckets for p in b.params if id(p) not in seen]
        self._arrival = None
        current = [p for b in self.buckets for p in b.params]
        if [id(p) for p in order + rest] != [id(p) for p in current]:
            self._build(order + rest)
            self.bucket_order_source = "gradient arrival order of the first step"

    @staticmethod
    def _reset(b):
        b.pending = len(b.params)
        b.work = None
        b.had_grad = [False] * len(b.params)

    def __call__(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def shard_batch(batch, rank, world):
    """Rank r takes samples [r*B/world, (r+1)*B/world) of a global batch (dim 0)."""
    b = batch.shape[0]
    if b % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (b, world))
    per = b // world
    return batch[rank * per:(rank + 1) * per]
