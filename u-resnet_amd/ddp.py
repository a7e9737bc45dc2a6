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

Backends: "nccl" (= RCCL on ROCm) for parameters on GPUs, "gloo" for parameters on the CPU.  Any other pairing is
refused at construction (check_backend): gloo collectives on device tensors of processes that share a GPU were seen
to hang in the first broadcast instead of failing.
"""
import torch
import torch.distributed as dist


def check_backend(backend, device_types):
    """Raise unless `backend` (the process group's backend name) can reduce tensors on every device type in
    `device_types` (e.g. {"cuda"}): GPU parameters need "nccl", CPU parameters need a backend with CPU support
    ("gloo", or a combined "cpu:gloo,cuda:nccl" group)."""
    backend = str(backend).lower()
    for kind in sorted(set(device_types)):
        if kind == "cuda" and "nccl" not in backend:
            raise RuntimeError("DataParallel: parameters live on a GPU but the process group's backend is %r; GPU tensors "
                               "are reduced over RCCL only: init_process_group('nccl') (gloo on device tensors can hang)"
                               % backend)
        if kind == "cpu" and "gloo" not in backend and "mpi" not in backend:
            raise RuntimeError("DataParallel: parameters live on the CPU but the process group's backend is %r, which "
                               "reduces GPU tensors only: use init_process_group('gloo')" % backend)
        if kind not in ("cuda", "cpu"):
            raise RuntimeError("DataParallel: unsupported parameter device type %r" % kind)


def broadcast_coalesced(tensors, src=0, group=None):
    """Overwrite every tensor with rank `src`'s values: one broadcast per (device, dtype) instead of one per tensor
    (a U-ResNet has about 400 parameters and buffers)."""
    families = {}
    for t in tensors:
        families.setdefault((t.device, t.dtype), []).append(t)
    with torch.no_grad():
        for members in families.values():
            flat = torch.cat([t.detach().reshape(-1) for t in members])
            dist.broadcast(flat, src=src, group=group)
            at = 0
            for t in members:
                t.copy_(flat[at:at + t.numel()].view_as(t))
                at += t.numel()
    return len(families)


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
        self.idle = False           # no parameter of the bucket received a gradient in the last step
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
        check_backend(dist.get_backend(process_group), {p.device.type for p in params})
        self._params = params                   # registration order: what the ranks exchange indices into
        if broadcast:
            self.init_broadcasts = broadcast_coalesced(list(module.parameters()) + list(module.buffers()), 0, process_group)
            from . import ops
            ops.bump_weights_epoch()            # packed weight images made before the broadcast are stale now
        self._where = {}
        self._slot_taken = set()                # parameters whose bucket slice a backward kernel has been handed this step
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
        self._slot_taken.clear()
        self._next = 0                          # buckets [0, _next) have had their all-reduce issued in this step
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                p._ur_grad_slot = (lambda b=b, i=i, p=p: self._take_slot(b, i, p))

    def _take_slot(self, b, i, p):
        """The parameter's bucket slice, at most once per step: a parameter that two backward nodes produce gradients
        for (a module applied twice) must not have both kernels write the same memory -- autograd would then sum two
        views of the second result.  The second asker gets None (ops.grad_buffer allocates) and the hook copies the sum."""
        if id(p) in self._slot_taken:
            return None
        self._slot_taken.add(id(p))
        return b.view(i)

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
        self._slot_taken.discard(id(p))
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
            self._launch_ready()

    def _launch_ready(self):
        """Start the all-reduce of every complete bucket that all earlier buckets have gone before.  Collectives must be
        issued in one order on every rank; the order in which buckets complete is a property of the rank (its autograd
        schedule), the bucket numbering is not."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.idle and b.pending == len(b.params):        # received nothing in the last step either: finish() skips it
                self._next += 1
                continue
            if b.pending != 0 or b.work is not None:
                break
            self._launch(b)
            self._next += 1

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
        for b in self.buckets:                   # in bucket order, whatever _launch_ready has started already
            if b.work is None:                   # first step, or some parameters never produced a gradient
                if not any(b.had_grad):
                    b.idle = True
                    self._reset(b)
                    continue
                b.idle = False
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
        self._next = 0
        if self._arrival is not None:
            self._reorder()

    def _reorder(self):
        """Rebuild the buckets once, in the order the first backward pass delivered the gradients ON RANK 0 (parameters
        that got none keep their place at the end).  Every rank adopts rank 0's order, sent as indices into the
        registration order: the ranks' own orders may differ (autograd scheduling, two streams), and buckets laid out
        from differing orders would add unrelated gradients at equal offsets without any error.  The reduced gradients
        of this step stay where they are: p.grad keeps the old bucket alive until the next zero_grad."""
        index = {id(p): k for k, p in enumerate(self._params)}
        seen, mine = set(), []
        for p in self._arrival:
            if id(p) not in seen:
                seen.add(id(p))
                mine.append(index[id(p)])
        mine += [index[id(p)] for b in self.buckets for p in b.params if id(p) not in seen]
        self._arrival = None
        dev = self.buckets[0].flat.device
        order = torch.tensor(mine, dtype=torch.int64, device=dev)
        dist.broadcast(order, src=0, group=self.group)
        # what was received must be a permutation of this rank's parameters, and the same on every rank
        lo, hi = order.clone(), order.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        adopted = order.tolist()
        if not torch.equal(lo, hi) or sorted(adopted) != list(range(len(self._params))):
            raise RuntimeError("DataParallel: the ranks do not agree on the bucket order (replicas differ in their parameters)")
        self.arrival_order_differs_from_rank0 = adopted != mine
        current = [index[id(p)] for b in self.buckets for p in b.params]
        if adopted != current:
            self._build([self._params[k] for k in adopted])
            self.bucket_order_source = "gradient arrival order of the first step on rank 0"

    def bucket_layout(self):
        """[(registration index, bucket number, offset)] of every parameter: equal on all ranks by construction."""
        index = {id(p): k for k, p in enumerate(self._params)}
        return [(index[id(p)], n, b.offsets[i]) for n, b in enumerate(self.buckets) for i, p in enumerate(b.params)]

    def _reset(self, b):
        b.pending = len(b.params)
        b.work = None
        b.had_grad = [False] * len(b.params)
        for p in b.params:
            self._slot_taken.discard(id(p))

    def __call__(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def shard_batch(batch, rank, world):
    """Rank r takes samples [r*B/world, (r+1)*B/world) of a global batch (dim 0)."""
    b = batch.shape[0]
    if b % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (b, world))
    per = b // world
    return batch[rank * per:(rank + 1) * per]
