"""Loss, mask and optimiser-step counterparts of the reference's training loops, on HIP kernels.

  compute_loss      Segmentation/Cityscapes/train_UResNetAblitionStudy.py:62-68
  Adam / SGD        optim.Adam(lr=1e-4) (same file :203), optim.SGD(lr=0.01) (Classification/CNN-CIFAR10.py:130)
  train steps       train_UResNetAblitionStudy.py:115-122, CNN-CIFAR10.py:145-149, Ice-Seg/TrainingCode.py:107-115
  masks             preds.argmax(1) (train_UResNetAblitionStudy.py:73), sigmoid(outputs) > 0.5 (TrainingCode.py:132)
  metrics           calculate_metrics / validate (train_UResNetAblitionStudy.py:70-100, 128-164)
  checkpoints       the dict written at train_UResNetAblitionStudy.py:244-252 (save_checkpoint / load_checkpoint)
"""
import torch

from . import _lib, ops


def compute_loss(outputs, targets, count_hook=None):
    """CrossEntropyLoss(ignore_index=255)(outputs, targets.long()): mean NLL over the valid pixels."""
    if isinstance(outputs, list):
        outputs = outputs[0]
    if targets.dtype != torch.int64:
        targets = targets.long()
    return ops.cross_entropy(outputs, targets, ignore_index=255, count_hook=count_hook)


def cross_entropy_loss(outputs, targets):
    """nn.CrossEntropyLoss() on [B, classes] logits (CNN-CIFAR10.py:129)."""
    return ops.cross_entropy(outputs, targets, ignore_index=-100)


def bce_with_logits_loss(outputs, masks):
    """nn.BCEWithLogitsLoss() (Ice-Seg/TrainingCode.py:97)."""
    return ops.bce_with_logits(outputs, masks)


def argmax_mask(outputs):
    return ops.argmax_channels(outputs)


def sigmoid_mask(outputs):
    return ops.sigmoid_mask(outputs)


_METRIC_KEYS = ('precision', 'recall', 'accuracy', 'f1', 'iou')


def _metrics_dict(acc):
    v = acc.cpu().tolist()                       # the only host synchronisation
    n = v[5] if v[5] > 0 else 1.0
    m = {k: v[i] / n for i, k in enumerate(_METRIC_KEYS)}
    m['dice'] = m['f1']                          # the reference reports the F1 score again as `dice`
    return m, v[6] / n


def calculate_metrics(preds, targets, num_classes):
    """train_UResNetAblitionStudy.py:70-100: macro precision / recall / F1 / IoU and accuracy of argmax(preds) on the
    pixels whose target is not 255 -- from a confusion matrix built on the GPU instead of sklearn on flattened copies."""
    if isinstance(preds, list):
        preds = preds[0]
    cm = ops.confusion_matrix(preds, targets, num_classes, ignore_index=255)
    acc = torch.zeros(7, dtype=torch.float64, device=cm.device)
    ops.seg_metrics_accumulate(cm, acc)
    return _metrics_dict(acc)[0]


def validate(model, data_loader, device, num_classes):
    """train_UResNetAblitionStudy.py:128-164: eval-mode forward, mean loss and mean per-batch metrics.  Loss and metrics
    accumulate on the device; the host reads them once after the last batch."""
    model.eval()
    acc = None
    with torch.no_grad():
        for images, targets in data_loader:
            images, targets = images.to(device), targets.to(device)
            outputs = model(images)
            loss = compute_loss(outputs, targets)
            if isinstance(outputs, list):
                outputs = outputs[0]
            if acc is None:
                acc = torch.zeros(7, dtype=torch.float64, device=outputs.device)
                cm = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=outputs.device)
            cm.zero_()
            ops.confusion_matrix(outputs, targets, num_classes, ignore_index=255, out=cm)
            ops.seg_metrics_accumulate(cm, acc, loss.detach().float().reshape(1))
    if acc is None:
        raise ZeroDivisionError("validate: empty data loader")
    metrics, mean_loss = _metrics_dict(acc)
    return mean_loss, metrics


def save_checkpoint(path, model, optimizer, epoch, val_loss, metrics):
    """The reference's best-model file (train_UResNetAblitionStudy.py:244-252): same keys, `module.`-free parameter
    names, CPU tensors, torch.optim-compatible optimiser state."""
    inner = getattr(model, 'module', model)
    torch.save({'epoch': epoch,
                'model_state_dict': {k: v.detach().cpu() for k, v in inner.state_dict().items()},
                'optimizer_state_dict': optimizer.state_dict(),
                'val_loss': val_loss,
                'metrics': metrics}, path)


def load_checkpoint(path, model, optimizer=None, map_location='cpu'):
    """Load a checkpoint written by the reference's loop or by save_checkpoint (or a bare state_dict).  The file is read
    with weights_only=True (tensors and plain containers only, nothing in it is executed); a `module.` prefix left by
    nn.DataParallel is stripped.  Returns the checkpoint dict without the two state dicts."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    state = ckpt.get('model_state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    state = {(k[7:] if k.startswith('module.') else k): v for k, v in state.items()}
    getattr(model, 'module', model).load_state_dict(state, strict=True)
    if optimizer is not None and isinstance(ckpt, dict) and 'optimizer_state_dict' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return {k: v for k, v in ckpt.items() if k not in ('model_state_dict', 'optimizer_state_dict')} if 'model_state_dict' in ckpt else {}


class _MultiTensorOptimizer(torch.optim.Optimizer):
    """One kernel launch per step over every parameter that has a gradient (fp32 master weights)."""
    _slots = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._table_key = None
        self._table = None
        self._pin = None
        self._ring = None
        self._ring_next = 0
        self.grad_scale = 1.0

    def _gather(self, group):
        ps = [p for p in group['params'] if p.grad is not None]
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise _lib.UResNetHipError("optimizer: parameters must be contiguous fp32 GPU tensors")
            if p.grad.dtype != torch.float32:
                raise _lib.UResNetHipError("optimizer: gradients must be fp32")
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        return ps

    def _pointer_table(self, rows, sizes, device):
        key = tuple(v for r in rows for v in r)
        if key != self._table_key:
            flat, sz = torch.tensor(key, dtype=torch.int64), torch.tensor(sizes, dtype=torch.int64)
            fits = self._pin is not None and self._pin[0].numel() == len(key) and self._pin[1].numel() == len(sizes)
            if torch.cuda.is_current_stream_capturing():
                # Under HIP graph capture the upload must be an asynchronous copy from pinned memory (it becomes a graph
                # node that re-reads the staging buffer at every replay); the buffer was allocated by an eager step.
                if not fits:
                    raise RuntimeError("optimizer: run at least one eager step with the same parameters before capturing")
                self._pin[0].copy_(flat)
                self._pin[1].copy_(sz)
                self._table = (self._pin[0].to(device, non_blocking=True), self._pin[1].to(device, non_blocking=True), max(sizes))
            else:
                # Eager: gradient tensors are new allocations every step, so the table changes every step.  A blocking
                # upload would drain the GPU once per step (the host waits for the whole backward pass, then the next
                # forward starts from an empty queue).  Instead: asynchronous copies from a ring of pinned staging
                # buffers; a slot is rewritten only after the event behind its last copy has completed (four steps
                # later, so the wait never blocks in practice).
                if not fits:
                    self._pin = (torch.empty(len(key), dtype=torch.int64).pin_memory(),
                                 torch.empty(len(sizes), dtype=torch.int64).pin_memory())
                    self._ring = [(torch.empty(len(key), dtype=torch.int64).pin_memory(),
                                   torch.empty(len(sizes), dtype=torch.int64).pin_memory(), torch.cuda.Event())
                                  for _ in range(4)]
                    self._ring_next = 0
                pf, ps_, ev = self._ring[self._ring_next]
                self._ring_next = (self._ring_next + 1) % len(self._ring)
                ev.synchronize()
                pf.copy_(flat)
                ps_.copy_(sz)
                self._table = (pf.to(device, non_blocking=True), ps_.to(device, non_blocking=True), max(sizes))
                ev.record(torch.cuda.current_stream(device))
            self._table_key = key
        return self._table


class Adam(_MultiTensorOptimizer):
    """torch.optim.Adam semantics (no amsgrad): exp_avg / exp_avg_sq state, bias correction, eps outside sqrt.

    Every parameter's step count lives in device memory (one int32 slot of a per-group pool; `state[p]['step']` is a
    0-dim view of it) and is advanced by the kernel, so a step carries no host-side value: a captured HIP graph
    replays it correctly, and parameters whose state began later are corrected with their own count, as
    torch.optim.Adam does.  state_dict() / load_state_dict() use torch.optim.Adam's layout (`step` as a float32 CPU
    scalar tensor), so optimiser states move between the two in either direction."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_pools = {}          # (group index, device) -> int32 pool of step counters
        self._pool_index = {}          # (group index, device) -> {parameter: slot}

    def _step_slot(self, gi, p):
        """0-dim int32 device view: the step counter of parameter p of group gi.  One pool per (group, device); a pool
        grows by copy when the group has gained parameters, so counters handed out earlier keep their values (their
        old views stay valid for the kernel table, which is rebuilt from `state[p]['step']` at every step)."""
        key = (gi, p.device)
        pool = self._step_pools.get(key)
        index = self._pool_index.setdefault(key, {})
        if p not in index:
            index[p] = len(index)
        need = index[p] + 1
        if pool is None or pool.numel() < need:
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("Adam: run at least one eager step before capturing")
            size = max(need, len(self.param_groups[gi]['params']))
            grown = torch.zeros(size, dtype=torch.int32, device=p.device)
            if pool is not None:
                grown[:pool.numel()].copy_(pool)
                for q, i in index.items():                  # counters already handed out move with their values
                    st = self.state.get(q)
                    if st and 'step' in st and i < pool.numel():
                        st['step'] = grown[i]
            pool = self._step_pools[key] = grown
        return pool[index[p]]

    def state_dict(self):
        sd = super().state_dict()
        # one device read per pool instead of one per parameter
        host = {key: pool.cpu() for key, pool in self._step_pools.items()}
        slot = {q: (key, i) for key, index in self._pool_index.items() for q, i in index.items()}
        order = [q for group in self.param_groups for q in group['params']]      # torch numbers the parameters this way

        def step_of(k, v):
            q = order[k]
            if q in slot and slot[q][0] in host:
                return float(host[slot[q][0]][slot[q][1]])
            return float(v['step'])
        sd['state'] = {k: ({**v, 'step': torch.tensor(step_of(k, v))} if 'step' in v else v) for k, v in sd['state'].items()}
        for g in sd['param_groups']:                 # torch.optim.Adam's remaining group keys, at their defaults
            for k, v in (('amsgrad', False), ('maximize', False), ('foreach', None), ('capturable', False),
                         ('differentiable', False), ('fused', None), ('decoupled_weight_decay', False)):
                g.setdefault(k, v)
        return sd

    def load_state_dict(self, state_dict):
        for g in state_dict['param_groups']:
            if g.get('amsgrad') or g.get('maximize'):
                raise NotImplementedError("Adam: amsgrad / maximize states cannot be loaded")
        keep = set(self.defaults) | {'params'}
        slim = {'state': state_dict['state'],
                'param_groups': [{k: v for k, v in g.items() if k in keep} for g in state_dict['param_groups']]}
        super().load_state_dict(slim)
        self._step_pools, self._pool_index = {}, {}
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                st = self.state.get(p)
                if st and 'step' in st:
                    count = int(round(float(st['step'])))
                    st['step'] = self._step_slot(gi, p)
                    st['step'].fill_(count)
        self._table_key = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            ps = self._gather(group)
            if not ps:
                continue
            rows, sizes = [], []
            for p in ps:                                  # states first: a new counter may move the whole pool
                st = self.state[p]
                if not st:
                    st['step'] = self._step_slot(gi, p)
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
            for p in ps:
                st = self.state[p]
                rows.append((p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                             st['step'].data_ptr()))
                sizes.append(p.numel())
            table, sz, mx = self._pointer_table(rows, sizes, ps[0].device)
            b1, b2 = group['betas']
            _lib.call("ur_adam_multi_dev", table.data_ptr(), sz.data_ptr(), len(ps), mx, group['lr'], b1, b2, group['eps'],
                      group['weight_decay'], self.grad_scale, ops._stream())
        ops.bump_weights_epoch()
        return loss


class SGD(_MultiTensorOptimizer):
    """torch.optim.SGD semantics without momentum (the reference uses plain SGD)."""

    def __init__(self, params, lr=1e-3, momentum=0, weight_decay=0.0):
        if momentum != 0:
            raise NotImplementedError("SGD momentum is not implemented (the reference trains with momentum=0)")
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = self._gather(group)
            if not ps:
                continue
            rows = [(p.data_ptr(), p.grad.data_ptr()) for p in ps]
            sizes = [p.numel() for p in ps]
            table, sz, mx = self._pointer_table(rows, sizes, ps[0].device)
            _lib.call("ur_sgd_multi", table.data_ptr(), sz.data_ptr(), len(ps), mx, group['lr'], group['weight_decay'],
                      self.grad_scale, ops._stream())
        ops.bump_weights_epoch()
        return loss


class GraphedStep:
    """One training step captured as a HIP graph and replayed: for small inputs the step is bound by the few hundred
    host-side launches, not by the GPU (CIFAR-shaped batch: 9 ms eager).  `step_fn(model, optimizer, *inputs)` is one of
    the *_train_step functions; inputs are copied into static buffers before each replay, the returned loss / outputs
    are static tensors that the next replay overwrites.  Works with this module's SGD and Adam (whose step counts live on
    the device); shapes are fixed at construction; BatchNorm running statistics and counters advance on the device as in
    eager mode."""

    def __init__(self, step_fn, model, optimizer, *example_inputs, warmup=3):
        import gc
        gc.collect()                                # dead models drop out of the shared weight-image table now, not mid-capture
        self._static = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                 # allocator pools, weight images and pointer tables reach steady state
                step_fn(model, optimizer, *self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._out = step_fn(model, optimizer, *self._static)
        self.warmup_steps = warmup

    def __call__(self, *inputs):
        for dst, src in zip(self._static, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replayed update rewrote the parameters through raw pointers: weight images cached by eager forwards
        # (an evaluation pass between training epochs) are stale now, and nothing else tells the caches so
        ops.bump_weights_epoch()
        return self._out


def seg_train_step(model, optimizer, images, targets, count_hook=None, grad_hook=None):
    """forward -> CE(ignore 255) -> zero_grad -> backward -> step; returns (loss, outputs) without syncing."""
    outputs = model(images)
    loss = compute_loss(outputs, targets, count_hook)
    optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook()
    optimizer.step()
    return loss.detach(), outputs.detach()


def cls_train_step(model, optimizer, data, target, grad_hook=None):
    optimizer.zero_grad()
    output = model(data)
    loss = cross_entropy_loss(output, target)
    loss.backward()
    if grad_hook is not None:
        grad_hook()
    optimizer.step()
    return loss.detach(), output.detach()


def ice_train_step(model, optimizer, images, masks, grad_hook=None):
    optimizer.zero_grad()
    outputs = model(images)
    loss = bce_with_logits_loss(outputs, masks)
    loss.backward()
    if grad_hook is not None:
        grad_hook()
    optimizer.step()
    return loss.detach(), outputs.detach()
