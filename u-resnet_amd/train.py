"""Loss, mask and optimiser-step counterparts of the reference's training loops, on HIP kernels.

  compute_loss      Segmentation/Cityscapes/train_UResNetAblitionStudy.py:62-68
  Adam / SGD        optim.Adam(lr=1e-4) (same file :203), optim.SGD(lr=0.01) (Classification/CNN-CIFAR10.py:130)
  train steps       train_UResNetAblitionStudy.py:115-122, CNN-CIFAR10.py:145-149, Ice-Seg/TrainingCode.py:107-115
  masks             preds.argmax(1) (train_UResNetAblitionStudy.py:73), sigmoid(outputs) > 0.5 (TrainingCode.py:132)
  metrics           calculate_metrics / validate (train_UResNetAblitionStudy.py:70-100, 128-164)
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


class _MultiTensorOptimizer(torch.optim.Optimizer):
    """One kernel launch per step over every parameter that has a gradient (fp32 master weights)."""
    _slots = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._table_key = None
        self._table = None
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
            flat = torch.tensor([v for r in rows for v in r], dtype=torch.int64)
            sz = torch.tensor(sizes, dtype=torch.int64)
            self._table = (flat.to(device, non_blocking=False), sz.to(device, non_blocking=False), max(sizes))
            self._table_key = key
        return self._table


class Adam(_MultiTensorOptimizer):
    """torch.optim.Adam semantics (no amsgrad): exp_avg / exp_avg_sq state, bias correction, eps outside sqrt."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            ps = self._gather(group)
            if not ps:
                continue
            rows, sizes = [], []
            for p in ps:
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                rows.append((p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()))
                sizes.append(p.numel())
            step = self.state[ps[0]]['step']
            table, sz, mx = self._pointer_table(rows, sizes, ps[0].device)
            b1, b2 = group['betas']
            _lib.call("ur_adam_multi", table.data_ptr(), sz.data_ptr(), len(ps), mx, group['lr'], b1, b2, group['eps'],
                      group['weight_decay'], step, self.grad_scale, ops._stream())
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
