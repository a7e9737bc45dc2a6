"""Randomised whole-model A/B: one bf16 training step of a segmentation network on a random batch / image size / base
   width, once with every specialised path on (LDS-tiled and 128-channel 3x3 kernels, streaming 1x1 kernels, BatchNorm
   sums in backward-data and max-pool epilogues, BatchNorm folded into the 1x1 layers, fused skip / residual gradients,
   two streams) and once with all of them off (generic implicit GEMM, separate passes, one stream).  Logits, loss and
   parameter gradients must agree to bf16 rounding: an indexing error at a ragged size shows up as a difference of order one.
       python tools/stress_models.py [cases] [seed]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd
from uresnet_amd import ops, train
from uresnet_amd.modules import _blocks
dev = torch.device("cuda:0")

SWITCHES = [(ops, "use_tile_kernel"), (ops, "use_wide_kernel"), (ops, "use_pw_kernel"), (ops, "fuse_bn_bwd_sums"),
            (ops, "fuse_pool_bn_sums"), (ops, "fuse_bn_into_conv1x1"), (ops, "branch_streams"), (_blocks, "fuse_pool_skip"),
            (_blocks, "fuse_residual_grad"), (ops, "fuse_merge_into_head"), (ops, "fuse_bn_pool"), (ops, "fuse_bn_out_stats"),
            (ops, "fuse_head_bn_sums"), (ops, "fuse_head_bn_apply"), (ops, "fuse_ce_head_grad")]


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def step(ctor, args, x, y, fast):
    saved = [(m, n, getattr(m, n)) for m, n in SWITCHES]
    for m, n in SWITCHES:
        setattr(m, n, fast)
    try:
        torch.manual_seed(0)
        net = ctor(*args).set_compute_dtype(torch.bfloat16).to(dev).train()
        out = net(x)
        loss = train.compute_loss(out, y)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters() if p.grad is not None}
        return out.detach().float().cpu(), float(loss.detach()), grads
    finally:
        for m, n, v in saved:
            setattr(m, n, v)


BACKWARD_SWITCHES = [(ops, "fuse_bn_bwd_sums"), (ops, "fuse_pool_bn_sums"), (_blocks, "fuse_pool_skip"), (_blocks, "fuse_residual_grad")]


def block_cases(n_cases, seed):
    """Chains of blocks at random channel counts and ragged sizes, the backward-pass fusions on and off.  The forward kernels
    are the same in both runs, so the ReLU masks are identical and the gradients agree to the order of the fp32 sums."""
    B = _blocks
    rnd = random.Random(seed + 1)
    worst = 0.0
    for case in range(n_cases):
        c0, c1, c2 = (rnd.choice([32, 64, 128, 256]) for _ in range(3))
        kind = rnd.choice(["double", "basic", "basic2", "double_down", "down_down", "basic_double"])
        torch.manual_seed(case)
        chain = {"double": lambda: [B.DoubleConv(c0, c1)], "basic": lambda: [B.BasicBlock(c0, c1)],
                 "basic2": lambda: [B.BasicBlock(c0, c1), B.BasicBlock(c1, c2)],
                 "double_down": lambda: [B.DoubleConv(c0, c1), B.Down(c1, c2)],
                 "down_down": lambda: [B.Down(c0, c1), B.Down(c1, c2)],
                 "basic_double": lambda: [B.BasicBlock(c0, c1), B.DoubleConv(c1, c2)]}[kind]()
        block = torch.nn.Sequential(*chain).to(dev).train()
        with torch.no_grad():
            for m in block.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        n, h, w = rnd.choice([1, 2, 3]), rnd.randint(4, 70), rnd.randint(4, 90)
        x0 = torch.randn(n, c0, h, w, device=dev)
        res = {}
        for fused in (False, True):
            saved = [(m, k, getattr(m, k)) for m, k in BACKWARD_SWITCHES]
            for m, k in BACKWARD_SWITCHES:
                setattr(m, k, fused)
            try:
                block.zero_grad(set_to_none=True)
                x = ops.as_activation(x0, torch.bfloat16).detach().requires_grad_(True)
                out = block(x)
                torch.manual_seed(case + 77)
                g0 = torch.randn(out.shape, device=dev)
                out.backward(ops.as_activation(g0, torch.bfloat16))
                torch.cuda.synchronize()
                res[fused] = (ops.dense(out.detach()).float().cpu(), ops.dense(x.grad).float().cpu(),
                              {k: p.grad.float().cpu() for k, p in block.named_parameters()})
            finally:
                for m, k, v in saved:
                    setattr(m, k, v)
        same_fwd = torch.equal(res[True][0], res[False][0])
        e_dx = rel(res[True][1], res[False][1])
        e_p = max(rel(res[True][2][k], res[False][2][k]) for k in res[True][2])
        worst = max(worst, e_dx, e_p)
        print("block case %2d %-12s %3d -> %3d -> %3d  %d x %2d x %2d  input gradient %.2e  parameter gradients max %.2e" % (
            case, kind, c0, c1, c2, n, h, w, e_dx, e_p), flush=True)
        if not same_fwd or e_dx > 4e-3 or e_p > 4e-3 or not torch.isfinite(res[True][1]).all():
            print("MISMATCH in block case %d (forward identical: %s)" % (case, same_fwd))
            sys.exit(1)
    print("block stress OK: %d cases (seed %d), worst gradient difference %.2e" % (n_cases, seed, worst))


HEAD_SWITCHES = [(ops, "fuse_head_bn_sums"), (ops, "fuse_head_bn_apply"), (ops, "fuse_ce_head_grad")]


def head_cases(n_cases, seed):
    """BatchNorm -> ReLU -> 1x1 convolution heads over one tensor or a pair of tensors at random channel counts and ragged pixel
    counts: backward with the BatchNorm sums from the backward-weight kernel and the apply pass in the backward-data kernel
    against the separate passes (same forward, so the masks are identical)."""
    B = _blocks
    rnd = random.Random(seed + 2)
    worst = 0.0
    for case in range(n_cases):
        ca, cb = rnd.choice([8, 16, 32, 64]), rnd.choice([0, 8, 16, 24])
        if ca + cb > 80:
            cb = 80 - ca
        classes = rnd.choice([3, 8, 16, 19, 21])
        n, h, w = rnd.choice([1, 2]), rnd.randint(190, 330), rnd.randint(181, 350)
        torch.manual_seed(case)
        head = B.OutConv(ca + cb, classes).to(dev).train()
        with torch.no_grad():
            head[0].weight.uniform_(0.5, 1.5); head[0].bias.uniform_(-0.3, 0.3)
        a0, b0 = torch.randn(n, ca, h, w, device=dev) + 0.2, torch.randn(n, max(cb, 8), h, w, device=dev)[:, :cb] * 0.7
        g0 = torch.randn(n, classes, h, w, device=dev)
        res = {}
        for fused in (False, True):
            saved = [(m, k, getattr(m, k)) for m, k in HEAD_SWITCHES]
            for m, k in HEAD_SWITCHES:
                setattr(m, k, fused)
            try:
                head.zero_grad(set_to_none=True)
                xa = ops.as_activation(a0, torch.bfloat16).detach().requires_grad_(True)
                if cb:
                    xb = ops.as_activation(b0.contiguous(), torch.bfloat16).detach().requires_grad_(True)
                    out = ops.bn_relu_conv1x1_pair(xa, xb, head[0], head[2])
                else:
                    xb = None
                    out = ops.bn_relu_conv1x1(xa, head[0], head[2])
                if out is None:
                    break
                out.backward(ops.as_activation(g0, torch.bfloat16))
                torch.cuda.synchronize()
                res[fused] = ([ops.dense(t.grad).float().cpu() for t in (xa, xb) if t is not None],
                              {k: p.grad.float().cpu() for k, p in head.named_parameters()})
            finally:
                for m, k, v in saved:
                    setattr(m, k, v)
        if len(res) < 2:
            print("head case %2d %2d + %2d -> %2d  %d x %3d x %3d  not covered by the streaming kernels" % (case, ca, cb, classes, n, h, w))
            continue
        e_dx = max(rel(a, b) for a, b in zip(res[True][0], res[False][0]))
        e_p = max(rel(res[True][1][k], res[False][1][k]) for k in res[True][1])
        worst = max(worst, e_dx, e_p)
        print("head case %2d %2d + %2d -> %2d  %d x %3d x %3d  input gradients %.2e  parameter gradients max %.2e" % (
            case, ca, cb, classes, n, h, w, e_dx, e_p), flush=True)
        if e_dx > 2e-3 or e_p > 6e-3 or not all(torch.isfinite(t).all() for t in res[True][0]):
            print("MISMATCH in head case %d" % case)
            sys.exit(1)
    print("head stress OK: %d cases (seed %d), worst gradient difference %.2e" % (n_cases, seed, worst))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    head_cases(n_cases, seed)
    block_cases(2 * n_cases, seed)
    rnd = random.Random(seed)
    U = uresnet_amd.modules.U_ResNet
    nets = [("UResNetNewEUCB", U.UResNetNewEUCB, 19), ("UResNet", U.UResNet, 19), ("UResNetNoEUCB", U.UResNetNoEUCB, 19),
            ("UNet", U.UNet, 19)]
    worst_out, worst_med = 0.0, 0.0
    for case in range(n_cases):
        name, ctor, classes = rnd.choice(nets)
        base = rnd.choice([8, 16, 32, 64])
        n = rnd.choice([1, 2, 3])
        h, w = 16 * rnd.randint(1, 11 if base < 64 else 6), 16 * rnd.randint(1, 15 if base < 64 else 8)
        g = torch.Generator().manual_seed(1000 * seed + case)
        x = torch.randn(n, 3, h, w, generator=g).to(dev)
        y = torch.randint(0, classes, (n, h, w), generator=g)
        y[torch.rand(n, h, w, generator=g) < 0.1] = 255
        y = y.to(dev)
        fast = step(ctor, (classes, 3, base), x, y, True)
        slow = step(ctor, (classes, 3, base), x, y, False)
        e_out = rel(fast[0], slow[0])
        assert fast[2].keys() == slow[2].keys(), "case %d: different sets of gradients" % case
        e_g = sorted(rel(fast[2][k], slow[2][k]) for k in fast[2])
        med = e_g[len(e_g) // 2]
        finite = bool(torch.isfinite(fast[0]).all()) and all(bool(torch.isfinite(t).all()) for t in fast[2].values())
        print("case %2d %-15s base %2d  %d x %3d x %3d  logits %.2e  loss %.5f / %.5f  gradients median %.2e max %.2e" % (
            case, name, base, n, h, w, e_out, fast[1], slow[1], med, e_g[-1]), flush=True)
        worst_out, worst_med = max(worst_out, e_out), max(worst_med, med)
        # the two runs round intermediate tensors differently, ReLU masks flip and 41 BatchNorm layers amplify that: whole-model
        # gradients differ by 0.1 - 0.3 in the median between any two correct bf16 implementations (the sharp gradient check
        # is block_cases); an indexing error in the forward pass moves the logits by order one
        if not finite or e_out > 4e-2 or abs(fast[1] - slow[1]) > 2e-2 * max(1.0, abs(slow[1])) or med > 0.4:
            print("MISMATCH in case %d" % case)
            sys.exit(1)
    print("model stress OK: %d cases (seed %d), worst logits difference %.2e, worst median gradient difference %.2e" % (
        n_cases, seed, worst_out, worst_med))


if __name__ == "__main__":
    main()
