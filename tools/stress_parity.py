"""Randomised checks.  BatchNorm (+residual, ReLU, channel windows) against torch; and an A/B of the specialised
   convolution kernels against the generic implicit GEMM (same bf16 operands):
   streaming 1x1 kernel (forward, backward-data, channel windows) and the tiled 3x3 kernel on 32-multiple channel counts
   and ragged image sizes.    python tools/stress_parity.py [cases]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd
from uresnet_amd import ops
dev = torch.device("cuda:0")


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def run(x, w, b, k, pad, gy, out_win):
    xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xa = ops.as_activation(xd, torch.bfloat16)
    out = None
    if out_win is not None:
        buf = ops.empty_nhwc(x.shape[0], out_win[1], x.shape[2], x.shape[3], torch.bfloat16, dev).fill_(3.0)
        out = ops.alias_window(buf, out_win[0], w.shape[0])
    y, _ = ops.conv2d(xa, wd, b, ops.PackedWeights(), 1, pad, out=out)
    (y.float() * gy).sum().backward()
    extra = None
    if out_win is not None:
        keep = torch.ones(out_win[1], dtype=torch.bool, device=dev)
        keep[out_win[0]:out_win[0] + w.shape[0]] = False
        extra = bool((buf[:, keep] == 3.0).all())
    return ops.dense(y).detach().clone() if out_win is not None else y.detach(), xd.grad, wd.grad, extra


def bn_cases(n_cases):
    """BatchNorm + residual + ReLU forward/backward against torch.nn.BatchNorm2d in train mode: random channel counts
    (also not multiples of 4 / 8), image sizes, storage types, residual inputs and channel-window outputs."""
    from uresnet_amd import nn as hnn, _lib
    rnd = random.Random(11)
    worst = 0.0
    for case in range(n_cases):
        c = rnd.choice([3, 8, 16, 19, 24, 32, 40, 64, 80, 100, 128, 256, 520]); n = rnd.choice([1, 2, 4])
        h = rnd.randint(1, 33); w = rnd.randint(1, 47)
        dtype = rnd.choice([torch.float32, torch.bfloat16]); res = rnd.random() < 0.4
        act = rnd.choice([_lib.ACT_RELU, _lib.ACT_NONE])
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
        r = torch.randn(n, c, h, w, generator=g) if res else None
        gy = torch.randn(n, c, h, w, generator=g)
        if dtype == torch.bfloat16:
            x = x.bfloat16().float(); gy = gy.bfloat16().float(); r = r.bfloat16().float() if res else None
        bn_t = torch.nn.BatchNorm2d(c)
        with torch.no_grad():
            bn_t.weight.uniform_(0.5, 1.5); bn_t.bias.uniform_(-0.3, 0.3)
        xr = x.clone().requires_grad_(True); rr = r.clone().requires_grad_(True) if res else None
        yr = bn_t(xr)
        yr = yr + rr if res else yr
        if act == _lib.ACT_RELU:
            yr = torch.relu(yr)
        (yr * gy).sum().backward()
        bn = hnn.BatchNorm2d(c).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(bn_t.weight); bn.bias.copy_(bn_t.bias)
        xd = x.to(dev).requires_grad_(True); rd = r.to(dev).requires_grad_(True) if res else None
        xa = ops.as_activation(xd, dtype); ra = ops.as_activation(rd, dtype) if res else None
        win = None
        if c % 8 == 0 and rnd.random() < 0.4:
            tot = c + 8 * rnd.randint(1, 4); off = 8 * rnd.randint(0, (tot - c) // 8)
            buf = ops.empty_nhwc(n, tot, h, w, dtype, dev).fill_(5.0); win = ops.alias_window(buf, off, c)
        z = ops.bn_act(xa, bn, residual=ra, act=act, out=win)
        (z.float() * ops.to_nhwc_raw(gy.to(dev), torch.float32)).sum().backward()
        if n * h * w < 4:
            continue                     # degenerate statistics: the gradients cancel to round-off
        tol = 2e-4 if dtype == torch.float32 else 2e-2
        zz = ops.dense(z) if win is not None else z
        errs = [rel(ops.to_nchw_raw(zz.detach(), torch.float32).cpu(), yr.detach()), rel(xd.grad.cpu(), xr.grad),
                rel(bn.weight.grad.cpu(), bn_t.weight.grad), rel(bn.bias.grad.cpu(), bn_t.bias.grad),
                rel(bn.running_var.cpu(), bn_t.running_var)]
        if res:
            errs.append(rel(rd.grad.cpu(), rr.grad))
        worst = max(worst, max(errs) / tol)
        if max(errs) > tol:
            print("MISMATCH bn case %d: n=%d c=%d %dx%d %s residual=%s act=%d window=%s errs=%s" % (
                case, n, c, h, w, dtype, res, act, win is not None, errs))
            sys.exit(1)
    print("bn stress OK: %d cases, worst error / tolerance %.3f" % (n_cases, worst))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bn_cases(n_cases)
    rnd = random.Random(7)
    worst = 0.0
    for case in range(n_cases):
        if case % 2 == 0:      # 1x1
            cin = rnd.choice([8, 16, 24, 32, 40, 48, 64, 72, 80]); cout = rnd.choice([3, 8, 16, 19, 24, 32, 40, 64, 80, 96])
            k, pad = 1, 0
        else:                  # 3x3, produced or reduced channels 32 mod 64
            cin = rnd.choice([32, 64, 96, 128]); cout = rnd.choice([32, 64, 96, 160])
            k, pad = 3, 1
        n, h, wd_ = rnd.choice([1, 2, 3]), rnd.randint(1, 40), rnd.randint(1, 70)
        bias = rnd.random() < 0.5 and k == 1
        win = None
        if k == 1 and cout % 8 == 0 and rnd.random() < 0.4:
            total = cout + 8 * rnd.randint(1, 6)
            win = (8 * rnd.randint(0, (total - cout) // 8), total)
        g = torch.Generator(device="cpu").manual_seed(case)
        x = torch.randn(n, cin, h, wd_, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev) if bias else None
        gy = ops.to_nhwc_raw(torch.randn(n, cout, h, wd_, generator=g).to(dev), torch.float32)
        res = {}
        for fast in (True, False):
            ops.use_pw_kernel = fast; ops.use_tile_kernel = fast
            try:
                res[fast] = run(x, w, b, k, pad, gy, win)
            finally:
                ops.use_pw_kernel = True; ops.use_tile_kernel = True
        e = [rel(res[True][i], res[False][i]) for i in range(3)]
        ok = max(e[:2]) < 3e-3 and e[2] < 1e-4 and res[True][3] in (None, True) and all(torch.isfinite(t).all() for t in res[True][:3])
        worst = max(worst, *e)
        if not ok:
            print("MISMATCH case %d: n=%d %dx%d %d->%d k=%d bias=%s win=%s errs=%s window_intact=%s" % (case, n, h, wd_, cin, cout, k, bias, win, e, res[True][3]))
            sys.exit(1)
    torch.cuda.synchronize()
    print("stress parity OK: %d cases, worst relative difference %.2e" % (n_cases, worst))


if __name__ == "__main__":
    main()
