"""HBM bandwidth reference points for the elementwise kernels: device copy, read-only sum and fill at the sizes of the
network's largest activations, next to the BatchNorm forward kernel on the same tensor."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

dev = torch.device("cuda:0")
for (n, h, w, c) in [(6, 512, 1024, 64), (6, 256, 512, 128), (6, 128, 256, 256)]:
    x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    nbytes = x.numel() * 2
    t = timed(lambda: y.copy_(x));      print("%-22s copy  %6.1f us %5.2f TB/s (read+write)" % ((n, h, w, c), t * 1e6, 2 * nbytes / t / 1e12))
    t = timed(lambda: x.float().sum()) if False else timed(lambda: torch.sum(x, dtype=torch.float32)); print("%-22s sum   %6.1f us %5.2f TB/s (read)" % ((n, h, w, c), t * 1e6, nbytes / t / 1e12))
    t = timed(lambda: y.fill_(1.0));    print("%-22s fill  %6.1f us %5.2f TB/s (write)" % ((n, h, w, c), t * 1e6, nbytes / t / 1e12))
    t = timed(lambda: torch.relu_(y));  print("%-22s relu_ %6.1f us %5.2f TB/s (read+write in place)" % ((n, h, w, c), t * 1e6, 2 * nbytes / t / 1e12))
    import uresnet_amd
    from uresnet_amd import ops
    from uresnet_amd._lib import call
    import ctypes as C
    mean = torch.zeros(c, device=dev); inv = torch.ones(c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rows = n * h * w
    f = lambda: call("ur_bn_act_fwd", x.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), None, 1, y.data_ptr(), uresnet_amd._lib.UR_BF16, rows, c, 0, st)
    t = timed(f); print("%-22s ur_bn_act_fwd %6.1f us %5.2f TB/s" % ((n, h, w, c), t * 1e6, 2 * nbytes / t / 1e12))
