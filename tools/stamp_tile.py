"""Print the in-kernel cycle stamps of tools/ablate/tile_stamp.so (an instrumented build of the tile kernel)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uresnet_amd  # noqa
from uresnet_amd._lib import ConvDesc, UR_BF16

for (n, h, w, cin, cout) in [(6, 256, 512, 128, 128), (6, 64, 128, 512, 512), (6, 512, 1024, 64, 64)]:
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(n, h, w, cin, device=dev).relu_().to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16)
    d = ConvDesc(UR_BF16, n, h, w, cin, h, w, cout, 3, 3, 1, 1, cin, cout)
    lib = C.CDLL(os.path.abspath(sys.argv[1]))
    lib.ur_conv3x3_tile_packed_elems.restype = C.c_size_t
    wp = torch.empty(lib.ur_conv3x3_tile_packed_elems(C.byref(d)), device=dev, dtype=torch.bfloat16)
    lib.ur_conv3x3_tile_pack_weight(C.c_void_p(wt.data_ptr()), C.c_void_p(wp.data_ptr()), C.byref(d), 0, st)
    rows = lib.ur_conv3x3_tile_stats_rows(C.byref(d))
    stats = torch.zeros(rows, cout, 2, device=dev)
    for _ in range(3):
        assert lib.ur_conv3x3_tile_fwd(C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(y.data_ptr()),
                                       C.c_void_p(stats.data_ptr()), C.byref(d), st) == 0
    torch.cuda.synchronize()
    s = stats.view(rows, -1)[:, :7].double().cpu()
    m = s.mean(0)
    ch = m[6].item()
    print("shape", (n, h, w, cin, cout), "slices %d" % ch)
    for name, v in zip(["dma + load issue", "compute", "commit+load issue", "final wait", "barrier"], m[:5].tolist()):
        print("   %-18s %8.0f ticks/slice  (%4.1f %%)" % (name, v / ch, 100 * v / m[5].item()))
    print("   loop total %.0f ticks; per slice %.0f" % (m[5].item(), m[5].item() / ch))
