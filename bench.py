#!/usr/bin/env python3
"""Headline benchmark: U-ResNet Cityscapes-shaped 512x1024 bf16 training, images/sec.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU.  Either an outer launcher provides the ranks
(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: WORLD_SIZE is set), or this
script starts them itself: with --gpus N > 1 and no WORLD_SIZE in the environment the parent process spawns
`python -m torch.distributed.run` with N ranks BEFORE importing torch or touching a GPU, relays rank 0's
JSON line and exits with the launcher's return code.

One step = one pass of the hot path over one synthetic batch: NCHW fp32 images already resident in
HBM -> UResNetNewEUCB(19, 3, 64) forward -> CE(ignore 255) -> backward -> [gradient all-reduce] ->
Adam.  Rank 0 prints ONE JSON line.  `roofline` describes the kernel family with the largest share of
the step (kernels that do one job -- e.g. the two 3x3 forward / backward-data kernels with all their
epilogue variants -- count as one family), timed live with events on the stream each launch runs on:
the top-level numbers are those of the timed region's own streams, `isolated_one_stream` repeats them
with every kernel alone on the chip, `matrix_pipe_utilisation_of_the_step` divides all convolution
FLOPs by the wall time, `against_the_lower_of_both_roofs` prices every launch of the family against
max(FLOPs / matrix-core peak, operand bytes / HBM peak).  `cpu_baseline` is the oracle (plain PyTorch fp32 restatement of the reference)
on the host cores.

A rank that hangs does not sit until the caller's limit: with more than one rank every process arms a
timer (URESNET_BENCH_DUMP_AFTER seconds, default 420) that prints every thread's stack and exits
non-zero; the launcher then ends the other ranks and this script relays the return code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Kernel arguments in device memory instead of host-coherent memory: the HIP runtime reads this switch when it is loaded
# (i.e. at `import torch`, which happens in main(), after this line; self-launched ranks inherit it).  A step is about
# 460 launches, a third of them five-microsecond kernels in dependent chains: 27.9 -> 27.7 ms per step.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (no sparsity)
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E peak (about 6300 GB/s is achievable by a streaming kernel)
FP32_MATRIX_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=6, help="images per GPU (the reference's ablation script uses 6)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--all-kernel-timing", action="store_true",
                    help="bracket every convolution and BatchNorm launch with events (adds about 1 ms per step); "
                         "the default brackets only the dominant kernel's launches")
    ap.add_argument("--per-shape", action="store_true", help="print per-(kernel, shape) conv timings to stderr")
    ap.add_argument("--per-shape-top", type=int, default=40)
    ap.add_argument("--ddp-single", action="store_true",
                    help="with --gpus 1: still wrap the model in ddp.DataParallel over a one-rank RCCL group, so that the hooks, "
                         "the gradient buckets (filled in place by the backward-weight kernels) and the all-reduce launches run; "
                         "the difference to the plain line is the per-step cost of the data-parallel machinery on one GPU")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"],
                    help="how the timed steps are launched: eagerly from Python, as one HIP graph replay per step (forward, loss, "
                         "backward, [all-reduce,] Adam, weight repack captured once), or `auto`: capture the graph, time a few "
                         "steps of both outside the timed region and use the faster (more than one rank: eager unless asked)")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launch-path rehearsal without a GPU: the ranks rendezvous over gloo and push a small stand-in "
                         "torch.nn network through ddp.DataParallel; exercises spawning, bucketed all-reduce, the "
                         "max-over-ranks clock and the JSON line.  Not a measurement (the line says so)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 without an outer launcher: start N ranks as children of this process, which has not imported torch
    and never touches a GPU (a process that has initialised the GPU must not be replaced or forked into ranks)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4" if args.rehearse_cpu else "16")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def rehearse_cpu(args):
    """Stand-in workload for --rehearse-cpu: same launch, sharding, DataParallel buckets, global valid-pixel count,
    fences and JSON line as the real run, on a two-layer torch.nn network over gloo."""
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    import uresnet_amd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    torch.manual_seed(rank)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 19, 1))
    ddp = uresnet_amd.ddp.DataParallel(net, bucket_mb=0.0005) if world > 1 else None
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    B = 2
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, 16, 32, generator=g)
    y = torch.randint(0, 19, (B, 16, 32), generator=g)
    y[torch.rand(y.shape, generator=g) < 0.1] = 255

    def step():
        out = net(x)
        nll = F.cross_entropy(out, y, ignore_index=255, reduction="sum")
        valid = (y != 255).sum().float()
        count = valid.reshape(1)
        if ddp is not None:
            count = ddp.count_hook(torch.stack([nll.detach(), valid, nll.detach() / valid]))
        loss = nll / count[0]
        opt.zero_grad()
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss.detach()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    w0 = torch.cat([p.detach().flatten() for p in net.parameters()])
    spread = torch.zeros(1)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lo, hi = w0.clone(), w0.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = (hi - lo).abs().max().reshape(1)
    if rank == 0:
        print(json.dumps({"metric": "launch-path rehearsal on CPU over gloo (NOT a measurement)", "rehearsal": True,
                          "value": round(world * B * args.steps / float(t.item()), 3), "unit": "images/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(t.item()) / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stand-in two-layer network 3x16x32, gloo", "per_gpu_batch": B,
                                     "global_batch": B * world, "parallelism": "dp%d" % world,
                                     "final_loss": round(float(loss), 5),
                                     "max_weight_spread_across_ranks": float(spread.item())},
                          "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """(HBM bytes per launch, launches, source file) of `kernel` from the newest committed counter summary (profiles/*_pmc_traffic.json, made by
    tools/pmc_bench.sh + tools/pmc_traffic.py from two rocprofv3 --pmc passes over this same default workload).
    Hardware counters cannot be read from inside the timed process, so this is the one number not measured live."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None, None
    rec = json.load(open(files[-1])).get("kernels", {}).get(kernel)
    return (rec["hbm_bytes_per_launch"], rec["launches"], os.path.relpath(files[-1], ROOT)) if rec else (None, None, None)


def cpu_baseline(height, width):
    """One training step of the oracle at batch 1, full resolution, on the host cores (rank 0 only)."""
    import torch
    from oracle import uresnet_oracle as O
    torch.manual_seed(0)
    model = O.UResNetNewEUCB(19, 3, 64).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 3, height, width, generator=g)
    y = torch.randint(0, 19, (1, height, width), generator=g)
    y[torch.rand(y.shape, generator=g) < 0.1] = 255
    # warm-up on a quarter-size crop (allocators, mkldnn primitive caches), then timed full-size steps
    O.seg_train_step(model, opt, x[:, :, :height // 2, :width // 2].contiguous(), y[:, :height // 2, :width // 2].contiguous())
    steps = 4
    t0 = time.perf_counter()
    for _ in range(steps):
        O.seg_train_step(model, opt, x, y)
    dt = (time.perf_counter() - t0) / steps
    return {"value": 1.0 / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d Adam training steps, batch 1, 3x%dx%d fp32, oracle/uresnet_oracle.py UResNetNewEUCB(19,3,64); "
                      "%.1f s per step" % (steps, height, width, dt)}


def main():
    args = parse()
    if args.graph:
        args.launch = "graph"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # a hung rank prints every thread's stack and exits non-zero: always with more than one rank, on request with one
    dump_after = os.environ.get("URESNET_BENCH_DUMP_AFTER", "420" if args.gpus > 1 else "")
    if dump_after and int(dump_after) > 0:
        import faulthandler
        faulthandler.dump_traceback_later(int(dump_after), exit=True)
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    import torch
    import torch.distributed as dist
    import uresnet_amd
    from uresnet_amd import ops, train
    from uresnet_amd.modules.U_ResNet import UResNetNewEUCB

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL prints a version banner on the process's stdout when its first communicator comes up.  The contract is ONE
    # JSON line on stdout: keep the real stdout aside for that line and point descriptor 1 at stderr meanwhile.
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif args.ddp_single:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1, device_id=dev)
    uresnet_amd._lib.load()
    if os.environ.get("URESNET_BATCH_PACK") == "0":          # A/B switches of two fusions
        ops.batch_repack = False
    if os.environ.get("URESNET_BN_SUMS") == "0":
        ops.fuse_bn_bwd_sums = False
    if os.environ.get("URESNET_PW") == "0":
        ops.use_pw_kernel = False
    if os.environ.get("URESNET_POOL_SKIP") == "0":
        uresnet_amd.modules._blocks.fuse_pool_skip = False
    if os.environ.get("URESNET_RES_GRAD") == "0":
        uresnet_amd.modules._blocks.fuse_residual_grad = False

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = UResNetNewEUCB(19, 3, 64).set_compute_dtype(dtype).to(dev).train()
    opt = train.Adam(model.parameters(), lr=1e-4)
    ddp = uresnet_amd.ddp.DataParallel(model) if (world > 1 or args.ddp_single) else None

    B, H, W = args.batch, args.height, args.width
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, H, W, generator=g).to(dev)
    y = torch.randint(0, 19, (B, H, W), generator=g)
    y[torch.rand(y.shape, generator=g) < 0.1] = 255
    y = y.to(dev)

    def step_fn(model_, opt_, x_, y_):
        if ddp is None:
            return train.seg_train_step(model_, opt_, x_, y_)
        return train.seg_train_step(model_, opt_, x_, y_, count_hook=ddp.count_hook, grad_hook=ddp.finish)

    def step():
        return step_fn(model, opt, x, y)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, _ = step()
    fence()

    FAMILIES = {          # kernels that do one job count as one family, whatever the number of kernels and epilogue variants
        "conv3x3_forward_and_backward_data": {
            "kernel": "conv3x3_tile_kernel<R, EPI> + conv3x3_wide_kernel<EPI>: the 3x3 s1 p1 forward and backward-data launches "
                      "(one family: every layer goes to one of the two kernels, each with epilogue variants)",
            "tags": ("tile_fwd", "tile_dgrad", "tile_dgrad_bn", "wide_fwd", "wide_dgrad", "wide_dgrad_bn"),
            "pmc": ("conv3x3_tile_kernel", "conv3x3_tile_kernel (BatchNorm-backward sums in the epilogue)", "conv3x3_wide_kernel",
                    "conv3x3_wide_kernel (BatchNorm-backward sums in the epilogue)")},
        "conv3x3_backward_weight": {
            "kernel": "wgrad3x3_strip_kernel: 3x3 s1 p1 backward-weight (its fp32 slab reduction, wgrad3x3_reduce_kernel, is an "
                      "HBM-bound pass of its own: counted in this family's share of the step, not in its FLOP/s)",
            "tags": ("tile_wgrad",), "time_only": ("hbm:wgrad3x3_reduce",), "pmc": ("wgrad3x3_strip_kernel",)},
        "conv3x3_backward_weight_32_channels": {"kernel": "wgrad32_kernel: 3x3 backward-weight of the layers with 32 channels on a side",
                                                "tags": ("tile_wgrad32",)},
        "generic_forward_and_backward_data": {"kernel": "conv_igemm_kernel + conv1x1_pw_kernel: stem, 1x1, ConvT, skinny layers",
                                              "tags": ("igemm_fwd", "igemm_dgrad", "pw_fwd", "pw_dgrad")},
        "generic_backward_weight": {"kernel": "conv_wgrad_kernel (+ slab reduce) and pw_wgrad_kernel", "tags": ("wgrad", "pw_wgrad")},
    }
    CONV_TAGS = {t for f in FAMILIES.values() for t in f["tags"]}
    TIMED_TAGS = set(FAMILIES["conv3x3_forward_and_backward_data"]["tags"]) | set(FAMILIES["conv3x3_backward_weight"]["tags"])

    def timed_steps(fn, tags, n=None):
        n = args.steps if n is None else n
        if tags is not None and not args.no_kernel_timing:
            ops.kernel_timings = []
            ops.kernel_timing_tags = None if (args.all_kernel_timing and tags is CONV_TAGS) else tags
        t0 = time.perf_counter()
        for _ in range(n):
            loss, _ = fn()
        host = time.perf_counter() - t0              # launches enqueued; the GPU is still working
        fence()
        total = time.perf_counter() - t0
        tm, ops.kernel_timings = ops.kernel_timings, None
        return loss, host, total, tm

    # How a step is launched: eagerly (about 460 launches from Python) or as one HIP graph replay.  Which is faster depends on
    # the box: one stream of graph nodes replays with fewer gaps than eager launches, but a graph with two branches is
    # enqueued node by node by the runtime and its branches overlap less.  `auto` captures the graph, times a few steps
    # of each outside the timed region and takes the faster one; with more than one rank the default is eager.
    launch = args.launch
    if launch == "auto" and ddp is not None:
        launch = "eager"
    graphed, calibration = None, None
    if launch in ("auto", "graph"):
        graphed = train.GraphedStep(step_fn, model, opt, x, y, warmup=1)
        fence()
    if launch == "auto":
        n_cal = max(3, min(5, args.steps))
        for fn in (step, lambda: graphed(x, y)):          # one untimed step of each after the capture
            fn()
        fence()
        _, _, t_eager, _ = timed_steps(step, None, n_cal)
        _, _, t_graph, _ = timed_steps(lambda: graphed(x, y), None, n_cal)
        calibration = {"steps_each": n_cal, "eager_ms_per_step": round(t_eager / n_cal * 1e3, 3),
                       "graph_replay_ms_per_step": round(t_graph / n_cal * 1e3, 3)}
        launch = "graph" if t_graph < t_eager else "eager"
    two_streams = bool(ops.branch_streams)

    # ---- the timed region: exactly K steps between two fences -------------------------------------------------------------
    # Eager: the launches of the two 3x3 families carry events on their own stream inside the timed region (about 190 events,
    # 0.1 ms per step).  Graph replay: events cannot be recorded inside a graph; the same K steps are run eagerly, on the same
    # streams, right after the timed region for the per-kernel numbers.
    if launch == "eager":
        loss, host_elapsed, elapsed, region_timings = timed_steps(step, TIMED_TAGS)
    else:
        loss, host_elapsed, elapsed, _ = timed_steps(lambda: graphed(x, y), None)
        _, _, _, region_timings = timed_steps(step, TIMED_TAGS)
    final_loss = float(loss.item())
    # With the two branches on two streams a kernel shares the chip with the other stream's kernels while it runs, so its
    # event-to-event time (what a rocprofv3 trace of this command shows as well) is not the kernel's own.  The same K steps
    # run once more on ONE stream give every kernel group alone on the chip.
    isolated_timings, isolated_elapsed = None, None
    if two_streams and region_timings is not None:
        ops.branch_streams = False
        _, _, isolated_elapsed, isolated_timings = timed_steps(step, CONV_TAGS)
        ops.branch_streams = True

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    spread = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # replicas must stay identical: largest difference of any weight between any two ranks (outside the timed region)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).abs().max().item())
    elapsed = float(t.item())

    if rank == 0:
        peak = BF16_DENSE_PEAK_TFLOPS if dtype == torch.bfloat16 else FP32_MATRIX_PEAK_TFLOPS

        def fold(tm):
            kernels, shapes = {}, {}
            for tag, flops, e0, e1, shape in (tm or []):
                sec = e0.elapsed_time(e1) * 1e-3
                for table, key in ((shapes, (tag, shape)), (kernels, tag)):
                    k = table.setdefault(key, [0, 0.0, 0.0])
                    k[0] += 1
                    k[1] += flops
                    k[2] += sec
            return kernels, shapes

        def family_stats(kernels, fam, wall):
            own = [kernels[t] for t in fam["tags"] if t in kernels]
            if not own:
                return None
            n, fl, sec = sum(k[0] for k in own), sum(k[1] for k in own), sum(k[2] for k in own)
            extra = sum(kernels[t][2] for t in fam.get("time_only", ()) if t in kernels)
            return {"achieved": round(fl / sec / 1e12, 2), "frac": round(fl / sec / 1e12 / peak, 4), "launches": n,
                    "avg_launch_ms": round(sec / n * 1e3, 4), "flops_per_launch": fl / n,
                    "time_share_of_step": round((sec + extra) / wall, 3)}

        region_kernels, _ = fold(region_timings)
        kernels, shapes = fold(isolated_timings if isolated_timings is not None else region_timings)
        iso_wall = isolated_elapsed if isolated_timings is not None else elapsed
        roof = None
        if kernels:
            iso = {name: family_stats(kernels, fam, iso_wall) for name, fam in FAMILIES.items()}
            iso = {k: v for k, v in iso.items() if v}
            dominant = max(iso, key=lambda k: iso[k]["time_share_of_step"])
            fam = FAMILIES[dominant]
            top = family_stats(region_kernels, fam, elapsed) or iso[dominant]
            # HBM bytes per launch of the family's kernels from the newest committed counter passes (launch-weighted)
            traffic = source = None
            if (B, H, W, args.dtype, world) == (6, 512, 1024, "bf16", 1) and fam.get("pmc"):
                got = [pmc_traffic(k) for k in fam["pmc"]]
                got = [(g[0], g[1], g[2]) for g in got if g[0] is not None]
                if got:
                    traffic = sum(b * n for b, n, _ in got) / sum(n for _, n, _ in got)
                    source = got[0][2]
            # operands of the family's launches: activations in and out once, the weight image once
            esz = 2 if dtype == torch.bfloat16 else 4
            fam_bytes = [(n_ * h_ * w_ * (ci + co) + ci * co * k_ * k_) * esz
                         for (tag, shp), (cnt, _, _) in shapes.items() if tag in fam["tags"] and shp
                         for (n_, h_, w_, ci, co, k_, _s) in [shp] for _ in range(cnt)]
            where = ("events inside the timed region (%s)" % ("two streams: a kernel shares the chip with the other stream's "
                     "kernels while it runs, so these are what a rocprofv3 trace of this command shows, not the kernel alone: see "
                     "isolated_one_stream" if two_streams else "one stream")) if launch == "eager" else \
                    ("events over the same %d steps launched eagerly on the same streams right after the timed graph replays" % args.steps)
            roof = {"bound": "mfma", "kernel": fam["kernel"], "family": dominant, "achieved": top["achieved"], "peak": peak,
                    "unit": "TFLOP/s", "frac": top["frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                    "traffic_source": source,
                    "algorithmic_bytes_per_launch": (sum(fam_bytes) / len(fam_bytes)) if fam_bytes else None,
                    "launches": top["launches"], "avg_launch_ms": top["avg_launch_ms"], "flops_per_launch": top["flops_per_launch"],
                    "time_share_of_step": top["time_share_of_step"], "timing_region": where,
                    "dominant_by": "largest share of the step among the kernel families (one-stream pass)"}
            if isolated_timings is not None:
                roof["isolated_one_stream"] = dict(iso[dominant], note="the same %d steps run on ONE stream right after the timed "
                                                   "region (%.3f ms per step): each kernel alone on the chip"
                                                   % (args.steps, isolated_elapsed / args.steps * 1e3))
            conv_flops = sum(k[1] for t_, k in kernels.items() if not t_.startswith("hbm:"))
            roof["matrix_pipe_utilisation_of_the_step"] = {
                "achieved": round(conv_flops / elapsed / 1e12, 2),
                "frac": round(conv_flops / elapsed / 1e12 / peak, 4), "unit": "TFLOP/s",
                "note": "FLOPs of every convolution launch of the K steps divided by the wall time of the timed region"}
            roof["by_tag"] = {t_: {"launches": kernels[t_][0], "achieved": round(kernels[t_][1] / kernels[t_][2] / 1e12, 2),
                                   "frac": round(kernels[t_][1] / kernels[t_][2] / 1e12 / peak, 4),
                                   "avg_launch_ms": round(kernels[t_][2] / kernels[t_][0] * 1e3, 4)}
                              for t_ in fam["tags"] if t_ in kernels}
            # The family against the LOWER of its two roofs, launch by launch: a 3x3 layer with 64 channels moves a byte per
            # 185-275 FLOP, so at 8 TB/s its memory roof lies below the matrix-core peak.  roof time of a launch =
            # max(FLOPs / peak, operand bytes / HBM peak), operands = activations in and out once, the weight image once and,
            # for the launches that also sum a BatchNorm's backward terms, that BatchNorm's input once.
            roof_sec = meas_sec = hbm_side_sec = 0.0
            for (tag, shp), (cnt, fl, sec) in shapes.items():
                if tag in fam["tags"] and shp:
                    n_, h_, w_, ci, co, k_, _s = shp
                    produced = ci if "dgrad" in tag else co
                    by = (n_ * h_ * w_ * (ci + co + (produced if tag.endswith("_bn") else 0)) + ci * co * k_ * k_) * esz
                    t_m, t_h = fl / cnt / (peak * 1e12), by / (HBM_PEAK_GBS * 1e9)
                    roof_sec += cnt * max(t_m, t_h)
                    meas_sec += sec
                    hbm_side_sec += sec if t_h > t_m else 0.0
            if meas_sec > 0:
                roof["against_the_lower_of_both_roofs"] = {
                    "frac": round(roof_sec / meas_sec, 4),
                    "time_share_of_launches_whose_memory_roof_is_the_lower": round(hbm_side_sec / meas_sec, 3),
                    "note": "sum over the family's launches of max(FLOPs / %.0f TFLOP/s, operand bytes / %.0f GB/s) divided by their "
                            "measured time (%s)" % (peak, HBM_PEAK_GBS, "one-stream pass" if isolated_timings is not None else "timed region")}
            roof["other_families"] = {FAMILIES[k]["kernel"]: v for k, v in iso.items() if k != dominant}
            if region_kernels and isolated_timings is not None:
                roof["other_families_in_timed_region"] = {
                    FAMILIES[k]["kernel"]: family_stats(region_kernels, FAMILIES[k], elapsed)
                    for k in FAMILIES if k != dominant and family_stats(region_kernels, FAMILIES[k], elapsed)}
            hbm = {}
            for tag, (n, by, sec) in kernels.items():
                if tag.startswith("hbm:") and n:
                    hbm[tag[4:]] = {"bound": "hbm", "achieved": round(by / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(by / sec / 1e9 / HBM_PEAK_GBS, 4), "launches": n,
                                    "avg_launch_ms": round(sec / n * 1e3, 4), "bytes_per_launch": by / n,
                                    "time_share_of_step": round(sec / iso_wall, 3)}
            if hbm:
                roof["hbm_bound_kernels"] = hbm
            if args.all_kernel_timing:
                conv_secs = sum(k[2] for t_, k in kernels.items() if not t_.startswith("hbm:"))
                roof["all_conv_mfma_util"] = round(conv_flops / conv_secs / 1e12 / peak, 4)
                roof["all_conv_time_share_of_step"] = round(conv_secs / iso_wall, 3)
            else:
                roof["other_kernels_note"] = "run with --all-kernel-timing for the BatchNorm, pooling and loss passes"
        if args.per_shape:
            for (tag, shape), (n, fl, sec) in sorted(shapes.items(), key=lambda kv: -kv[1][2])[:args.per_shape_top]:
                print("%-12s %-34s calls %4d  %8.3f ms/step  %7.1f %s" % (tag, shape, n, sec / args.steps * 1e3, fl / sec / 1e12,
                                                                          "TB/s" if tag.startswith("hbm:") else "TF"), file=sys.stderr)
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # the host baseline is reported on the single-GPU line only
            cpu = cpu_baseline(H, W)
        out = {
            "metric": "train images/sec (node) Cityscapes 512x1024 bf16 @1/2/4/8 GPU; conv MFMA util %",
            "value": round(world * B * args.steps / elapsed, 3),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[2]: U-ResNet Cityscapes %dx%d segmentation (19 classes), "
                                   "UResNetNewEUCB(19,3,64) training step: forward + CE(ignore 255) + backward + Adam(1e-4)"
                                   % (H, W),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "bn": "per-replica batch statistics", "final_loss": round(final_loss, 5),
                       "host_enqueue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
                       "data_parallel_wrapper": None if ddp is None else
                       {"buckets": len(ddp.buckets), "bucket_order": ddp.bucket_order_source,
                        "max_weight_difference_between_ranks": spread,
                        "gradient_fraction_written_in_place_by_the_kernels":
                            round(ddp.in_place_elems / max(1, ddp.in_place_elems + ddp.copied_elems), 4)},
                       "branch_streams": ({"second_stream_probe": list(ops.stream_probe_log)} if ops.branch_streams else None),
                       "hip_force_dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG"), "step_launch": "one HIP graph replay per step" if launch == "graph" else "eager (about 460 launches per step)",
                       "step_launch_requested": args.launch, "step_launch_calibration": calibration},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
