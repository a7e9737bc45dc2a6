"""Per-kernel HBM traffic of the bench workload from the two counter passes of tools/pmc_bench.sh.

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads as 64 bytes, so the read side is doubled.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv out.json
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    # the tile kernel's epilogue variants are different kernels for this purpose: <R, 0|1|2> store the convolution result,
    # <R, 3..6> also read a BatchNorm's input and emit its backward sums
    m = re.search(r"conv3x3_tile_kernel<\d+, (\d+)>", name)
    if m:
        return "conv3x3_tile_kernel" if int(m.group(1)) <= 2 else "conv3x3_tile_kernel (BatchNorm-backward sums in the epilogue)"
    m = re.search(r"conv3x3_wide_kernel<(\d+)>", name)
    if m:
        return "conv3x3_wide_kernel" if int(m.group(1)) <= 2 else "conv3x3_wide_kernel (BatchNorm-backward sums in the epilogue)"
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def collect(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, (0, 0.0))
        nw, w = write.get(k, (0, 0.0))
        n = max(nf, nw)
        out[k] = {"launches": n, "read_bytes_per_launch": 2 * f * 1024 / max(nf, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1)}
        out[k]["hbm_bytes_per_launch"] = out[k]["read_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump({"command": "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing [--launch eager]: per-launch averages over every launch of the run",
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:16]:
        print("%-34s %5d launches  %8.1f MB read %8.1f MB written per launch  %5.1f %% of all traffic" % (
            k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6,
            100 * v["hbm_bytes_per_launch"] * v["launches"] / tot))
    print("total GB over the run: %.1f" % (tot / 1e9))


if __name__ == "__main__":
    main()
