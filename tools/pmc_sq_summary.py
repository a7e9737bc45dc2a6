"""Matrix-pipe and LDS figures per kernel from the counter passes of tools/pmc_sq.sh.

python tools/pmc_sq_summary.py gpurun_out/pmc_sq_wide gpurun_out/pmc_sq_wgrad ...   (prefixes; <prefix>_a holds the SQ pass, <prefix>_b the
GRBM pass).  Per kernel, averages over its launches:
  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
  LDS busy         = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256 CUs);  conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  clock            = GRBM_GUI_ACTIVE / 8 / duration
"""
import collections
import csv
import re
import sys

KEEP = ("conv3x3_wide_kernel", "conv3x3_tile_kernel", "wgrad3x3_strip")


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(s in k for s in KEEP):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg, dur


def main():
    print("%-44s %8s %9s %9s %9s %9s %9s %8s" % ("kernel (128 -> 128 channels, 6 x 256 x 512)", "launches", "us", "GHz", "MFMA busy",
                                               "LDS busy", "conflicts", "MFMAs"))
    for prefix in sys.argv[1:]:
        a, _ = load(prefix + "_a/a_counter_collection.csv")
        b, dur = load(prefix + "_b/b_counter_collection.csv")
        for k in a:
            if k not in b:
                continue
            c = {n: sum(v) / len(v) for n, v in a[k].items()}
            gui = sum(b[k]["GRBM_GUI_ACTIVE"]) / len(b[k]["GRBM_GUI_ACTIVE"]) / 8.0
            d = sum(dur[k]) / len(dur[k])
            short = re.sub(r"_ZN12_GLOBAL__N_1\d+|\(anonymous namespace\)::|void |\(.*$", "", k)
            print("%-44s %8d %9.1f %9.2f %8.1f%% %8.1f%% %8.1f%% %8.2fM" % (
                short[:44], len(dur[k]), d / 1e3, gui / d, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024),
                100 * c["SQ_LDS_IDX_ACTIVE"] / (gui * 256), 100 * c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
                c["SQ_INSTS_MFMA"] / 1e6))


if __name__ == "__main__":
    main()
