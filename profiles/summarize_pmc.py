#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (csv output).  Usage:
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
Kernels are keyed by short name + grid + dynamic LDS size (what every pass records for a dispatch whatever its counters: the R1 and
R2 launches of the fused decode, k_mm8f, share name and grid and differ in their LDS image -- round 4's summary averaged the two
when the passes' launch counts differed), and launches of one kernel whose written bytes / instruction counts differ (two
significant digits) are separate rows where the passes have the same number of launches (launch i of one pass is then
matched with launch i of the others).  Every counter found in the given directories is averaged over the launches of a row.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB units; gfx950 FETCH_SIZE
reports half of a wide coalesced read: MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name if len(name) < 60 else name[:57] + "..."


def sig2(v):
    """two significant digits: launches of one kernel with different shapes differ by more than that"""
    if v == 0:
        return 0
    import math
    e = int(math.floor(math.log10(abs(v)))) - 1
    return int(round(v / 10 ** e)) * 10 ** e


def main(dirs):
    # per (kernel, grid): one list of launches per pass, in dispatch order -- the benchmark is deterministic, so launch i of
    # one pass is launch i of another
    passes = defaultdict(list)
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(lambda: defaultdict(float))
            meta = {}
            for row in csv.DictReader(open(path)):
                disp = int(row["Dispatch_Id"])
                per_dispatch[disp][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[disp] = (short(row["Kernel_Name"]), int(row["Grid_Size"]), int(float(row.get("LDS_Block_Size") or 0)))
            by_kernel = defaultdict(list)
            for disp in sorted(per_dispatch):
                by_kernel[meta[disp]].append(dict(per_dispatch[disp]))
            for k, launches in by_kernel.items():
                passes[k].append(launches)
    acc = defaultdict(lambda: defaultdict(list))
    for k, plist in passes.items():
        n = len(plist[0])
        if all(len(p) == n for p in plist):
            merged = [dict() for _ in range(n)]
            for p in plist:
                for i, l in enumerate(p):
                    merged[i].update(l)
            for l in merged:
                # same kernel, different shapes (the fused R1 / R2 decodes of an open, the input generator): separate rows
                shape = tuple(sig2(l[c]) for c in ("WRITE_SIZE", "SQ_INSTS_VALU") if c in l)
                for c, v in l.items():
                    acc[k + (shape,)][c].append(v)
        else:
            for p in plist:
                for l in p:
                    for c, v in l.items():
                        acc[k + ((),)][c].append(v)
    keep = [k for k in acc if any(s in k[0] for s in ("k_mm8", "k_prescale", "k_matvec3", "k_decode_check", "k_ntt", "k_gao", "k_wb", "k_matvec2", "k_mv64"))]
    ctrs = sorted({c for k in keep for c in acc[k]})
    print(f"{'kernel':<44} {'grid':>9} {'lds':>7} {'launches':>8} " + " ".join(f"{c:>24}" for c in ctrs) + f" {'HBM bytes/launch':>18}")
    for k in sorted(keep, key=lambda k: (k[0], k[1], k[2], -sum(acc[k].get("WRITE_SIZE", [0])) / max(1, len(acc[k].get("WRITE_SIZE", [0]))))):
        vals = {c: sum(v) / len(v) for c, v in acc[k].items()}
        n = max(len(v) for v in acc[k].values())
        hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals else None
        print(f"{k[0]:<44} {k[1]:>9} {k[2]:>7} {n:>8} " + " ".join(f"{vals.get(c, float('nan')):>24.1f}" for c in ctrs)
              + (f" {hbm / 1e6:>15.2f} MB" if hbm else ""))


if __name__ == "__main__":
    main(sys.argv[1:])
