#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (csv output).  Usage:
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
Kernels are keyed by short name + grid; every counter found in the given directories is averaged over the
launches of that kernel.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB units; gfx950 FETCH_SIZE
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


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)
            meta = {}
            for row in csv.DictReader(open(path)):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                meta[row["Dispatch_Id"]] = (short(row["Kernel_Name"]), int(row["Grid_Size"]))
            for (disp, ctr), v in per_dispatch.items():
                acc[meta[disp]][ctr].append(v)
    keep = [k for k in acc if any(s in k[0] for s in ("k_mm8", "k_prescale", "k_matvec3", "k_decode_check", "k_ntt", "k_gao", "k_wb", "k_matvec2"))]
    ctrs = sorted({c for k in keep for c in acc[k]})
    print(f"{'kernel':<44} {'grid':>9} {'launches':>8} " + " ".join(f"{c:>24}" for c in ctrs) + f" {'HBM bytes/launch':>18}")
    for k in sorted(keep, key=lambda k: -sum(acc[k].get("WRITE_SIZE", [0]))):
        vals = {c: sum(v) / len(v) for c, v in acc[k].items()}
        n = max(len(v) for v in acc[k].values())
        hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals else None
        print(f"{k[0]:<44} {k[1]:>9} {n:>8} " + " ".join(f"{vals.get(c, float('nan')):>24.1f}" for c in ctrs)
              + (f" {hbm / 1e6:>15.2f} MB" if hbm else ""))


if __name__ == "__main__":
    main(sys.argv[1:])
