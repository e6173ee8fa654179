#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel AND launch geometry, so the
encode / decode / validate launches of k_matvec are separated.  Usage:
    python profiles/summarize_rocpd.py gpurun_out/prof_r1/bench_results.db > profiles/r01_....txt"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name if len(name) < 90 else name[:87] + "..."


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, duration, stream_id from kernels").fetchall()
    main_stream = min(r[8] for r in rows) if rows else 0
    groups = {}
    for name, gx, wx, vg, av, sg, lds, dur, stream in rows:
        # launches on the secondary streams (bench.py's two-opens-in-flight loop) share the GPU with each other: kept apart
        tag = "" if stream == main_stream else " [2 streams]"
        groups.setdefault((short(name) + tag, gx, wx, vg, av, sg, lds), []).append(dur)
    # a kernel launched with the same geometry for different jobs (encode vs decode) is split where its
    # durations are clearly bimodal
    split = {}
    for key, durs in groups.items():
        lo, hi = min(durs), max(durs)
        if len(durs) >= 6 and hi > 1.45 * lo:
            mid = (lo + hi) / 2
            a = [d for d in durs if d < mid]
            b = [d for d in durs if d >= mid]
            if len(a) >= 2 and len(b) >= 2:
                split[(key[0] + " [short launches]",) + key[1:]] = a
                split[(key[0] + " [long launches]",) + key[1:]] = b
                continue
        split[key] = durs
    groups = split
    total = sum(sum(v) for v in groups.values())
    print(f"{'kernel':<60} {'grid':>9} {'wg':>4} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>9} {'%':>6}")
    for key, durs in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        name, gx, wx, vg, av, sg, lds = key
        n = len(durs)
        print(f"{name:<60.60} {gx:>9} {wx:>4} {vg + av:>5} {sg:>5} {lds:>6} {n:>6} {sum(durs) / n / 1e3:>10.2f} {min(durs) / 1e3:>10.2f} {max(durs) / 1e3:>10.2f} {sum(durs) / 1e6:>9.3f} {100 * sum(durs) / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
