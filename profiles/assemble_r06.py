#!/usr/bin/env python3
"""profiles/r06_* from the outputs of collect_r06.sh: copies the tables and writes their header lines from the data itself.
usage: python profiles/assemble_r06.py gpurun_out/<tag>"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from assemble_r02 import clean, last_json, put, rows  # noqa: E402
from assemble_r04 import pick, read  # noqa: E402


def main(src):
    tag = os.path.basename(os.path.normpath(src))
    j = last_json(os.path.join(src, "bench_default.json"))
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(HERE, "r06_bench_default_run.json"))
    t3 = rows(os.path.join(src, "kernel_stats_cfg3.txt"))
    enc = pick(t3, "hb::k_mm8<3, false, false")
    r2, r1 = pick(t3, "hb::k_mm8f<3>", "max"), pick(t3, "hb::k_mm8f<3>", "min")
    d = j["detail"]
    put("r06_bench_cfg3_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0   (MI355X, round 6, final state, collection {tag}; summarised per kernel and launch geometry by profiles/summarize_rocpd.py)",
        f"# One open = hb::k_mm8<3,false,false,true> (R1 encode, {enc:.1f} us) + hb::k_mm8f<3> with the smaller LDS size (R1 decode + validate, {r1:.1f} us) + hb::k_mm8f<3> with the larger one",
        f"#          (R2 decode + validate, 43 x 22, {r2:.1f} us) in a {j['ms_per_step'] * 1e3:.0f} us step ({j['value'] / 1e9:.2f} G shares/s on this box).  The rows mix the launches of every leg of the line (plan opens, the",
        f"# first-sight legs, two streams): the plan open alone, launch by launch, is in r06_first_sight_and_decoder.txt (open3).  detail: three_full_encodes {d.get('shares_per_s_per_gpu_three_full_encodes', 0) / 1e9:.2f} G,",
        f"# first-sight protocol path {(d.get('shares_per_s_per_gpu_first_sight_protocol_path') or 0) / 1e9:.2f} G shares/s, with R2's columns early {(d.get('shares_per_s_per_gpu_first_sight_r2_columns_early') or 0) / 1e9:.2f} G (round 5: 4.74-4.80 G on the builder's boxes, 4.36 on the driver's)."],
        read(src, "kernel_stats_cfg3.txt"))
    name = "traffic_cfg3.json"
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 10:
        shutil.copy(os.path.join(src, name), os.path.join(HERE, name))
    put("r06_pmc_cfg3.txt", [
        "# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0 --no-two-streams-extra   (MI355X, round 6; separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters)",
        "# HBM bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 FETCH correction of /opt/skills/guides/MI355X_MICROARCH.md).  --prewarm 0 makes every pass see the same launches, so launch i of one",
        "# pass is launch i of the others and rows split on (WRITE_SIZE, SQ_INSTS_VALU): k_mm8f's R2 launches (33.5 MB written) and R1 launches (1.5 MB) are separate rows.",
        "# traffic_cfg3.json (what bench.py copies into roofline.traffic / roofline.second) is the R2 row that writes 32 C d bytes, asserted by make_traffic.py.  Round 5's rows of the same kernels: r05_pmc_cfg3.txt",
        "# (k_mm8f R2: SQ_INSTS_VALU 15.45 M, SQ_WAIT_ANY 31.8 M; k_mm8 encode: 12.8 M)."],
        read(src, "pmc_summary_cfg3.txt"))
    if os.path.exists(os.path.join(src, "kernel_stats_cfg3-p64.txt")):
        jp = last_json(os.path.join(src, "bench_cfg3-p64.json"))
        segs = jp["roofline"]["segments"]
        put("r06_bench_cfg3-p64_kernel_stats.txt", [
            f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg3-p64 --cpu-sample 0   (MI355X, round 6, collection {tag}): config 3's open over p = 2^64 - 59, 8-byte elements,",
            f"# three launches of hb::k_mv64m<3> (hb_narrow.hip: int8 matrix cores).  The un-profiled line of the same collection: {jp['value'] / 1e9:.2f} G shares/s, {jp['ms_per_step'] * 1e3:.1f} us an open "
            f"(encode {segs[0]['ms'] * 1e3:.1f}, R1 {segs[1]['ms'] * 1e3:.1f}, R2 {segs[2]['ms'] * 1e3:.1f} us).",
            "# Round 5, the integer-VALU kernel k_mv64: r05_bench_cfg3-p64_kernel_stats.txt (14.2 G shares/s as this round's bench.py measures it)."],
            read(src, "kernel_stats_cfg3-p64.txt"))
        put("r06_pmc_cfg3-p64.txt", ["# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ counters (separate passes) -- python bench.py --workload cfg3-p64 --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0   (MI355X, round 6)"],
            read(src, "pmc_summary_cfg3-p64.txt"))
        name = "traffic_cfg3-p64.json"
        if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 10:
            shutil.copy(os.path.join(src, name), os.path.join(HERE, name))
    other = ""
    for w in ("cfg5-shard", "cfg3-omega", "cfg2", "cfg5", "cfg4", "cfg4_erasures10", "cfg3-p64"):
        p = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(p) and os.path.getsize(p) > 10:
            other += open(p).read().strip().splitlines()[-1] + "\n"
    with open(os.path.join(HERE, "r06_bench_other_workloads.json"), "w") as f:
        f.write(other)
    parts = [("open3.txt", "scratch/time_open3.py 100 (config 3's plan open: each launch between two HIP events, median of 100; `open` = the three back to back)"),
             ("first_sight_flows.txt", "scratch/first_sight_timeline.py <flow>, twice each (config 3, points 1 .. n): wait = rounds 4-5 (every quorum's add() waits for its verdict); defer --r1-in-order = verdicts "
                                       "deferred, R2's decoder made while R1's launch runs, every build in order on the caller's stream; defer = R1's first half beside the encode on the context's side stream "
                                       "(what bench.py's value_first_sight_protocol_path runs); early = R2's columns announced while R1's launch runs"),
             ("first_sight_timeline.txt", "the same under rocprofv3 --kernel-trace: kernels of one first-sight open in START order, idle gap before each (negative: it started before the previous one ended)"),
             ("first_sight_host.txt", "scratch/first_sight_host.py: when, after the open's start, each step of the host loop returns (flow `defer`; averages of 200 opens, us)"),
             ("open_p64.txt", "scratch/time_open_p64.py 100 (the same over the 64-bit prime: k_mv64m, then with HB_NO_MFMA=1 the integer-VALU kernel k_mv64)"),
             ("dec21_cfg3.txt", "scratch/dec21.py (the 21-liar open at config 3's shape alone: wall clock of the sixth, add() times)"),
             ("dec21_cfg5.txt", "scratch/dec21.py 256 85 (the 85-liar open at config 5's shard shape)"),
             ("dec21_cfg3_spread.txt", "scratch/dec21.py 64 21 spread freeze (the 21 liars one after every two honest senders: no candidate stands, the probe decides)"),
             ("dec21_cfg5_spread.txt", "scratch/dec21.py 256 85 spread freeze (the same at config 5's shard shape)")]
    body = []
    for name, what in parts:
        p = os.path.join(src, name)
        if os.path.exists(p):
            body.append(f"## {name}: {what}\n" + clean(p))
    put("r06_first_sight_and_decoder.txt", [f"# profiles/collect_r06.sh {tag}, one MI355X box, round 6 final state"], "".join(body))
    body = []
    for name, what in (("stress_decoder.txt", "scratch/stress_decoder.py 100 61 (the device decoder, its optimistic phase in C, against the host mirror after every column)"),
                       ("stress_gao.txt", "scratch/stress_gao.py 80 62 (hb_gao_decode / hb_wb_decode against the oracle: structured messages, coordinated liars, per-word and shared erasure patterns)"),
                       ("stress_open_paths.txt", "scratch/stress_open_paths.py 60 63"),
                       ("stress_narrow.txt", "scratch/stress_narrow.py 60 64 (opens over word-size primes: k_mv64m against the generic one-limb kernels and exact integers)")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            body.append(f"## {what}\n" + "\n".join(clean(p).splitlines()[-3:]) + "\n")
    put("r06_stress_runs.txt", [f"# bounded randomised differential runs of collection {tag}"], "".join(body))


if __name__ == "__main__":
    main(sys.argv[1])
