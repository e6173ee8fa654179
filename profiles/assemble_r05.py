#!/usr/bin/env python3
"""profiles/r05_* from the outputs of collect_r05.sh: copies the tables and writes their header lines from the data itself.
usage: python profiles/assemble_r05.py gpurun_out/<tag>"""
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
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(HERE, "r05_bench_default_run.json"))
    t3 = rows(os.path.join(src, "kernel_stats_cfg3.txt"))
    enc = pick(t3, "hb::k_mm8<3, false, false")
    r2, r1 = pick(t3, "hb::k_mm8f<3>", "max"), pick(t3, "hb::k_mm8f<3>", "min")
    d = j["detail"]
    put("r05_bench_cfg3_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0   (MI355X, round 5, final state, collection {tag}; summarised per kernel and launch geometry by profiles/summarize_rocpd.py)",
        f"# One open = hb::k_mm8<3,false,false,true> (R1 encode, {enc:.1f} us) + hb::k_mm8f<3> with the smaller LDS size (R1 decode + validate, {r1:.1f} us) + hb::k_mm8f<3> with the larger one",
        f"#          (R2 decode + validate, 43 x 22, {r2:.1f} us) = {enc + r1 + r2:.1f} us of kernels in a {j['ms_per_step'] * 1e3:.0f} us step ({j['value'] / 1e9:.2f} G shares/s on this box; these kernels were not changed in round 5).",
        f"# detail: three_full_encodes {d.get('shares_per_s_per_gpu_three_full_encodes', 0) / 1e9:.2f} G, first-sight protocol path {(d.get('shares_per_s_per_gpu_first_sight_protocol_path') or 0) / 1e9:.2f} G shares/s (round 4: 3.99-4.36 G; the decoder's",
        "# optimistic phase is an hb_dec object behind the C ABI now).  hb::k_fs_build_z_cand: the first half of a first-sight decode at small-integer points (one launch of two workgroups)."],
        read(src, "kernel_stats_cfg3.txt"))
    for name in ("traffic_cfg3.json", "traffic_cfg4.json", "traffic_cfg3-p64.json"):
        if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 10:
            shutil.copy(os.path.join(src, name), os.path.join(HERE, name))
    put("r05_pmc_cfg3.txt", [
        "# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0 --no-two-streams-extra   (MI355X, round 5; separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters)",
        "# HBM bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 FETCH correction of /opt/skills/guides/MI355X_MICROARCH.md).  --prewarm 0 makes every pass see the same launches, so launch i of one",
        "# pass is launch i of the others and rows split on (WRITE_SIZE, SQ_INSTS_VALU): k_mm8f's R2 launches (33.5 MB written) and R1 launches (1.5 MB) are separate rows (round 4's summary averaged them).",
        "# traffic_cfg3.json (what bench.py copies into roofline.traffic / roofline.second) is the R2 row that writes 32 C d bytes, asserted by make_traffic.py."],
        read(src, "pmc_summary_cfg3.txt"))
    other = ""
    for w in ("cfg5-shard", "cfg3-omega", "cfg2", "cfg5"):
        p = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(p):
            other += open(p).read().strip().splitlines()[-1] + "\n"
    with open(os.path.join(HERE, "r05_bench_other_workloads.json"), "w") as f:
        f.write(other)
    j4 = last_json(os.path.join(src, "bench_cfg4.json"))
    j4e = last_json(os.path.join(src, "bench_cfg4_erasures10.json"))
    shutil.copy(os.path.join(src, "bench_cfg4.json"), os.path.join(HERE, "r05_bench_cfg4.json"))
    shutil.copy(os.path.join(src, "bench_cfg4_erasures10.json"), os.path.join(HERE, "r05_bench_cfg4_erasures10.json"))
    t4 = rows(os.path.join(src, "kernel_stats_cfg4.txt"))
    g, itp = pick(t4, "k_gao<9, 8>"), pick(t4, "hb::k_mm8w<false, 3, 4>", "max")
    fins = sorted(r[3] for r in t4 if r[0].startswith("k_gao_finish<9, 8>"))
    put("r05_bench_cfg4_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --steps 5 --cpu-sample 0   (MI355X, round 5, final state, collection {tag})",
        f"# One decode of 2^18 codewords (n = 100, k = 34, 33 errors each) = hb::k_mm8w<false,3,4> (the interpolant, {itp / 1e3:.2f} ms) + k_gao ({g / 1e3:.2f} ms) + k_gao_finish ({fins[0] / 1e3:.2f} ms through the Welch-Berlekamp",
        f"# entry point, {fins[-1] / 1e3:.2f} with the locators: a lane produces its codeword's factors, the wave scales a slice of 64 codewords coalesced; round 4: 0.86); the bench line: {j4['value'] / 1e6:.2f} M codewords/s",
        f"# ({j4['ms_per_step']:.1f} ms a step); with 10 symbols of every codeword erased (the same positions: Gao's kernels on the 90 surviving points, 28 errors each): {j4e['value'] / 1e6:.2f} M codewords/s",
        "# (round 4: any erasure sent the whole batch through the row reduction, 72 k codewords/s)."],
        read(src, "kernel_stats_cfg4.txt"))
    put("r05_pmc_cfg4.txt", [
        "# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ counters (separate passes) --kernel-trace -- python bench.py --workload cfg4 --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0   (MI355X, round 5)",
        "# traffic_cfg4.json: per call 1.72 GB interpolant (the k_mm8w<false,...> row that writes n symbols a codeword; the other row of that kernel is the workload's own encode of its inputs) + 1.16 GB k_gao",
        "# + 0.68 GB finisher = 3.56 GB = 13.6 KB per codeword through the Welch-Berlekamp entry point (round 4: 14.7 KB; the finisher 1.06 -> 0.68 GB)."],
        read(src, "pmc_summary_cfg4.txt"))
    jp = last_json(os.path.join(src, "bench_cfg3-p64.json"))
    shutil.copy(os.path.join(src, "bench_cfg3-p64.json"), os.path.join(HERE, "r05_bench_cfg3-p64.json"))
    seg = jp["roofline"]["segments"]
    put("r05_bench_cfg3-p64_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg3-p64 --cpu-sample 0   (MI355X, round 5, collection {tag}): config 3's open over p = 2^64 - 59, 8-byte elements",
        f"# One open = three launches of hb::k_mv64<24, .> (hb_narrow.hip): encode {seg[0]['ms'] * 1e3:.1f} us, R1 decode + validate {seg[1]['ms'] * 1e3:.1f} us, R2 {seg[2]['ms'] * 1e3:.1f} us; the bench line: {jp['value'] / 1e9:.2f} G shares/s",
        f"# (start of the round, on the generic 3-digit kernels: 6.64 G), roofline.frac {jp['roofline']['frac']:.3f} of the HBM peak for the slowest segment; CPU baseline (orc_batch_open_u64, {jp['cpu_baseline']['cores']} threads): {jp['cpu_baseline']['value'] / 1e6:.0f} M shares/s."],
        read(src, "kernel_stats_cfg3-p64.txt"))
    put("r05_pmc_cfg3-p64.txt", ["# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ counters (separate passes) -- python bench.py --workload cfg3-p64 --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0   (MI355X, round 5)"],
        read(src, "pmc_summary_cfg3-p64.txt"))
    parts = [("device_decoder.txt", "scratch/bench_device_decoder.py (columns received in place)"),
             ("decoder_cfg5_shape.txt", "scratch/decoder_cfg5_shape.py (n = 256, t = 85, one GPU's shard of config 5, omega points)"),
             ("first_sight_host_phases.txt", "scratch/time_first_sight.py (host phases of a fault-free first-sight decode, config 3's shape; each add() is timed, which costs ~0.1 us a call)"),
             ("first_halves.txt", "scratch/time_builder.py (hb_quick_dec_arrivals alone, 280 calls back to back on one stream; the n = 64 omega n_coef = 1 row is an artefact of that queueing: 22.9 us a call when each is waited for)"),
             ("first_sight_timeline.txt", "scratch/first_sight_timeline.py, plain and under rocprofv3 --kernel-trace (config 3, points 1 .. n): kernels of one first-sight open in order, idle gap before each"),
             ("first_sight_timeline_omega.txt", "the same at omega points and at config 5's shard shape"),
             ("dec21_cfg3.txt", "scratch/dec21.py (the 21-liar open at config 3's shape alone: wall clock of the sixth, add() times)"),
             ("dec21_cfg5.txt", "scratch/dec21.py 256 85 (the 85-liar open at config 5's shard shape)"),
             ("dec21_cfg3_spread.txt", "scratch/dec21.py 64 21 spread freeze (the 21 liars one after every two honest senders: no candidate stands, the probe decides)"),
             ("dec21_cfg5_spread.txt", "scratch/dec21.py 256 85 spread freeze (the same at config 5's shard shape)")]
    body = []
    for name, what in parts:
        p = os.path.join(src, name)
        if os.path.exists(p):
            body.append(f"## {name}: {what}\n" + clean(p))
    put("r05_device_decoder_and_first_sight.txt", [f"# profiles/collect_r05.sh {tag}, one MI355X box, round 5 final state"], "".join(body))
    body = []
    for name, what in (("stress_decoder.txt", "scratch/stress_decoder.py 150 51 (the device decoder, its optimistic phase in C, against the host mirror after every column)"),
                       ("stress_gao.txt", "scratch/stress_gao.py 120 52 (hb_gao_decode / hb_wb_decode against the oracle: structured messages, coordinated liars, per-word and shared erasure patterns)"),
                       ("stress_open_paths.txt", "scratch/stress_open_paths.py 90 53")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            body.append(f"## {what}\n" + "\n".join(clean(p).splitlines()[-3:]) + "\n")
    put("r05_stress_runs.txt", [f"# bounded randomised differential runs of collection {tag}"], "".join(body))


if __name__ == "__main__":
    main(sys.argv[1])
