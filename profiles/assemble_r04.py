#!/usr/bin/env python3
"""profiles/r04_* from the outputs of collect_r04.sh: copies the tables and writes their header lines from the data itself.
usage: python profiles/assemble_r04.py gpurun_out/<tag>"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from assemble_r02 import last_json, put, rows  # noqa: E402


def read(src, name):
    p = os.path.join(src, name)
    return open(p).read() if os.path.exists(p) else f"(missing: {name})\n"


def pick(table, prefix, lds=None):
    """average duration of the most-launched one-stream row whose kernel name starts with `prefix` (lds: 'min' / 'max' of the LDS sizes)"""
    c = [r for r in table if r[0].startswith(prefix) and "streams" not in r[0]]
    if c and lds:
        want = (min if lds == "min" else max)(x[1] for x in c)
        c = [r for r in c if r[1] == want]
    return max(c, key=lambda r: r[2])[3] if c else float("nan")


def main(src):
    tag = os.path.basename(os.path.normpath(src))
    j = last_json(os.path.join(src, "bench_default.json"))
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(HERE, "r04_bench_default_run.json"))
    t3 = rows(os.path.join(src, "kernel_stats_cfg3.txt"))
    enc = pick(t3, "hb::k_mm8<3, false, false")
    r2, r1 = pick(t3, "hb::k_mm8f<3>", "max"), pick(t3, "hb::k_mm8f<3>", "min")
    d = j["detail"]
    put("r04_bench_cfg3_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0   (MI355X, round 4, final state, collection {tag}; summarised per kernel and launch geometry by profiles/summarize_rocpd.py)",
        f"# One open = hb::k_mm8<3,false,false,true> (R1 encode, {enc:.1f} us) + hb::k_mm8f<3> with the smaller LDS size (R1 decode + validate: [N row 0 ; P](y ./ den), {r1:.1f} us)",
        f"#          + hb::k_mm8f<3> with the larger one (R2 decode + validate: [N ; P](y ./ den), 43 x 22, {r2:.1f} us) = {enc + r1 + r2:.1f} us of kernels in a {j['ms_per_step'] * 1e3:.0f} us step ({j['value'] / 1e9:.2f} G shares/s on this box).",
        f"# detail: three_full_encodes {d.get('shares_per_s_per_gpu_three_full_encodes', 0) / 1e9:.2f} G, first-sight protocol path {(d.get('shares_per_s_per_gpu_first_sight_protocol_path') or 0) / 1e9:.2f} G shares/s.",
        "# k_fs_build / k_fs_cand: the device-side build of the small-integer images (plan creation and the first-sight decoders); hb::k_mm8w rows: the full-size kernel, checked beside the default;",
        "# k_decode_check / k_matvec3: the integer-VALU family (bench's secondary figure); [2 streams] rows: the two-opens-in-flight figure."],
        read(src, "kernel_stats_cfg3.txt"))
    for name in ("traffic_cfg3.json", "traffic_cfg4.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(HERE, name))
    put("r04_pmc_cfg3.txt", [
        "# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra   (MI355X, round 4; separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters)",
        "# HBM bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 FETCH correction of /opt/skills/guides/MI355X_MICROARCH.md); traffic_cfg3.json is what bench.py copies into roofline.traffic"],
        read(src, "pmc_summary_cfg3.txt"))
    other = ""
    for w in ("cfg5-shard", "cfg3-omega", "cfg2", "cfg5"):
        p = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(p):
            other += open(p).read().strip().splitlines()[-1] + "\n"
    with open(os.path.join(HERE, "r04_bench_other_workloads.json"), "w") as f:
        f.write(other)
    j4 = last_json(os.path.join(src, "bench_cfg4.json"))
    shutil.copy(os.path.join(src, "bench_cfg4.json"), os.path.join(HERE, "r04_bench_cfg4.json"))
    t4 = rows(os.path.join(src, "kernel_stats_cfg4.txt"))
    g, fin, itp = pick(t4, "k_gao<9, 8>"), pick(t4, "k_gao_finish<9, 8>"), pick(t4, "hb::k_mm8w<false, 3, 4>", "max")
    put("r04_bench_cfg4_kernel_stats.txt", [
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --steps 5 --cpu-sample 0   (MI355X, round 4, final state, collection {tag})",
        f"# One decode of 2^18 codewords (n = 100, k = 34, 33 errors each) = hb::k_mm8w<false,3,4> (the interpolant g1 = V^-1 y, {itp / 1e3:.2f} ms) + k_gao ({g / 1e3:.2f} ms: Euclid with the generic step fused,",
        f"# the next step's scalars in idle lanes; the fraction-free division) + k_gao_finish ({fin / 1e3:.2f} ms: one inversion per four codewords) = {(itp + g + fin) / 1e3:.1f} ms; the bench line: {j4['value'] / 1e6:.2f} M codewords/s ({j4['ms_per_step']:.1f} ms a step).",
        "# Start of round 4: k_gao 31.3 ms, finisher 1.18 ms, 7.3 M codewords/s."],
        read(src, "kernel_stats_cfg4.txt"))
    put("r04_pmc_cfg4.txt", [
        "# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-sample 0   (MI355X, round 4)",
        "# then the SQ pass of the same command (instruction counts of k_gao: VALU / LDS / SALU wave-instructions, wave cycles)"],
        read(src, "pmc_summary_cfg4.txt") + "\n## SQ counters\n" + read(src, "pmc_sq_cfg4.txt"))
    parts = [("device_decoder.txt", "scratch/bench_device_decoder.py (columns received in place)"),
             ("device_decoder_copied_columns.txt", "scratch/bench_device_decoder.py --copy (columns copied in by add())"),
             ("decoder_cfg5_shape.txt", "scratch/decoder_cfg5_shape.py (n = 256, t = 85, one GPU's shard of config 5, omega points)"),
             ("dec21_cfg3.txt", "scratch/dec21.py (the 21-liar open at config 3's shape alone: wall clock of the sixth, add() times)"),
             ("dec21_cfg5.txt", "scratch/dec21.py 256 85 (the 85-liar open at config 5's shard shape)"),
             ("dec21_cfg3_spread.txt", "scratch/dec21.py 64 21 spread freeze (the 21 liars one after every two honest senders: no candidate stands, the probe decides)"),
             ("dec21_cfg5_spread.txt", "scratch/dec21.py 256 85 spread freeze (the same at config 5's shard shape)"),
             ("kernel_stats_dec21_cfg5.txt", "rocprofv3 --kernel-trace --stats -- python scratch/dec21.py 256 85 (six 85-liar opens)"),
             ("symbols_fetch.txt", "scratch/time_fetch.py (hb_symbols_fetch alone; one add() while a candidate waits, by its parts)"),
             ("first_sight_host_phases.txt", "scratch/time_first_sight.py (where the host time of a fault-free first-sight decode goes)"),
             ("first_sight_outliers.txt", "scratch/first_sight_outliers.py 1500 all [nogc] (per-decode wall time; the interpreter's full collections)"),
             ("coalescer.txt", "scratch/bench_coalescer.py"),
             ("boundary_rates.txt", "scratch/boundary_rates.py"),
             ("stress_decoder.txt", "scratch/stress_decoder.py 240 31 (random shapes / liars -- half of the runs coordinated on one fake polynomial -- against the host mirror, after every column)")]
    body = ""
    for name, what in parts:
        txt = read(src, name)
        txt = "\n".join(ln for ln in txt.splitlines() if "amdgpu.ids" not in ln and "Optimistic decoding failed" not in ln)
        body += f"## {name}: {what}\n{txt}\n"
    put("r04_device_decoder_and_coalescer.txt", [f"# profiles/collect_r04.sh {tag}, one MI355X box, round 4 final state"], body)
    print("assembled from", src)


if __name__ == "__main__":
    main(sys.argv[1])
