#!/usr/bin/env python3
"""profiles/r02_* from the outputs of collect_r02.sh: copies the tables and writes their header lines from the data itself.
usage: python profiles/assemble_r02.py gpurun_out/<tag> [gpurun_out/<tag of a bench-only rerun>]"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def rows(path):
    """kernel-stats table -> list of (name, lds bytes, calls, avg us)"""
    out = []
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"(.+?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m:
            out.append((m.group(1).strip(), int(m.group(6)), int(m.group(7)), float(m.group(8))))
    return out


def avg(table, name, pick):
    """average launch time of kernel `name` (exact short name, no [2 streams] variants); pick: 'min' / 'max' LDS size or None"""
    c = [r for r in table if r[0] == name]
    if not c:
        return float("nan")
    if pick == "min":
        c = [r for r in c if r[1] == min(x[1] for x in c)]
    elif pick == "max":
        c = [r for r in c if r[1] == max(x[1] for x in c)]
    return max(c, key=lambda r: r[2])[3]


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def clean(path):
    return "\n".join(ln for ln in open(path).read().splitlines() if "amdgpu.ids" not in ln) + "\n"


def put(name, hdr, body):
    with open(os.path.join(HERE, name), "w") as f:
        f.write("\n".join(hdr) + "\n" + body)


def main(src, bench_src):
    j = last_json(os.path.join(bench_src, "bench_default.json"))
    t3 = rows(os.path.join(src, "kernel_stats_cfg3.txt"))
    enc = avg(t3, "hb::k_mm8<3, false, false>", None)
    dec = avg(t3, "hb::k_mm8<3, false, true>", None)
    chk = avg(t3, "hb::k_mm8<3, true, false>", "min")
    chk_c = avg(t3, "hb::k_mm8<3, true, false>", "max")
    pre = avg(t3, "k_prescale_tab", None)
    fk = next((r[0] for r in t3 if r[0].startswith("hb::k_mm8w<true")), "hb::k_mm8w<true, 3, 3>")
    f1 = min((r[3] for r in t3 if r[0].startswith("hb::k_mm8w<true") and "streams" not in r[0]), default=float("nan"))
    f2 = max((r[3] for r in t3 if r[0].startswith("hb::k_mm8w<true") and "streams" not in r[0] and r[2] > 50), default=float("nan"))
    three = j["detail"]["shares_per_s_per_gpu_three_full_encodes"]
    put("r02_bench_cfg3_kernel_stats.txt", [
        "# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0   (MI355X, round 2, final state; summarised per kernel and launch geometry by profiles/summarize_rocpd.py)",
        f"# One open = hb::k_mm8<3,false,false> (R1 encode, {enc:.1f} us) + hb::k_mm8w<true,..> (R1 decode + validate fused: rows [V^-1 row 0 ; V[zc] V^-1], 22 x 22, row tiles of 12, {f1:.1f} us)",
        f"#          + hb::k_mm8w<true,..> with 104 KB of LDS (R2 decode + validate fused: rows [V^-1 ; V[zc] V^-1], 43 x 22, {f2:.1f} us) = {enc + f1 + f2:.1f} us of kernels in a {j['ms_per_step'] * 1e3:.0f} us step ({j['value'] / 1e9:.2f} G shares/s on this box).",
        f"# The rows hb::k_mm8<3,false,true> (decode over the numerators, {dec:.1f} us), hb::k_mm8<3,true,false> with 53.9 KB of LDS (validating re-encode of all 64 points, {chk:.1f} us) and k_prescale_tab ({pre:.1f} us) are the same open",
        f"# with HB_OPEN_OPT_FUSED_VALIDATE = 0 -- the round-1 definition of the headline, bench's detail.shares_per_s_per_gpu_three_full_encodes: {enc:.1f} + 2 x ({pre:.1f} + {dec:.1f} + {chk:.1f}) = {enc + 2 * (pre + dec + chk):.0f} us, {three / 1e9:.2f} G shares/s on this box",
        f"# (round 1: 61.4 / 19.5 / 34 / 62 us, 0.286-0.327 ms); hb::k_mm8<3,true,false> with 64 KB of LDS is the compact check matrix of HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY ({chk_c:.1f} us).",
        "# k_decode_check / k_matvec3 are the integer-VALU family (bench's secondary figure), [2 streams] rows the two-opens-in-flight figure; the template arguments of k_mm8w are <CHECK, written-out K-blocks, sums kept per lane>."],
        open(os.path.join(src, "kernel_stats_cfg3.txt")).read())
    for w, shape, r1txt in (("cfg3-omega", "config 3 at omega-power points (n=64, t=21, 2^20 shares)", "0.72 ms / 1.46 G shares/s; first version of round 2 (decode on k_mm8w + NTT check): 0.47 ms / 2.23 G"),
                            ("cfg5-shard", "one GPU's 1/8 shard of BASELINE config 5 (n=256, t=85, 2^19 shares, omega points, 6097 chunks)", "k_matvec2 228-240 us per decode, k_ntt_lds 97-108 us; 0.82 ms, 0.64 G shares/s")):
        t = rows(os.path.join(src, f"kernel_stats_{w}.txt"))
        jw = last_json(os.path.join(bench_src, f"bench_{w}.json"))
        ntt = avg(t, "k_ntt_lds<9, 8, false, true>", None)
        fused = sorted(r[3] for r in t if r[0].startswith("hb::k_mm8w<true"))
        put(f"r02_bench_{w}_kernel_stats.txt", [
            f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {w} --steps 20 --warmup 3 --cpu-sample 0 --no-two-streams-extra  (MI355X, round 2, final state)",
            f"# {shape}: k_ntt_lds<..,false,..> R1 encode ({ntt:.1f} us) + two hb::k_mm8w<true,..> launches (R1 / R2 decode + validate fused: {fused[0]:.1f} / {fused[-1]:.1f} us)"
            f" = {ntt + fused[0] + fused[-1]:.0f} us of kernels in a {jw['ms_per_step'] * 1e3:.0f} us step ({jw['value'] / 1e9:.2f} G shares/s; fusion off {jw['detail']['shares_per_s_per_gpu_three_full_encodes'] / 1e9:.2f} G).",
            "# k_ntt_lds<..,true,..> (validating re-encode by NTT), hb::k_mm8w<false,..> (plain decode; the 2-call row is the input generator), k_matvec2/3, k_prescale: the unfused pipeline and the integer-VALU family (bench's secondary figures).",
            f"# Round 1: {r1txt}."],
            open(os.path.join(src, f"kernel_stats_{w}.txt")).read())
    tr = json.load(open(os.path.join(HERE, "traffic_cfg3.json")))
    passes = 3 * 2979
    put("r02_pmc_cfg3.txt", [
        "# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra   (MI355X, round 2, final state)",
        "# three separate passes (FETCH_SIZE | WRITE_SIZE | SQ counters), per-launch averages by profiles/summarize_pmc.py; launches of one kernel with different shapes are separate rows.",
        "# HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read, MI355X_MICROARCH.md).",
        f"# hb::k_mm8w<true,..>, the row writing 32.8 MB = the R2 launch (fused decode + validate, 43 x 22): {tr['hbm_bytes_per_launch'] / 1e6:.1f} MB of HBM traffic against 99.14 MB algorithmic (32 x 47663 x (22 + 21 + 22)); {tr['valu_wave_instr_per_launch'] / 1e6:.2f} M wave-instructions",
        f"#   (VALU + MFMA; {tr['valu_wave_instr_per_launch'] / passes:.0f} per pass of 16 chunks x 16 rows), {tr['mfma_per_launch'] / 1e6:.2f} M MFMAs = {tr['mfma_per_launch'] * 16 / 1024 / 1e3:.1f} k cycles per SIMD of the kernel's {tr['kernel_cycles'] / 1e3:.1f} k ({100 * tr['mfma_busy_frac']:.0f} %); 4 cycles per instruction would be {4 * tr['valu_wave_instr_per_launch'] / 1024 / 1e3:.1f} k ({100 * tr['valu_busy_frac']:.0f} %): one wave per SIMD issues every ~5.5 cycles.",
        "#   (Before the 8 x 8 K-block layout: 28.26 M wave-instructions, 5.04 M MFMAs, 200.3 k cycles.)",
        "# hb::k_mm8<3,false,false> = the R1 encode (131.66 MB vs 131.17 MB algorithmic, 18.6 M wave-instructions, VALU 63.5 % / matrix pipe 38 % busy)."],
        open(os.path.join(src, "pmc_summary_cfg3.txt")).read())
    put("r02_pmc_cfg3-omega.txt", ["# same passes for --workload cfg3-omega (final state).  hb::k_mm8w<true,..> rows: R1 (22 rows, writes 1.5 MB) and R2 (43 rows, writes 32.8 MB) fused decode + validate launches."],
        open(os.path.join(src, "pmc_summary_cfg3-omega.txt")).read())
    t5 = json.load(open(os.path.join(HERE, "traffic_cfg5-shard.json")))
    put("r02_pmc_cfg5-shard.txt", [
        f"# same passes for --workload cfg5-shard (final state).  hb::k_mm8w<true,..> rows: R1 (86 rows) and R2 (171 rows x 86): {t5['hbm_bytes_per_launch'] / 1e6:.1f} MB of HBM traffic against 50.1 MB algorithmic = 32 x 6097 x (86 + 85 + 86).",
        f"# Writes are exact (16.9 MB); the reads are ~45.7 MB against 33.4 MB.  Not re-reads of the tile (a unit's 44 KB tile is DMA'd once and serves all 11 row tiles): probably line-granularity overfetch -- every 16-chunk run is 512 B starting at a multiple of 32 B only (C = 6097 is odd) -- plus the 484 KB digit image fetched once per XCD.  MFMA busy {100 * t5['mfma_busy_frac']:.0f} % of the kernel's cycles."],
        open(os.path.join(src, "pmc_summary_cfg5-shard.txt")).read())
    put("r02_config4_robust_decoders.txt", ["# scratch/bench_robust.py 262144 (config 4: n=100, t=33, 33 errors per codeword) and FETCH/WRITE passes at 16384 codewords (final state)"],
        clean(os.path.join(src, "robust_cfg4.txt")) + open(os.path.join(src, "pmc_summary_cfg4.txt")).read())
    extra = os.path.join(src, "plan.txt")
    put("r02_device_decoder_and_coalescer.txt", ["# scratch/bench_device_decoder.py, scratch/bench_coalescer.py, scratch/boundary_rates.py, scratch/plan_create_cost.py (final state: fused matrices built at a plan's third decode;",
                                                 "# built at plan creation: plan creation 3.35 / 2.79 / 5.95 / 2.23 ms, 5 liars 112.8 M shares/s)"],
        clean(os.path.join(src, "device_decoder.txt")) + clean(os.path.join(src, "coalescer.txt")) + clean(os.path.join(src, "boundary_rates.txt")) + (clean(extra) if os.path.exists(extra) else ""))
    with open(os.path.join(HERE, "r02_bench_default_run.json"), "w") as f:
        f.write(open(os.path.join(bench_src, "bench_default.json")).read().strip().splitlines()[-1] + "\n")
    with open(os.path.join(HERE, "r02_bench_other_workloads.json"), "w") as f:
        for w in ("cfg5-shard", "cfg3-omega", "cfg2", "cfg5", "valu"):
            pth = os.path.join(bench_src, f"bench_{w}.json")
            if os.path.exists(pth):
                f.write(open(pth).read().strip().splitlines()[-1] + "\n")
    print("headline", j["value"] / 1e9, "three full encodes", three / 1e9, "enc/f1/f2", enc, f1, f2)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
