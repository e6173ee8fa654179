#!/usr/bin/env python3
"""profiles/r03_* from the outputs of collect_r03.sh: copies the tables and writes their header lines from the data itself.
usage: python profiles/assemble_r03.py gpurun_out/<tag>"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from assemble_r02 import avg, clean, last_json, put, rows  # noqa: E402


def main(src):
    j = last_json(os.path.join(src, "bench_default.json"))
    t3 = rows(os.path.join(src, "kernel_stats_cfg3.txt"))
    def avg_prefix(table, prefix, pick=None):
        """the most-launched row whose kernel name starts with `prefix` (one stream; pick = 'min': of the smallest LDS size)"""
        c = [r for r in table if r[0].startswith(prefix) and "streams" not in r[0]]
        if c and pick == "min":
            c = [r for r in c if r[1] == min(x[1] for x in c)]
        return max(c, key=lambda r: r[2])[3] if c else float("nan")

    enc = avg_prefix(t3, "hb::k_mm8<3, false, false")      # <K-blocks, CHECK, RAGGED, SKIP (K-block 0 without its second digit group)>
    dec = avg_prefix(t3, "hb::k_mm8<3, false, true")
    chk = avg_prefix(t3, "hb::k_mm8<3, true, false", "min")
    pre = avg(t3, "k_prescale_tab", None)

    f1 = avg_prefix(t3, "hb::k_mm8w<true, 3, 3")        # <CHECK, written-out K-blocks, sums per lane>
    f2 = avg_prefix(t3, "hb::k_mm8w<true, 3, 4")
    three = j["detail"]["shares_per_s_per_gpu_three_full_encodes"]
    put("r03_bench_cfg3_kernel_stats.txt", [
        "# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-sample 0   (MI355X, round 3, final state; summarised per kernel and launch geometry by profiles/summarize_rocpd.py)",
        f"# One open = hb::k_mm8<3,false,false,true> (R1 encode, {enc:.1f} us) + hb::k_mm8w<true,3,3> (R1 decode + validate fused: rows [V^-1 row 0 ; V[zc] V^-1], 22 x 22, row tiles of 12, {f1:.1f} us)",
        f"#          + hb::k_mm8w<true,3,4> (R2 decode + validate fused: rows [V^-1 ; V[zc] V^-1], 43 x 22, {f2:.1f} us) = {enc + f1 + f2:.1f} us of kernels in a {j['ms_per_step'] * 1e3:.0f} us step ({j['value'] / 1e9:.2f} G shares/s on this box).",
        "# Between the dependent launches of one stream the GPU idles 5.7 us (encode -> R1, R1 -> R2) and 10 us between steps (start / end timestamps of the 212 timed opens in this trace):",
        "# 21 us of a 208 us step; two opens in flight on two streams fill it (detail.shares_per_s_per_gpu_two_opens_in_flight).",
        f"# hb::k_mm8<3,false,true> ({dec:.1f} us), hb::k_mm8<3,true,false> with 53.9 KB of LDS ({chk:.1f} us) and k_prescale_tab ({pre:.1f} us) are the same open with HB_OPEN_OPT_FUSED_VALIDATE = 0 --",
        f"# the round-1 definition of the headline, detail.shares_per_s_per_gpu_three_full_encodes: {enc:.1f} + 2 x ({pre:.1f} + {dec:.1f} + {chk:.1f}) = {enc + 2 * (pre + dec + chk):.0f} us, {three / 1e9:.2f} G shares/s.",
        "# k_decode_check / k_matvec3 are the integer-VALU family (bench's secondary figure), [2 streams] rows the two-opens-in-flight figure; the template arguments of k_mm8w are <CHECK, written-out K-blocks, sums kept per lane>.",
        "# Round 3, second half: both headline kernels fold the high half of a sum on the matrix cores (DESIGN 4b / 4c; round 2: encode 54.8, R1 53.0, R2 85.1 us on the box of r03's first collections);",
        "# round 3's other kernel work is in hb_quick.hip (r03_device_decoder_*.txt) and in k_mm8w's unit numbering (r03_pmc_cfg5-shard_row_groups.txt)."],
        open(os.path.join(src, "kernel_stats_cfg3.txt")).read())
    for w, shape in (("cfg3-omega", "config 3 at omega-power points (n=64, t=21, 2^20 shares)"),
                     ("cfg5-shard", "one GPU's 1/8 shard of BASELINE config 5 (n=256, t=85, 2^19 shares, omega points, 6097 chunks)")):
        t = rows(os.path.join(src, f"kernel_stats_{w}.txt"))
        jw = last_json(os.path.join(src, f"bench_{w}.json"))
        ntt = avg(t, "k_ntt_lds<9, 8, false, true>", None)
        fused = sorted(r[3] for r in t if r[0].startswith("hb::k_mm8w<true"))
        put(f"r03_bench_{w}_kernel_stats.txt", [
            f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {w} --steps 20 --warmup 3 --cpu-sample 0 --no-two-streams-extra  (MI355X, round 3, final state: with the XCD-aware unit numbering of k_mm8w,",
            "# which took cfg5-shard's R2 launch from 114.4 to 110.4 us: r03_pmc_cfg5-shard_row_groups.txt)",
            f"# {shape}: k_ntt_lds<..,false,..> R1 encode ({ntt:.1f} us) + two hb::k_mm8w<true,..> launches (R1 / R2 decode + validate fused: {fused[0]:.1f} / {fused[-1]:.1f} us)"
            f" = {ntt + fused[0] + fused[-1]:.0f} us of kernels in a {jw['ms_per_step'] * 1e3:.0f} us step ({jw['value'] / 1e9:.2f} G shares/s; fusion off {jw['detail']['shares_per_s_per_gpu_three_full_encodes'] / 1e9:.2f} G)."],
            open(os.path.join(src, f"kernel_stats_{w}.txt")).read())
    for name in ("traffic_cfg3.json", "traffic_cfg3-omega.json", "traffic_cfg5-shard.json"):
        shutil.copy(os.path.join(src, name), os.path.join(HERE, name))
    tr = json.load(open(os.path.join(HERE, "traffic_cfg3.json")))
    put("r03_pmc_cfg3.txt", [
        "# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra   (MI355X, round 3)",
        "# three separate passes (FETCH_SIZE | WRITE_SIZE | SQ counters), per-launch averages by profiles/summarize_pmc.py; launches of one kernel with different shapes are separate rows.",
        "# HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; checked against TCC_EA0_RDREQ in r03_pmc_cfg5-shard_row_groups.txt).",
        f"# hb::k_mm8w<true,3,4> = the R2 launch (fused decode + validate, 43 x 22): {tr['hbm_bytes_per_launch'] / 1e6:.1f} MB of HBM traffic against 99.14 MB algorithmic (32 x 47663 x (22 + 21 + 22)); "
        f"{tr['valu_wave_instr_per_launch'] / 1e6:.2f} M wave-instructions, {tr['mfma_per_launch'] / 1e6:.2f} M MFMAs, matrix pipe busy {tr['mfma_busy_frac']:.2f}, VALU busy {tr['valu_busy_frac']:.2f} of the kernel's cycles."],
        open(os.path.join(src, "pmc_summary_cfg3.txt")).read())
    put("r03_pmc_cfg3-omega.txt", ["# same passes for --workload cfg3-omega (round 3)."], open(os.path.join(src, "pmc_summary_cfg3-omega.txt")).read())
    put("r03_pmc_cfg5-shard.txt", ["# same passes for --workload cfg5-shard (round 3, final state: the R2 launch (65536 threads) moves 62 MB = 1.24 x algorithmic; before the XCD-aware unit numbering 103.5 MB, r03_pmc_cfg5-shard_row_groups.txt)."],
        open(os.path.join(src, "pmc_summary_cfg5-shard.txt")).read())
    put("r03_config4_robust_decoders.txt", [
        "# scratch/bench_robust.py 262144 (config 4: n=100, t=33, 33 errors per codeword) and FETCH/WRITE passes at 16384 codewords.  Round 3 took k_gao from 117.4 ms / 2.23 M codewords/s to the figures",
        "# below in five measured steps (DESIGN 4e): one reduction for the two products of a coefficient update 117.4 -> 99.1 ms; pseudo-division + the field inversion moved to k_gao_finish, one LANE per",
        "# codeword (the Fermat power, computed by all 64 lanes of the codeword's wave, was 46 % of the kernel: 55.1 ms with the inversion stubbed out against 102.2) 99.1 -> 57.9 ms; the scale factor",
        "# in an idle lane and a degree fast path 57.9 -> 49.1 ms; the cofactor's update in the lanes the remainder's last round leaves free (two rounds a step, not three) 49.1 -> 40.5 ms; three waves to a SIMD 40.5 -> 38 ms.",
        "# HBM traffic per codeword: the interpolant g1 on k_mm8w 6.3 KB (ys in, g1 out), k_gao ~15 KB by the x2-corrected FETCH counter (g1 back in, raw quotient digits and cofactor out), k_gao_finish 4.6 KB",
        "# (the raw outputs in and the scaled ones out, one lane per codeword: uncoalesced) against 4.3 KB algorithmic -- the interpolant's round trip is the price of computing it on the matrix cores",
        "# (V^-1 of 100 x 100 does not fit beside the EEA's polynomials in LDS), the finishing kernel's re-read the price of 64 inversions per wave instead of one."],
        clean(os.path.join(src, "robust_cfg4.txt")) + open(os.path.join(src, "pmc_summary_cfg4.txt")).read())
    put("r03_device_decoder_and_coalescer.txt", [
        "# scratch/bench_device_decoder.py, scratch/bench_coalescer.py, scratch/boundary_rates.py, scratch/plan_create_cost.py (round 3: the device decoder is plan-free -- hb_quick_interp_check + hb_probe_*;",
        "# 'building them' = first sight of the arrival pattern, the figure VERDICT r2 item 2 asks for; round 2: fault-free 305 M, 5 liars 154 M, 21 liars 49 M shares/s.  'late chunks only' = every liar corrupts ONE chunk k > 0",
        "# of its own, polynomial 0 decodes clean.  The wb rows use the plan-free optimistic launch and the plan-based Welch-Berlekamp robust path.)"],
        clean(os.path.join(src, "device_decoder.txt")) + clean(os.path.join(src, "coalescer.txt")) + clean(os.path.join(src, "boundary_rates.txt")) + clean(os.path.join(src, "plan.txt"))
        + (clean(os.path.join(src, "decoder_cfg5_shape.txt")) if os.path.exists(os.path.join(src, "decoder_cfg5_shape.txt")) else ""))
    with open(os.path.join(HERE, "r03_bench_default_run.json"), "w") as f:
        f.write(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1] + "\n")
    with open(os.path.join(HERE, "r03_bench_other_workloads.json"), "w") as f:
        for w in ("cfg5-shard", "cfg3-omega", "cfg2", "cfg5"):
            f.write(open(os.path.join(src, f"bench_{w}.json")).read().strip().splitlines()[-1] + "\n")
    print("headline", j["value"] / 1e9, "three full encodes", three / 1e9, "enc/f1/f2", enc, f1, f2)


if __name__ == "__main__":
    main(sys.argv[1])
