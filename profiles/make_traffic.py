#!/usr/bin/env python3
"""profiles/traffic_<workload>.json from a PMC summary (profiles/summarize_pmc.py output): the counters of the roofline kernel of
a workload, which bench.py copies into roofline.traffic / roofline.second.  For plans that decode and validate in one launch
the roofline kernel is the R2 launch of k_mm8w<true, PEEL> (the row of that kernel that writes the most).
usage: python profiles/make_traffic.py <pmc_summary.txt> <workload> <source note>"""
import json
import sys


def main(path, workload, source):
    lines = open(path).read().splitlines()
    hdr = lines[0].split()
    has_lds = hdr[2] == "lds"              # summaries since round 5 carry the dispatch's dynamic LDS size after the grid
    skip = 3 if has_lds else 2
    cols = hdr[skip + 1:-2]                # counters between 'launches' and 'HBM bytes/launch'
    def parse(prefixes):
        rows = []
        for ln in lines[1:]:
            if not any(ln.startswith(p) for p in prefixes):
                continue
            name = ln[:44].strip()
            f = ln[44:].split()
            vals = dict(zip(cols, map(float, f[skip:skip + len(cols)])))
            vals["hbm_mb"] = float(f[skip + len(cols)]) if len(f) > skip + len(cols) else None
            vals["launches"] = int(f[skip - 1])
            rows.append((name, vals))
        return rows

    if workload.startswith("cfg4"):
        # one decode call = the interpolant on k_mm8w, k_gao, k_gao_finish: bytes of a call = the sum over its kernels' launches
        rows = parse(("hb::k_mm8w<false", "k_gao<", "k_gao_finish<"))
        if not rows:
            raise SystemExit("no Gao kernels in " + path)
        # a kernel appears in several rows (launch groups of the summary); two KINDS of call run in the profiled command: the
        # Welch-Berlekamp entry point (bench's `value`: Gao's kernels without the locators -- the rows with the least traffic) and
        # hb_gao_decode itself (bench's detail figure: locators written and scaled -- the rows with the most).  One row per kernel.
        by, wr = {}, {}
        for n, v in rows:
            if v.get("hbm_mb") is not None:
                by.setdefault(n, []).append(v["hbm_mb"])
                wr.setdefault(n, []).append((v.get("WRITE_SIZE", 0.0), v["hbm_mb"]))
        wb = {n: min(m) for n, m in by.items()}
        gao = {n: max(m) for n, m in by.items()}
        # k_mm8w<false, ...> also ENCODES the workload's inputs (34 -> 100 symbols, outside the timed region): the interpolant is the launch that
        # writes the most (n symbols a codeword), the same in both kinds of call
        for n in list(wb):
            if n.startswith("hb::k_mm8w"):
                wb[n] = gao[n] = max(wr[n])[1]
        print(json.dumps({"workload": workload, "kernel": "k_mm8w<false,...> (interpolant g1 = V^-1 y) + k_gao (fraction-free extended Euclid + pseudo-division, one wave per "
                          "codeword) + k_gao_finish (one field inversion per four codewords): the launches of one hb_wb_decode call (no locators written)",
                          "hbm_bytes_per_launch": sum(wb.values()) * 1e6, "per_kernel_MB": wb,
                          "hb_gao_decode_call": {"hbm_bytes_per_launch": sum(gao.values()) * 1e6, "per_kernel_MB": gao}, "source": source}, indent=1))
        return
    if workload.startswith("cfg3-p64"):
        # the 64-bit prime: three launches of k_mv64m (k_mv64 where the matrix-core image is not built) an open -- encode, R1, R2 -- told apart by what
        # they write (8 n C, 8 C, 8 C d bytes); bench.py's roofline segment is the slowest of the three: `by_segment` holds each one's bytes
        rows = parse(("hb::k_mv64m<", "hb::k_mv64<"))
        if not rows:
            raise SystemExit("no k_mv64m / k_mv64 row in " + path)
        rows = [r for r in rows if r[1].get("hbm_mb") is not None]
        order = sorted(rows, key=lambda r: r[1].get("WRITE_SIZE", 0.0))
        name, v = order[-1]
        seg = {"R1 encode": order[-1][1]["hbm_mb"] * 1e6}
        if len(order) >= 3:
            seg["R2 decode + validate"] = order[-2][1]["hbm_mb"] * 1e6
            seg["R1 decode + validate"] = order[-3][1]["hbm_mb"] * 1e6
        print(json.dumps({"workload": workload, "kernel": f"{name} (the R1 encode: n x d mat-vec at 8-byte elements, hb_narrow.hip)", "hbm_bytes_per_launch": v["hbm_mb"] * 1e6,
                          "by_segment": seg, "launches_averaged": v.get("launches"),
                          "per_launch_MB": {f"{n} [{int(x.get('WRITE_SIZE', 0))} KB written]": x["hbm_mb"] for n, x in rows}, "source": source}, indent=1))
        return
    # plans at small-integer points decode + validate on k_mm8f (hb_mfma_fused.hip); the others on k_mm8w<true, PEEL>
    rows = parse(("hb::k_mm8f<",))
    small = bool(rows)
    if not rows:
        rows = parse(("hb::k_mm8w<true",))
    if not rows:
        raise SystemExit("no k_mm8f / k_mm8w<true,...> row in " + path)
    name, v = max(rows, key=lambda r: r[1].get("WRITE_SIZE", 0.0))
    # the R2 launch writes the d coefficient rows of every chunk, 32 C d bytes: a row that is a MIX of launches (R1's write one row) shows here
    expect_kb = {"cfg3": 32 * 47663 * 22, "cfg3-omega": 32 * 47663 * 22, "cfg5-shard": 32 * 6097 * 86}.get(workload)
    if expect_kb is not None:
        expect_kb /= 1024.0
        assert abs(v["WRITE_SIZE"] - expect_kb) <= 0.05 * expect_kb, f"the chosen row writes {v['WRITE_SIZE']:.0f} KB a launch, the R2 launch of {workload} {expect_kb:.0f} KB: not the R2 launch alone"
    simds, ses = 1024, 32                  # SQ_BUSY_CYCLES sums 32 shader-engine instances, instruction counters all 1024 SIMDs
    cycles = v["SQ_BUSY_CYCLES"] / ses
    out = {
        "workload": workload,
        "kernel": (f"{name} (R2: decode + validate as [N ; P] (y ./ den) on the small-entry kernel, the division by den_j inside the kernel)" if small else
                   f"{name} (R2: fused decode + validate; the sums of a pass are reduced, stored and compared inside the next pass)"),
        "hbm_bytes_per_launch": v["hbm_mb"] * 1e6 if v["hbm_mb"] is not None else None,
        "launches_averaged": v.get("launches"),
        "valu_wave_instr_per_launch": int(v["SQ_INSTS_VALU"]),
        "mfma_per_launch": int(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 16),
        "valu_busy_frac": round(4 * v["SQ_INSTS_VALU"] / simds / cycles, 4),
        "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cycles, 4),
        "kernel_cycles": round(cycles),
        "source": source,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
