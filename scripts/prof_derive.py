#!/usr/bin/env python3
"""Derived figures from the text summaries scripts/prof_summary.py writes (no GPU, no sqlite needed).

    scripts/prof_derive.py <summary.txt>... [--traffic-json workload]

Per traversal kernel: VALU busy share of the kernel's cycles, active lanes per VALU instruction,
resident waves per SIMD, TA / TD busy, L2 hit rate, HBM-side bytes per launch (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE).  With --traffic-json the per-kernel
traffic is printed as the JSON object bench.py reads from profiles/r01_traffic.json.
"""
import json
import re
import sys

NSIMD, NCU, NSE, NXCD = 1024, 256, 32, 8


DISP = {}          # (kernel, counter) -> dispatches


def load(files):
    d = {}
    for fn in files:
        for line in open(fn):
            m = re.match(r"(.{50}) (\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m:
                kern = re.sub(r"^void ", "", m.group(1).strip())
                kern = re.sub(r"^rtx::", "", kern).split("(")[0]
                d[(kern, m.group(2))] = float(m.group(5))
                DISP[(kern, m.group(2))] = int(m.group(3))
    return d


def main():
    args = sys.argv[1:]
    want_json = None
    if "--traffic-json" in args:
        i = args.index("--traffic-json")
        want_json = args[i + 1]
        del args[i:i + 2]
    pmc_json = None
    if "--pmc-json" in args:
        i = args.index("--pmc-json")
        pmc_json = args[i + 1]
        del args[i:i + 2]
    frames_per_launch = None          # (frames a render call of the profiled command carries: bench.py scales per-launch bytes by it)
    if "--frames-per-launch" in args:
        i = args.index("--frames-per-launch")
        frames_per_launch = int(args[i + 1])
        del args[i:i + 2]
    d = load(args)
    kernels = sorted({k for k, _ in d if k.startswith("k_")})
    if pmc_json:
        # what bench.py's roofline reads (profiles/rNN_pmc.json): per kernel, HBM-side bytes per launch and the
        # vector-ALU figures; of a kernel's template variants the one launched most often counts (the scheduler the
        # tuner settled on; the other variant's launches are its trials)
        out = {}
        for k in kernels:
            g = lambda c: d.get((k, c))  # noqa: E731
            name = k.split("<")[0]
            rec = {}
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
                rec.update({"fetch_kb_raw": g("FETCH_SIZE"), "write_kb_raw": g("WRITE_SIZE"),
                            "hbm_bytes_per_launch": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024})
            if g("SQ_BUSY_CYCLES") and g("SQ_INSTS_VALU"):
                busy = g("SQ_BUSY_CYCLES") / NSE
                rec["valu_insts"] = g("SQ_INSTS_VALU")
                # Two estimators of "share of the kernel's SIMD cycles issuing vector ALU instructions":
                #  * SQ_ACTIVE_INST_VALU (quad-cycles waves spent executing VALU) over the shader engines' busy cycles: it
                #    saturates -- and, being normalised by the SE's busy time rather than each SIMD's, overshoots 1 by a few
                #    per cent on a saturated kernel (1.06 on the hall's camera kernel) -- so it is clamped: 1.0 = saturated;
                #  * 4 cycles (the issue time of a wave64 fp32 instruction on a 16-lane SIMD; longer ones exist) x
                #    SQ_INSTS_VALU over SIMDs x the kernel's own cycles (GRBM_GUI_ACTIVE per XCD): a lower bound that
                #    cannot exceed 1.
                rec["valu_busy_raw"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (NSIMD * busy)
                rec["valu_busy"] = min(1.0, rec["valu_busy_raw"])
                if g("GRBM_GUI_ACTIVE"):
                    rec["valu_issue_min"] = 4.0 * g("SQ_INSTS_VALU") / (NSIMD * g("GRBM_GUI_ACTIVE") / NXCD)
                if g("SQ_THREAD_CYCLES_VALU"):
                    rec["active_lanes"] = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU"))
                if g("SQ_WAVE_CYCLES"):
                    rec["waves_per_simd"] = 4.0 * g("SQ_WAVE_CYCLES") / (NSIMD * busy)
            if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
                cyc = g("GRBM_GUI_ACTIVE") / NXCD
                rec["ta_busy"] = g("TA_TA_BUSY_sum") / (NCU * cyc)
                rec["td_busy"] = g("TD_TD_BUSY_sum") / (NCU * cyc)
            if g("TCC_REQ_sum"):
                rec["l2_hit"] = g("TCC_HIT_sum") / g("TCC_REQ_sum")
            if not rec:
                continue
            rec["launches_profiled"] = max([v for (kk, _), v in DISP.items() if kk == k] or [0])
            if frames_per_launch:
                rec["frames_per_launch"] = frames_per_launch
            if name in out and out[name]["launches_profiled"] >= rec["launches_profiled"]:
                continue
            out[name] = rec
        print(json.dumps({pmc_json: out}, indent=1))
        return
    if want_json:
        out = {}
        for k in kernels:
            f, w = d.get((k, "FETCH_SIZE")), d.get((k, "WRITE_SIZE"))
            if f is None or w is None:
                continue
            name = k.split("<")[0]
            if name in out and "<false>" in k:
                continue
            out[name] = {"fetch_kb_raw": f, "write_kb_raw": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
        print(json.dumps({want_json: out}, indent=1))
        return
    print("== derived (scripts/prof_derive.py; per launch) ==")
    for k in kernels:
        g = lambda c: d.get((k, c))  # noqa: E731
        if g("SQ_BUSY_CYCLES") is None or g("SQ_INSTS_VALU") is None:
            continue
        busy = g("SQ_BUSY_CYCLES") / NSE          # cycles the kernel kept a shader engine busy
        line = "%-24s VALU insts %.4g, VALU busy %.1f %% of SIMD cycles%s" % (k[:24], g("SQ_INSTS_VALU"), min(100.0, 400.0 * g("SQ_ACTIVE_INST_VALU") / (NSIMD * busy)),
                                                                             " (estimator saturated: raw %.1f %%)" % (400.0 * g("SQ_ACTIVE_INST_VALU") / (NSIMD * busy)) if 4.0 * g("SQ_ACTIVE_INST_VALU") > NSIMD * busy else "")
        if g("GRBM_GUI_ACTIVE"):
            line += ", VALU issue >= %.1f %% (4 cycles x instructions / SIMD cycles of the kernel)" % (400.0 * g("SQ_INSTS_VALU") / (NSIMD * g("GRBM_GUI_ACTIVE") / NXCD))
        if g("SQ_THREAD_CYCLES_VALU"):
            line += ", active lanes %.1f %%" % (100.0 * g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU")))
        if g("SQ_WAVE_CYCLES"):
            line += ", waves/SIMD %.2f" % (4.0 * g("SQ_WAVE_CYCLES") / (NSIMD * busy))
        if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
            cyc = g("GRBM_GUI_ACTIVE") / NXCD
            line += ", TA busy %.0f %%, TD busy %.0f %%" % (100 * g("TA_TA_BUSY_sum") / (NCU * cyc), 100 * g("TD_TD_BUSY_sum") / (NCU * cyc))
        if g("TCC_REQ_sum"):
            line += ", L2 hit %.0f %%" % (100 * g("TCC_HIT_sum") / g("TCC_REQ_sum"))
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            line += ", HBM-side bytes %.3g" % ((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
        print(line)


if __name__ == "__main__":
    main()
