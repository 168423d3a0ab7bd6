"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: sums of every counter, dispatch count and the MFMA
utilisation the way rocprof's derived MfmaUtil defines it:  sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) * 1024 SIMDs).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES \
              --kernel-trace --output-format csv -d DIR -- python bench.py ...
    python tools/pmc_summary.py DIR/.../*_counter_collection.csv [substring ...] [--json OUT]

--json OUT also writes {kernel: {"dispatches", "mfma_util", "valu_busy", "waves_per_simd"}} (fractions, not percent) --
profiles/attn_pmc.json, which tools/bench_s1.py quotes as `mfma_util_pmc`, is made this way.
"""
import collections
import csv
import sys


def main():
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    path, pats = args[0], args[1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if pats and not any(p in k for p in pats):
            continue
        k = k[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    order = sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0.0))
    for k in order[:30]:
        v = agg[k]
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0      # the CSV sums the counter over the 8 XCDs: /8 = kernel cycles
        line = f"{k:70s} dispatches {len(disp[k]):5d}"
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            line += f"  MfmaUtil {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):5.1f}%"
        if gui and "SQ_ACTIVE_INST_VALU" in v:
            line += f"  VALU-busy {100 * 4 * v['SQ_ACTIVE_INST_VALU'] / (gui * 1024):5.1f}%"      # quad-cycles
        if "SQ_WAVE_CYCLES" in v and gui:
            line += f"  waves/SIMD {4 * v['SQ_WAVE_CYCLES'] / (gui * 1024):4.2f}"
        print(line)
        print("      " + "  ".join(f"{c}={x:.3g}" for c, x in sorted(v.items())))
    if out_json:
        import json
        import re

        res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES "
                         "SQ_BUSY_CU_CYCLES --kernel-trace; MfmaUtil = sum(MFMA busy cycles) / (kernel cycles * 1024 SIMDs)"}
        for k in order:
            v = agg[k]
            gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if not gui:
                continue
            m = re.search(r"((?:attn|mha|gemm256|conv|wgrad)_\w+(?:<[^>]*>)?)", k)
            name = m.group(1) if m else k
            res[name] = dict(dispatches=len(disp[k]),
                             mfma_util=v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024),
                             valu_busy=4 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / (gui * 1024),
                             waves_per_simd=4 * v.get("SQ_WAVE_CYCLES", 0.0) / (gui * 1024))
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
