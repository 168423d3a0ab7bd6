"""profiles/pmc_traffic.json from two rocprofv3 counter-collection CSVs (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE;
MI355X_MICROARCH.md, HBM section): per kernel the mean KB per dispatch; FETCH_SIZE is doubled (gfx950 reports half of the
bytes of wide streaming reads), WRITE_SIZE is used as is (calibration kernel printed: dact_mul reads two tensors and writes
one of the same size).  Kernels are keyed by the tag bench.py's roofline leg uses.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv [eager steps in the run = 4] > profiles/pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys

TAGS = {   # tag of bench.py's roofline leg -> substring of the rocprof kernel name
    "conv_deep<bf16, 128, 128, 64>": "::conv_deep(",
    "conv_deep32<bf16, 128, 128, 32>": "::conv_deep32(",
    "conv_ring<bf16, 64, 64, 64, x4>": "::conv_ring<2, 2, 4>(",
    "wgrad_deep<bf16, 128, 5x32, 64>": "::wgrad_deep(",
    "wgrad_ring<bf16, 64, 5x32, 64, x4>": "::wgrad_ring<2, 5, 4>(",
    "dact_mul_kernel (calibration)": "dact_mul_kernel",
}
# kernels that only the HiFi-GAN vocoder (`dec`) launches: the fused ResBlock steps and their helpers.  Together with the
# shared conv kernels above they are what roofline.hifigan_dec times; their counter bytes go into the "hifigan_dec" row.
DEC_ONLY = ["resunit_fwd_multi<", "resunit_bwd_multi<", "resunit_wide<", "fold_partials_multi(", "wgrad_halo<",
            "conv_wgrad_tr<", "conv_narrow"]


def means(path, counter, with_counts=False):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        cnt[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    if with_counts:
        return {k: tot[k] / max(len(cnt[k]), 1) for k in tot}, {k: len(cnt[k]) for k in tot}
    return {k: tot[k] / max(len(cnt[k]), 1) for k in tot}


def main():
    (f, fc), w = means(sys.argv[1], "FETCH_SIZE", True), means(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4          # eager steps in the profiled run (warmup + steps)
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) --kernel-trace -- python bench.py --workload s2 "
                   "--steps 2 --warmup 2 --no-extras --graphs 0; mean KB per dispatch; traffic = 2 x FETCH (gfx950 note) + WRITE"}
    for tag, sub in TAGS.items():
        fk = [v for k, v in f.items() if sub in k]
        wk = [v for k, v in w.items() if sub in k]
        if fk and wk:
            out[tag] = {"fetch_kb_raw": round(fk[0], 1), "write_kb": round(wk[0], 1),
                        "traffic_bytes_per_launch": int((2 * fk[0] + wk[0]) * 1024)}
    # every instantiation of the vocoder-only kernels, and their sum per step
    dec_rows, dec_bytes, dec_launches = {}, 0.0, 0
    for k in sorted(f):
        if not any(sub in k for sub in DEC_ONLY) or k not in w:
            continue
        short = re.sub(r"\(anonymous namespace\)::|evt_conv::|^void ", "", k)
        short = re.match(r"\s*([\w:]+(<[^>]*>)?)", short).group(1)
        per = (2 * f[k] + w[k]) * 1024
        dec_rows[short] = {"fetch_kb_raw": round(f[k], 1), "write_kb": round(w[k], 1), "traffic_bytes_per_launch": int(per),
                           "launches_per_step": round(fc[k] / steps, 2)}
        dec_bytes += per * fc[k] / steps
        dec_launches += fc[k]
    out["hifigan_dec_only_kernels"] = {"_what": "fused ResBlock steps of the vocoder and their helpers (kernels no other "
                                                "module launches); 2 x FETCH + WRITE, per launch and summed per step",
                                       "traffic_gb_per_step": round(dec_bytes / 1e9, 4),
                                       "launches_per_step": round(dec_launches / steps, 1), "kernels": dec_rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
