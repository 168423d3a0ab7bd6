"""profiles/pmc_traffic.json from two rocprofv3 counter-collection CSVs (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE;
MI355X_MICROARCH.md, HBM section): per kernel the mean KB per dispatch; FETCH_SIZE is doubled (gfx950 reports half of the
bytes of wide streaming reads), WRITE_SIZE is used as is (calibration kernel printed: dact_mul reads two tensors and writes
one of the same size).  Kernels are keyed by the tag bench.py's roofline leg uses.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv > profiles/pmc_traffic.json
"""
import collections
import csv
import json
import sys

TAGS = {   # tag of bench.py's roofline leg -> substring of the rocprof kernel name
    "conv_deep<bf16, 128, 128, 64>": "::conv_deep(",
    "conv_deep32<bf16, 128, 128, 32>": "::conv_deep32(",
    "conv_ring<bf16, 64, 64, 64, x4>": "::conv_ring<2, 2, 4>(",
    "wgrad_deep<bf16, 128, 5x32, 64>": "::wgrad_deep(",
    "wgrad_ring<bf16, 64, 5x32, 64, x4>": "::wgrad_ring<2, 5, 4>(",
    "resunit_fwd<bf16, 16>": "resunit_fwd<16,",
    "resunit_fwd<bf16, 32>": "resunit_fwd<32,",
    "dact_mul_kernel (calibration)": "dact_mul_kernel",
}


def means(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        cnt[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return {k: tot[k] / max(len(cnt[k]), 1) for k in tot}


def main():
    f, w = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) --kernel-trace -- python bench.py --workload s2 "
                   "--steps 2 --warmup 2 --no-extras --graphs 0; mean KB per dispatch; traffic = 2 x FETCH (gfx950 note) + WRITE"}
    for tag, sub in TAGS.items():
        fk = [v for k, v in f.items() if sub in k]
        wk = [v for k, v in w.items() if sub in k]
        if fk and wk:
            out[tag] = {"fetch_kb_raw": round(fk[0], 1), "write_kb": round(wk[0], 1),
                        "traffic_bytes_per_launch": int((2 * fk[0] + wk[0]) * 1024)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
