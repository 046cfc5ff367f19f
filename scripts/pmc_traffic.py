"""Per-launch HBM traffic of the bench kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE runs: together they do not fit the TCC counter slots).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
  python scripts/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, "HBM"): both counters are reported in KB; on gfx950 FETCH_SIZE counts
128-byte requests of wide coalesced (16 B/lane) streaming reads as 64 bytes, so it is DOUBLED here; WRITE_SIZE is
exact for 16 B/lane streaming stores.  Infinity-Cache hits are included in both (memory-side L2 requests).
"""
import collections
import csv
import json
import re
import sys

CLASSES = [  # kernel-name regex -> bench.py timer class
    (r"ig_conv8[a-d]?_kernel", "conv_igemm_fprop_dgrad"),
    (r"ig_wgrad8(v2|h)_kernel", "conv_wgrad_igemm"),
    (r"ig_wgrad_reduce_v2_kernel|wgrad_reduce_many_kernel", "conv_wgrad_reduce"),
    (r"ncv_fwd", "conv_narrow_fprop_dgrad"),
    (r"ncv_wgrad2?_kernel", "conv_wgrad_narrow"),
    (r"bn_stats_kernel", "bn_fwd_stats"),
    (r"bn_apply2?_kernel", "bn_fwd_apply"),
    (r"bn_bwd_reduce_kernel", "bn_bwd_reduce"),
    (r"bn_bwd_apply2?_kernel", "bn_bwd_apply"),
    (r"bn_bwd_fused_kernel", "bn_bwd_fused"),
    (r"ew_vec_kernel", "elementwise"),
    (r"gemm_bf16", "gemm_bf16"),
]


def collect(path, counter):
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for rx, tag in CLASSES:
            if re.search(rx, name):
                per_kernel[tag][0] += 1
                per_kernel[tag][1] += float(r["Counter_Value"]) * 1024.0
                break
    return per_kernel


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {"_note": "bytes per launch; fetch_bytes = 2 x FETCH_SIZE (gfx950 wide-read correction), write_bytes = WRITE_SIZE; separate --pmc passes",
           "classes": {}}
    for tag in sorted(set(f) | set(w)):
        fb = 2.0 * f[tag][1] / max(f[tag][0], 1)
        wb = w[tag][1] / max(w[tag][0], 1)
        res["classes"][tag] = {"launches_fetch_pass": f[tag][0], "launches_write_pass": w[tag][0], "fetch_bytes": fb, "write_bytes": wb,
                               "traffic_bytes": fb + wb}
    json.dump(res, open(out, "w"), indent=1)
    for tag, v in res["classes"].items():
        print(f"{tag:28s} fetch {v['fetch_bytes'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
