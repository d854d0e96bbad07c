#!/usr/bin/env python3
"""Parse the two rocprofv3 counter CSVs of tools/pmc_traffic.sh into the per-launch traffic record
(profiles/roundN_pmc_wino_l4_0.json).  FETCH_SIZE is doubled (gfx950 counts a wide coalesced read at half its
bytes, MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported; both are in KB."""
import csv
import hashlib
import json
import os
import re
import sys

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfun_amd", "csrc", os.environ.get("CFUN_PMC_SRC", "conv3d_wino.hip"))


def git_blob_sha1(path):
    """= `git hash-object path` (there is no .git on the GPU box)."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()

# the kernel(s) of ONE conv call (regex over rocprofv3's kernel names; several kernels' per-dispatch averages are summed):
# the forward / data gradient of l4.0 runs k_conv_wino<NSUB = 3, S2D = false, TWOD = false, STATS = false, SB = false>
KERNEL = os.environ.get("CFUN_PMC_KERNEL", r"k_conv_wino<3, false, false, false, false>")


def kernel_label(name):
    m = re.search(r"(k_\w+<[^>]*>)", name)
    return m.group(1) if m else name[:60]


def per_kernel_avg(path, counters):
    """{kernel label: {counter: (average per dispatch, dispatches)}} of the kernels matching KERNEL"""
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if re.search(KERNEL, row["Kernel_Name"]) and row["Counter_Name"] in counters:
                acc.setdefault(kernel_label(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not acc:
        raise SystemExit("no %s rows for %s in %s" % (counters, KERNEL, path))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in acc.items()}


def avg_counter(path, counter):
    """per-call figure: the sum over the call's kernels of each kernel's per-dispatch average"""
    per = per_kernel_avg(path, (counter,))
    return sum(d[counter][0] for d in per.values()), min(d[counter][1] for d in per.values()), per


def main():
    fetch, n, per_f = avg_counter(sys.argv[1], "FETCH_SIZE")
    write, _, per_w = avg_counter(sys.argv[2], "WRITE_SIZE")
    algorithmic = 4 * (2 * 4 * 96 ** 3 * 40 + 27 * 40 * 40)
    rec = {
        "what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around "
                "tools/bench_layers.py --filter l4.0 --iters 1 on MI355X (tools/pmc_traffic.sh)",
        "kernel": "%s (conv_norm_lrelu_l4.0 forward / data gradient: 3x3x3 40->40 @ 4x96^3; per CALL = the sum over "
                  "these kernels of their per-dispatch averages)" % " + ".join(sorted(per_f)),
        "dispatches": n,
        "per_kernel_avg_KB": {k: {"FETCH_SIZE": round(per_f[k]["FETCH_SIZE"][0], 1),
                                  "WRITE_SIZE": round(per_w.get(k, {}).get("WRITE_SIZE", (0.0, 0))[0], 1)} for k in sorted(per_f)},
        "FETCH_SIZE_avg_KB": round(fetch, 1),
        "WRITE_SIZE_avg_KB": round(write, 1),
        "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts half of a wide (16 B/lane) coalesced read "
                      "on gfx950 -> doubled; WRITE_SIZE uncorrected",
        "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
        "algorithmic_bytes_per_launch": algorithmic,
        "kernel_src": "cfun_amd/csrc/" + os.path.basename(SRC),
        "kernel_src_blob": git_blob_sha1(SRC),      # bench.py reports `traffic` only while this still matches
    }
    print(json.dumps(rec, indent=2))


if __name__ == "__main__":
    main()
