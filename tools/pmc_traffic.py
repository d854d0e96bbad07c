#!/usr/bin/env python3
"""Parse the two rocprofv3 counter CSVs of tools/pmc_traffic.sh into the per-launch traffic record
(profiles/round1_pmc_conv_l4_0.json).  FETCH_SIZE is doubled (gfx950 counts a wide coalesced read at half its
bytes, MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported; both are in KB."""
import csv
import json
import sys

KERNEL = "k_conv_mfma<3, 3, 3, 1, 3, false, 0>"


def avg_counter(path, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, KERNEL, path))
    return sum(vals) / len(vals), len(vals)


def main():
    fetch, n = avg_counter(sys.argv[1], "FETCH_SIZE")
    write, _ = avg_counter(sys.argv[2], "WRITE_SIZE")
    algorithmic = 4 * (2 * 4 * 96 ** 3 * 40 + 27 * 40 * 40)
    rec = {
        "what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around "
                "tools/bench_layers.py --filter l4.0 --iters 1 on MI355X (tools/pmc_traffic.sh)",
        "kernel": "cfun_mfma::k_conv_mfma<3,3,3,1,3,false,0> (conv_norm_lrelu_l4.0 forward / data gradient: "
                  "3x3x3 40->40 @ 4x96^3)",
        "dispatches": n,
        "FETCH_SIZE_avg_KB": round(fetch, 1),
        "WRITE_SIZE_avg_KB": round(write, 1),
        "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts half of a wide (16 B/lane) coalesced read "
                      "on gfx950 -> doubled; WRITE_SIZE uncorrected",
        "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
        "algorithmic_bytes_per_launch": algorithmic,
    }
    print(json.dumps(rec, indent=2))


if __name__ == "__main__":
    main()
