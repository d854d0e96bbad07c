#!/usr/bin/env python3
"""Instruction mix per MFMA of every conv / wgrad launch of tools/bench_layers.py from one rocprofv3 --pmc pass:

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \\
        --output-format csv -d DIR -o p -- python tools/bench_layers.py --iters 1
    python tools/pmc_inst_table.py DIR/.../p_counter_collection.csv

MFMA-pipe utilisation as tools/pmc_mfma_table.py; VALU / SALU / LDS instructions per MFMA instruction: where a kernel spends
issue slots between its MFMAs (a 16x16x4 fp32 MFMA occupies the pipe for 32 cycles = 8 issue slots of its SIMD)."""
import collections
import csv
import re
import sys

disp = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = disp.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {"grid": r["Grid_Size"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for (_, name), v in disp.items():
    if not re.search(r"mfma|wgrad|wino", name) or "weights" in name:
        continue
    m = re.search(r"(k_\w+<[^>]*>)", name)
    a = agg.setdefault((m.group(1) if m else name[:60], v["grid"]), collections.defaultdict(float))
    for k, x in v.items():
        if k != "grid":
            a[k] += x
    a["n"] += 1
print("%-52s %9s %3s %9s %5s %6s %6s %6s" % ("kernel", "grid", "n", "us@2.4GHz", "util", "VALU/M", "SALU/M", "LDS/M"))
for (short, grid), a in agg.items():
    if a["GRBM_GUI_ACTIVE"] > 0 and a["SQ_INSTS_MFMA"] > 0:
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        mf = a["SQ_INSTS_MFMA"]
        print("%-52s %9s %3d %9.1f %5.2f %6.2f %6.2f %6.2f" % (short[:52], grid, a["n"], cyc / a["n"] / 2400.0,
              a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), (a["SQ_INSTS_VALU"] - mf) / mf, a["SQ_INSTS_SALU"] / mf, a["SQ_INSTS_LDS"] / mf))
