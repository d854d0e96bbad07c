#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from one rocprofv3 counter pass over tools/bench_layers.py:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o p -- \\
        python tools/bench_layers.py --iters 1
    python tools/pmc_mfma_table.py DIR/.../p_counter_collection.csv > profiles/roundN_pmc_mfma_all_kernels.txt

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): GRBM_GUI_ACTIVE is summed over the 8 XCDs, the MFMA
counter over the 1024 SIMDs (tools/pmc_mfma.sh has the derivation and the cross-check against the instruction count)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {"grid": r["Grid_Size"], "vgpr": r["VGPR_Count"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for (_, name), v in disp.items():
    if "mfma" not in name and "wgrad" not in name and "wino" not in name:
        continue
    m = re.search(r"(k_\w+<[^>]*>)", name)
    a = agg.setdefault((m.group(1) if m else name[:60], v["grid"], v["vgpr"]), [0.0, 0.0, 0])
    a[0] += v.get("GRBM_GUI_ACTIVE", 0.0)
    a[1] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[2] += 1
print("%-50s %9s %5s %3s %9s %6s" % ("kernel", "grid", "vgpr", "n", "us@2.4GHz", "util"))
for (short, grid, vgpr), (act, busy, n) in agg.items():
    if act > 0:
        cyc = act / 8.0
        print("%-50s %9s %5s %3d %9.1f %6.2f" % (short[:50], grid, vgpr, n, cyc / n / 2400.0, busy / (cyc * 1024.0)))
