#!/bin/bash
# MFMA-pipe utilisation of the dominant conv launch (conv_norm_lrelu_l4.0) from the SQ / GRBM counters
# (MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (busy cycles * 256 CUs * 4 SIMDs); GRBM_GUI_ACTIVE comes back summed over
# the 8 XCDs, so busy cycles = GRBM_GUI_ACTIVE / 8 -- it then equals kernel time x 2.4 GHz), one rocprofv3 --pmc pass
# with --kernel-trace only.  Prints the JSON committed as profiles/roundN_pmc_mfma_wino_l4_0.json.
#   usage (on the GPU box, from the repo root):  bash tools/pmc_mfma.sh
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_mfma -o p -- \
    python "$REPO/tools/bench_layers.py" --filter l4.0 --iters 1 > "$REPO/gpurun_out/pmc_mfma.log" 2>&1
cp "$(find /tmp/pmc_mfma -name '*counter_collection.csv' | head -1)" "$REPO/gpurun_out/pmc_mfma.csv"
REPO="$REPO" python - "$REPO/gpurun_out/pmc_mfma.csv" <<'PY'
import csv, json, os, re, sys
sys.path.insert(0, os.path.join(os.environ["REPO"], "tools"))
from pmc_traffic import git_blob_sha1, SRC, KERNEL, kernel_label
acc = {}
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(KERNEL, row["Kernel_Name"]):
        acc.setdefault(kernel_label(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
per = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
avg = {}
for d in per.values():            # one conv call = these kernels one after the other: counters add
    for c, v in d.items():
        avg[c] = avg.get(c, 0.0) + v
cus = 256
rec = {"what": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES (one pass, --kernel-trace only) "
               "around tools/bench_layers.py --filter l4.0 --iters 1 on MI355X (tools/pmc_mfma.sh)",
       "kernel": "%s (3x3x3 40->40 @ 4x96^3, forward / data gradient; counters summed over the call's kernels)" % " + ".join(sorted(per)),
       "kernel_src": "cfun_amd/csrc/" + os.path.basename(SRC), "kernel_src_blob": git_blob_sha1(SRC),
       "dispatches": min(len(next(iter(d.values()))) for d in acc.values()) if acc else 0, "counters_avg": avg,
       "per_kernel": {k: dict(d, mfma_util=d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * cus * 4))
                      for k, d in per.items() if d.get("GRBM_GUI_ACTIVE")}}
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
    busy = avg["GRBM_GUI_ACTIVE"] / 8.0            # the counter is reported summed over the 8 XCDs
    rec["busy_cycles_per_xcd"] = busy
    rec["mfma_util"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * cus * 4)
    rec["formula"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)"
print(json.dumps(rec, indent=2))
PY
