#!/usr/bin/env python3
"""Whole-step HBM traffic from two rocprofv3 --pmc passes over bench.py (tools/pmc_step.sh).

  --compact in.csv out.csv : reduce rocprofv3's per-dispatch counter CSV to (kernel, calls, sum KB) rows (what is committed)
  --steps N fetch.csv write.csv : per kernel-family table, bytes per step (N = warm-up + timed steps of the profiled run)

FETCH_SIZE is doubled (gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md section HBM),
WRITE_SIZE is taken as reported; both counters are in KB and count the L2's fabric-side requests (Infinity-Cache hits
included), so the sums are an UPPER bound on DRAM bytes."""
import csv
import re
import sys
from collections import OrderedDict

FAMILIES = [
    ("conv fwd / dgrad: Winograd (k_conv_wino)", r"k_conv_wino"),
    ("conv fwd / dgrad: direct MFMA (k_conv_mfma)", r"k_conv_mfma"),
    ("weight gradients (k_wgrad_*)", r"k_wgrad"),
    ("stem / pointwise / direct VALU convs", r"k_conv_stem|k_conv_pointwise|k_conv_.*direct"),
    ("split-K finish, partial reductions, weight packs / transforms", r"splitk|reduce_partials|reduce_unpack|transpose_pad|k_wino_weights|weight_pack|k_fold|k_pack|k_unpack"),
    ("InstanceNorm / LeakyReLU / channel reductions", r"instnorm|channel_reduce|channel_finalize|lrelu|k_act_bwd|k_norm"),
    ("mask losses (softmax, CE, edge)", r"softmax|k_ce_|k_edge|loss|k_mask_fused|k_finalize_sum"),
    ("RoIAlign / NMS / classifier / targets / resize / pool / upsample", r"roi_align|nms|k_fc_|mask_target|resize|maxpool|upsample|k_add|halo"),
    ("torch glue (at::native, copies, fills)", r"at::native|rocclr|Memset|fill"),
]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:110]


def compact(src, dst):
    acc = OrderedDict()
    with open(src) as f:
        for row in csv.DictReader(f):
            k = (short(row["Kernel_Name"]), row["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Sum_KB"])
        for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, n, "%.1f" % s])


def load(path):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out[row["Kernel"]] = (int(row["Dispatches"]), float(row["Sum_KB"]) * 1024.0)
    return out


def table(steps, fetch_csv, write_csv):
    fe, wr = load(fetch_csv), load(write_csv)
    fam = OrderedDict((n, [0, 0.0, 0.0]) for n, _ in FAMILIES)
    fam["other"] = [0, 0.0, 0.0]
    for k in sorted(set(fe) | set(wr)):
        n, fb = fe.get(k, (0, 0.0))
        _, wb = wr.get(k, (0, 0.0))
        for name, pat in FAMILIES:
            if re.search(pat, k):
                break
        else:
            name = "other"
        fam[name][0] += n
        fam[name][1] += 2.0 * fb      # gfx950 correction
        fam[name][2] += wb
    print("HBM-side traffic per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; %d steps profiled;"
          " FETCH x2 per MI355X_MICROARCH.md)" % steps)
    print("%-68s %9s %10s %10s %10s" % ("kernel family", "launches", "read GB", "write GB", "total GB"))
    tot = [0, 0.0, 0.0]
    for name, (n, r, w) in fam.items():
        print("%-68s %9.0f %10.3f %10.3f %10.3f" % (name, n / steps, r / steps / 1e9, w / steps / 1e9, (r + w) / steps / 1e9))
        tot[0] += n
        tot[1] += r
        tot[2] += w
    print("%-68s %9.0f %10.3f %10.3f %10.3f" % ("TOTAL", tot[0] / steps, tot[1] / steps / 1e9, tot[2] / steps / 1e9,
                                                 (tot[1] + tot[2]) / steps / 1e9))
    print("algorithmic (SURVEY.md section 8(d), each tensor once, dense algorithm): ~30 GB/step")


if __name__ == "__main__":
    if sys.argv[1] == "--compact":
        compact(sys.argv[2], sys.argv[3])
    else:
        table(int(sys.argv[2]), sys.argv[3], sys.argv[4])
