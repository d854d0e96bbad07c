#!/usr/bin/env python3
"""Where is the GPU idle inside a step?  From a rocprofv3 --kernel-trace CSV (one row per dispatch with start / end
timestamps): per step (delimited by the k_roi_align_fwd launch that opens the mask head) the busy time, the idle time, and the
largest gaps with the kernels on either side.   python tools/gap_report.py kernel_trace.csv [--min-gap-us 20] [--top 25]"""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--min-gap-us", type=float, default=15.0)
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--marker", default="k_mask_fused_fwd|k_mask_losses|k_softmax_ce|k_ce_fwd", help="regex of a kernel that runs once per step")
args = ap.parse_args()
rows = []
for r in csv.DictReader(open(args.csv)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
marks = [i for i, r in enumerate(rows) if re.search(args.marker, r[2])]
# steps = marker to marker (the same phase of consecutive steps); skip the first two (warm-up)
bounds = marks[2:]
tot_busy = tot_idle = 0.0
gaps = []
for a, b in zip(bounds[:-1], bounds[1:]):
    seg = rows[a:b + 1]
    t0, t1 = seg[0][0], seg[-1][0]
    end = seg[0][1]
    busy = 0.0
    cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, n in seg[1:-1]:
        if s > cur_e:
            gaps.append(((s - cur_e) / 1e3, prev_n if False else None, n, cur_e))
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += min(cur_e, t1) - cur_s
    tot_busy += busy
    tot_idle += (t1 - t0) - busy
n = max(1, len(bounds) - 1)
print("steps analysed: %d; per step: span %.3f ms, busy %.3f ms, idle %.3f ms" % (n, (tot_busy + tot_idle) / n / 1e6, tot_busy / n / 1e6, tot_idle / n / 1e6))
# name the kernel before each gap
ends = {}
for s, e, nme in rows:
    ends[e] = nme
big = sorted([g for g in gaps if g[0] >= args.min_gap_us], reverse=True)
print("gaps >= %.0f us: %d per step, %.3f ms per step" % (args.min_gap_us, len(big) / n, sum(g[0] for g in big) / n / 1e3))
hist = {}
for us, _, nxt, e in big:
    key = (ends.get(e, "?")[:60], nxt[:60])
    c, t = hist.get(key, (0, 0.0))
    hist[key] = (c + 1, t + us)
for (prv, nxt), (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print("%7.1f us/step %5.1f x  after %-60s before %s" % (t / n, c / n, prv, nxt))
small = [g for g in gaps if g[0] < args.min_gap_us]
print("gaps < %.0f us: %d per step, %.3f ms per step (avg %.1f us)" % (args.min_gap_us, len(small) / n, sum(g[0] for g in small) / n / 1e3,
                                                                    sum(g[0] for g in small) / max(1, len(small))))
