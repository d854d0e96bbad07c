#!/usr/bin/env python3
"""Two-stream critical path of a step, from a rocprofv3 --kernel-trace CSV: per step (delimited by the mask-loss kernel) the time
during which only one queue, both queues or no queue has a kernel running, and per queue the kernels that ran ALONE (their time is
step time; a kernel running beside the other queue is overlapped work).
   python tools/stream_report.py kernel_trace.csv [--top 25] [--dump steps.csv.gz]"""
import argparse
import csv
import gzip
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--marker", default="k_mask_fused_fwd|k_mask_losses|k_softmax_ce|k_ce_fwd")
ap.add_argument("--dump", default=None, help="write the analysed rows (queue, start, end, name) here")
args = ap.parse_args()
opener = gzip.open if args.csv.endswith(".gz") else open
rows = []
with opener(args.csv, "rt") as f:
    raw = list(csv.DictReader(f))
# the column that tells the two HIP streams apart: Stream_Id where rocprofv3 fills it, else the HSA queue
qcol = next((c for c in ("Stream_Id", "Queue_Id") if c in raw[0] and len({r[c] for r in raw}) > 1), "Queue_Id")
for r in raw:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get(qcol, "0")))
del raw
rows.sort()
marks = [i for i, r in enumerate(rows) if re.search(args.marker, r[2])]
bounds = marks[2:]
nsteps = max(1, len(bounds) - 1)
seg_rows = rows[bounds[0]:bounds[-1]] if len(bounds) > 1 else rows
if args.dump:
    with gzip.open(args.dump, "wt") as f:
        w = csv.writer(f)
        w.writerow(["Queue_Id", "Start_Timestamp", "End_Timestamp", "Kernel_Name"])
        for s, e, n, q in seg_rows:
            w.writerow([q, s, e, n])
queues = sorted({r[3] for r in seg_rows}, key=lambda q: -sum(e - s for s, e, n, qq in seg_rows if qq == q))
print("steps analysed: %d, span per step %.3f ms, queues by kernel time: %s" % (
    nsteps, (seg_rows[-1][0] - seg_rows[0][0]) / nsteps / 1e6,
    ", ".join("%s %.2f ms" % (q, sum(e - s for s, e, n, qq in seg_rows if qq == q) / nsteps / 1e6) for q in queues)))
# sweep: events per queue -> number of running kernels per queue over time
ev = []
for s, e, n, q in seg_rows:
    ev.append((s, 1, q, n))
    ev.append((e, -1, q, n))
ev.sort(key=lambda x: (x[0], x[1]))
run = {q: 0 for q in queues}
state_time = {}
alone = {}            # (queue, kernel) -> time it ran while no other queue had a kernel running
active = {}           # queue -> name of the running kernel (kernels of one queue do not overlap much)
last = ev[0][0]
for t, d, q, n in ev:
    dt = t - last
    if dt > 0:
        on = tuple(qq for qq in queues if run[qq] > 0)
        state_time[on] = state_time.get(on, 0) + dt
        if len(on) == 1:
            k = (on[0], active.get(on[0], "?"))
            alone[k] = alone.get(k, 0) + dt
    last = t
    run[q] += d
    if d > 0:
        active[q] = n
for on, t in sorted(state_time.items(), key=lambda kv: -kv[1]):
    print("  %-40s %8.3f ms per step" % ("+".join(on) if on else "(idle)", t / nsteps / 1e6))
for q in queues:
    ks = sorted(((t, n) for (qq, n), t in alone.items() if qq == q), reverse=True)
    print("queue %s alone: %.3f ms per step; top kernels:" % (q, sum(t for t, n in ks) / nsteps / 1e6))
    for t, n in ks[:args.top]:
        print("   %8.1f us/step  %s" % (t / nsteps / 1e3, n[:110]))
