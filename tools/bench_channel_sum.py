#!/usr/bin/env python3
"""cfun_channel_sum (bias-gradient sums): the one-launch kernel against reduce + finalize over tensor sizes, and the (rows, channels)
shapes one bench step asks for.   python tools/bench_channel_sum.py [--shapes-of-step]
Each size runs in a subprocess per CFUN_SUM_DIRECT_LOG2 setting (the knob is read once per process)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--shapes-of-step", action="store_true")
ap.add_argument("--child", action="store_true")
args = ap.parse_args()

SIZES = [(v, c) for c in (16, 64, 256) for v in (256, 1024, 4096, 16384, 65536, 262144) if v * c <= (1 << 24)]

if args.child:
    import torch
    from cfun_amd import ops
    dev = torch.device("cuda:0")
    for v, c in SIZES:
        g = torch.randn(v, c, device=dev)
        for _ in range(5):
            ops.channel_sum(g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.channel_sum(g)
        e1.record()
        torch.cuda.synchronize()
        print("%d %d %.2f" % (v, c, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
    sys.exit(0)

if args.shapes_of_step:
    import collections
    import bench
    from cfun_amd import ops
    seen = collections.Counter()
    orig = ops.channel_sum

    def logged(g2d):
        seen[tuple(g2d.shape)] += 1
        return orig(g2d)

    ops.channel_sum = logged
    sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-loop"]
    bench.main()
    print("# channel_sum shapes over 2 steps (+ the parity step): rows x channels : calls")
    for (v, c), k in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1]):
        print("%9d x %4d = %9d elements : %d" % (v, c, v * c, k))
    sys.exit(0)

res = {}
for tag, log2 in (("direct", 40), ("two_launch", 0)):
    env = dict(os.environ, CFUN_SUM_DIRECT_LOG2=str(log2))
    out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        f = line.split()
        if len(f) == 3:
            res[(int(f[0]), int(f[1]), tag)] = float(f[2])
print("%10s %6s %12s %14s %16s" % ("rows", "chan", "elements", "one launch us", "reduce+final us"))
for v, c in SIZES:
    print("%10d %6d %12d %14.2f %16.2f" % (v, c, v * c, res.get((v, c, "direct"), -1), res.get((v, c, "two_launch"), -1)))
