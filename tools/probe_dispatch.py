#!/usr/bin/env python3
"""Where does the hardware dispatcher put the workgroups of a launch?  (MI355X probe, not part of the product.)

    python tools/probe_dispatch.py [nblocks] [lds_bytes] [spin]

Prints, per XCD, how many workgroups each CU received and how many ran concurrently, for a 256-thread kernel
holding `lds_bytes` of LDS -- the facts behind the launch geometry of the wgrad / split-K kernels."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import _lib  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "bin", "libcfun_probe.so"))   # make -C tools/probes
lib.cfun_debug_dispatch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lds = int(sys.argv[2]) if len(sys.argv) > 2 else 52 * 1024
spin = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
out = torch.zeros(nblocks * 4, dtype=torch.int64, device="cuda")
for _ in range(2):
    rc = lib.cfun_debug_dispatch(nblocks, lds, spin, out.data_ptr(), None)
    assert rc == 0, rc
torch.cuda.synchronize()
o = out.cpu().view(nblocks, 4).tolist()
t_min = min(r[2] for r in o)
per_cu = collections.Counter()
for b, (hw, xcc, t0, t1) in enumerate(o):
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    per_cu[(xcc & 15, se, sh, cu)] += 1
print(f"{nblocks} blocks, lds {lds} B: {len(per_cu)} distinct CUs used")
print("wave slot (HW_ID WAVE_ID) of wave 0, histogram:", dict(sorted(collections.Counter(r[0] & 15 for r in o).items())), " simd:", dict(sorted(collections.Counter((r[0] >> 4) & 3 for r in o).items())))
hist = collections.Counter(per_cu.values())
print("blocks per CU histogram:", dict(sorted(hist.items())))
for x in range(8):
    cus = {k: v for k, v in per_cu.items() if k[0] == x}
    print(f"  xcd {x}: {len(cus)} CUs, blocks {sum(cus.values())}, per-CU {sorted(cus.values())}")
dur = [(r[3] - r[2]) / 100.0 for r in o]
start = [(r[2] - t_min) / 100.0 for r in o]
print(f"block duration us: min {min(dur):.1f} max {max(dur):.1f}; last start {max(start):.1f} us; "
      f"makespan {max(s + d for s, d in zip(start, dur)):.1f} us")
print("first 24 blocks (xcc,se,sh,cu):", [((r[1] & 15), (r[0] >> 13) & 7, (r[0] >> 12) & 1, (r[0] >> 8) & 15) for r in o[:24]])
