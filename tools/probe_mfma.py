#!/usr/bin/env python3
"""Dump the operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on the GPU (facts for kernel design)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import _lib
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "bin", "libcfun_probe.so"))   # make -C tools/probes
lib.cfun_debug_mfma_4x4x1.argtypes = [ctypes.c_void_p] * 4
a = torch.arange(1, 65, dtype=torch.float32, device="cuda")            # a[l] = l + 1
b = torch.arange(0, 64, dtype=torch.float32, device="cuda") + 1000.0   # b[l] = 1000 + l
d = torch.zeros(256, device="cuda")
rc = lib.cfun_debug_mfma_4x4x1(a.data_ptr(), b.data_ptr(), d.data_ptr(), None)
torch.cuda.synchronize()
d = d.cpu().view(64, 4)
print("rc", rc)
for lane in range(64):
    row = []
    for r in range(4):
        v = float(d[lane, r]); found = None
        for la in range(64):
            for lb in range(64):
                if (la + 1) * (1000 + lb) == v:
                    found = (la, lb)
        row.append(found)
    if lane < 12 or lane % 16 == 0:
        print("lane %2d:" % lane, " ".join("reg%d=A[l%s]*B[l%s]" % (r, f[0], f[1]) if f else "reg%d=?(%g)" % (r, float(d[lane, r])) for r, f in enumerate(row)))
