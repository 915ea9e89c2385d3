#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the dK/dV step loop.
Build the stamped library first: python vllm-flash-attention_amd/build.py --stamp"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
cabi.LIB_PATH = os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip_stamp.so")
lib = cabi.lib()
causal = len(sys.argv) > 1 and sys.argv[1] == "causal"
pp = "pp" in sys.argv[1:]   # two-phase kernel: stamps per 32-row sub-block
if pp:
    lib.vfa_debug_set_tuning.argtypes = [C.c_int, C.c_int]
    lib.vfa_debug_set_tuning(0, 3)
q = torch.randn(8, 4096, 32, 128, device="cuda", dtype=torch.bfloat16)
k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
for _ in range(2):
    cabi.bwd(do, q, k, v, out, lse, causal=causal)
torch.cuda.synchronize()
n = 512 * 8 * 8
buf = (C.c_ulonglong * n)()
lib.vfa_debug_read_stamps_dkdv.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vfa_debug_read_stamps_dkdv(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8).astype(np.float64)
names = ["issue_loads", "S=QK", "mask+exp", "dP=dOV", "pack+dV+dK", "store", "barrier"]
if pp:
    names = ["issue", "V:exp+dS", "wait+barrier(V)", "M:dV+dK", "M:S+dP", "wait+barrier(M)", "-"]
steps = a[:, :, 7]
sel = steps > 0
print("blocks with data:", int(sel[:, 0].sum()))
for w in (0, 3, 4, 7):
    per = [a[sel[:, w], w, i].sum() / steps[sel[:, w], w].sum() for i in range(7)]
    print(f"wave {w}: " + "  ".join(f"{names[i]}={per[i]:.0f}" for i in range(7)) + f"  total={sum(per):.0f}")
