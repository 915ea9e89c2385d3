#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the step loop of the 64-keys-per-wave dK/dV kernel.
Build the stamped library first: python vllm-flash-attention_amd/build.py --stamp"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
cabi.LIB_PATH = os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip_stamp.so")
lib = cabi.lib()
lib.vfa_debug_set_tuning(0, 2)
causal = len(sys.argv) > 1 and sys.argv[1] == "causal"
q = torch.randn(8, 4096, 32, 128, device="cuda", dtype=torch.bfloat16)
k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
for _ in range(2):
    cabi.bwd(do, q, k, v, out, lse, causal=causal)
torch.cuda.synchronize()
n = 512 * 4 * 8
buf = (C.c_ulonglong * n)()
lib.vfa_debug_read_stamps_w64.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vfa_debug_read_stamps_w64(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 8).astype(np.float64)
names = ["issue_loads", "S(x2)", "mask(x2)", "dP+exp(x2)", "pack+dV+dK(x2)", "vmcnt_wait", "barrier"]
steps = a[:, :, 7]
sel = steps > 0
print("blocks with data:", int(sel[:, 0].sum()), "(cycles per 64-row step; MFMA floor 4096)")
for w in range(4):
    per = [a[sel[:, w], w, i].sum() / steps[sel[:, w], w].sum() for i in range(7)]
    print(f"wave {w}: " + "  ".join(f"{names[i]}={per[i]:.0f}" for i in range(7)) + f"  total={sum(per):.0f}")
