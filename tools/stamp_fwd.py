#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the forward tile loop.
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
# optional shape: stamp_fwd.py [causal] b s h d   (4-wave blocks fill waves 0-3 of the table only)
dims = [int(x) for x in sys.argv[1:] if x.isdigit()]
b_, s_, h_, d_ = dims if len(dims) == 4 else (8, 4096, 32, 128)
q = torch.randn(b_, s_, h_, d_, device="cuda", dtype=torch.bfloat16)
k = torch.randn_like(q); v = torch.randn_like(q)
for _ in range(3):
    cabi.fwd(q, k, v, causal=causal)
torch.cuda.synchronize()
n = 512 * 8 * 8
buf = (C.c_ulonglong * n)()
lib.vfa_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.vfa_debug_read_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8).astype(np.float64)
names = ["issue_loads", "QK", "mask+softmax", "PV", "store+vmcnt", "barrier", "total_loop", "ntiles"]
tiles = a[:, :, 7]
sel = tiles > 0
print("blocks with data:", int(sel[:, 0].sum()))
for w in (0, 3, 4, 7):
    if not sel[:, w].any():
        continue
    per = [a[sel[:, w], w, i].sum() / tiles[sel[:, w], w].sum() for i in range(7)]
    print(f"wave {w}: " + "  ".join(f"{names[i]}={per[i]:.0f}" for i in range(7)))
