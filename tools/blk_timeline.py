#!/usr/bin/env python3
"""Diagnostic: when the workgroups of a dK/dV launch enter, start and end their step loop and leave
(100 MHz wall clock), from the stamped build (python vllm-flash-attention_amd/build.py --stamp):
blk_timeline.py [causal] [dq|fwd] b s h d"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
cabi.LIB_PATH = os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip_stamp.so")
lib = cabi.lib()
causal = "causal" in sys.argv[1:]
which = "dq" if "dq" in sys.argv[1:] else ("fwd" if "fwd" in sys.argv[1:] else "dkdv")   # which kernel's stamps
dims = [int(x) for x in sys.argv[1:] if x.isdigit()]
b, s, h, d = dims if len(dims) == 4 else (32, 512, 16, 128)
q, k, v, do = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(4))
out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
for _ in range(3):
    cabi.bwd(do, q, k, v, out, lse, causal=causal)
torch.cuda.synchronize()
n = 4096 * 8
buf = (C.c_ulonglong * n)()
reader = getattr(lib, "vfa_debug_read_blk_" + which)
reader.argtypes = [C.c_void_p, C.c_size_t]
assert reader(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
a = (a - t0) / 100.0   # microseconds
for nm, x in (("entry -> block mapped", a[:, 4] - a[:, 0]), ("-> first slice requested", a[:, 5] - a[:, 4]),
              ("-> own loads landed", a[:, 6] - a[:, 5]), ("-> loop (barrier)", a[:, 1] - a[:, 6])):
    print(f"  prologue: {nm:26s} mean {x.mean():6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f} us")
print(f"{which} kernel, b{b} s{s} h{h} d{d} causal={causal}: {len(a)} workgroups, launch span {a[:, 3].max():.1f} us")
pro, loop, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
for nm, x in (("prologue", pro), ("loop", loop), ("epilogue", epi)):
    print(f"  {nm:9s} mean {x.mean():6.2f}  median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f} us")
print(f"  sum of workgroup lives / (span x 256 CUs) = {(a[:, 3] - a[:, 0]).sum() / (a[:, 3].max() * 256):.2f}")
# entry times in rounds: how many workgroups enter within each 5 us bin
bins = np.arange(0, a[:, 3].max() + 5, 5)
hist_in = np.histogram(a[:, 0], bins)[0]
hist_out = np.histogram(a[:, 3], bins)[0]
print("  5-us bins, workgroups entering:", " ".join(str(x) for x in hist_in))
print("  5-us bins, workgroups leaving: ", " ".join(str(x) for x in hist_out))
