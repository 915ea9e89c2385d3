#!/usr/bin/env python3
"""Phase times of the 64-rows-per-wave forward kernel's steady iterations, from a -DVFA_STAMP build
(development tool): shader cycles per iteration, averaged over the first 512 workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import numpy as np
import torch
import cabi
import bench

lib = cabi.load(os.path.join(ROOT, "vllm-flash-attention_amd", "libvfa_hip_stamp.so"))
cabi._lib = lib
lib.vfa_debug_set_tuning(4, 1)
stream = torch.cuda.current_stream().cuda_stream
names = ["phase 1 (S | exp)", "block A (PV | max)", "decision", "block B (PV | exp)", "rescale", "wait loads", "barrier"]
for causal in (False, True):
    torch.manual_seed(0)
    b, s, h, d = 8, 4096, 32, 128
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
    P = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
    for _ in range(3):
        assert lib.vfa_mha_fwd(C.byref(P)) == 0
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 4 * 8))()
    assert lib.vfa_debug_read_stamps_f64(buf, 512 * 4 * 8) == 0
    a = np.array(buf, dtype=np.float64).reshape(512, 4, 8)
    n = a[:, :, 7].sum()
    print(f"causal={causal}: {int(n)} steady iterations stamped")
    tot = 0.0
    for i, nm in enumerate(names):
        c = a[:, :, i].sum() / n
        tot += c
        print(f"   {nm:22s} {c:8.0f} cycles / iteration")
    print(f"   {'total':22s} {tot:8.0f}   (64 MFMAs = 2048)")
    per_wave = a[:, :, :7].sum(axis=(0, 2)) / np.maximum(a[:, :, 7].sum(axis=0), 1)
    print("   per wave:", [int(x) for x in per_wave])
