#!/usr/bin/env python3
"""8-wave against 4-wave forward workgroups over sequence lengths (development switch 6),
interleaved in one process (development tool): ab_fwd_waves.py [head dim, default 128]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches

lib = cabi.lib()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for causal in (True, False):
    for s in (256, 512, 1024, 2048, 3072, 4096):
        b, h, d = 16384 // s, 2048 // D, D
        q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        times = {1: [], 2: []}
        for rnd in range(12):
            for sw in (1, 2):
                lib.vfa_debug_set_tuning(6, sw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cabi.fwd(q, k, v, causal=causal)
                e0.record()
                for _ in range(5):
                    cabi.fwd(q, k, v, causal=causal)
                e1.record(); torch.cuda.synchronize()
                if rnd >= 2:
                    times[sw].append(e0.elapsed_time(e1) / 5)
        lib.vfa_debug_set_tuning(6, 0)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        fl = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
        print(f"s{s} causal={causal}: 8 waves {fl / med(times[1]) / 1e9:.0f} TF   4 waves {fl / med(times[2]) / 1e9:.0f} TF", flush=True)
