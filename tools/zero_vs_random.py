#!/usr/bin/env python3
"""Headline shape, forward + backward kernels timed on random and on all-zero inputs (development
tool): equal instruction streams, different data toggling -- a time difference is the clock the chip
holds under the kernel's power draw, not a property of the schedule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches
import bench

hip = bench.Hip()
lib = cabi.lib()
stream = torch.cuda.current_stream().cuda_stream
b, s, h, d, causal = 8, 4096, 32, 128, True
for name in ("random", "zeros", "random", "zeros"):
    mk = (lambda *sh: torch.randn(*sh, device="cuda", dtype=torch.bfloat16)) if name == "random" else \
         (lambda *sh: torch.zeros(*sh, device="cuda", dtype=torch.bfloat16))
    q, k, v, do = (mk(b, s, h, d) for _ in range(4))
    out = torch.empty_like(q)
    lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sd = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
    acc = torch.empty(1, device="cuda", dtype=torch.float32)
    Pf = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
    Pb = bench.build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, acc)
    for var in (1, 2):
        lib.vfa_debug_set_tuning(0, var)
        tf, t1, t2 = [], [], []
        for it in range(8):
            e0, e1 = hip.event(), hip.event()
            hip.record(e0, stream); cabi.call("vfa_mha_fwd", Pf); hip.record(e1, stream)
            ev = (C.c_void_p * 4)(*[hip.event() for _ in range(4)])
            Pb.stage_events = C.cast(ev, C.c_void_p)
            cabi.call("vfa_mha_bwd", Pb)
            torch.cuda.synchronize()
            if it >= 2:
                tf.append(hip.elapsed_ms(e0, e1))
                t1.append(hip.elapsed_ms(C.c_void_p(ev[1]), C.c_void_p(ev[2])))
                t2.append(hip.elapsed_ms(C.c_void_p(ev[2]), C.c_void_p(ev[3])))
        med = lambda xs: sorted(xs)[len(xs) // 2]
        print(f"{name:7s} dkdv-variant {var}: fwd {med(tf):.3f} ms  dkdv {med(t1):.3f} ms  dq {med(t2):.3f} ms", flush=True)
lib.vfa_debug_set_tuning(0, 0)
