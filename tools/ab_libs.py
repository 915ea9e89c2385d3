#!/usr/bin/env python3
"""Interleaved timing, in one process, of the backward (dK/dV and dQ kernels) from several builds of
the library (development tool):  ab_libs.py lib1.so lib2.so ... [-- shape indices] [--tune K=V]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import cabi
import bench

SHAPES = [(8, 4096, 32, 32, 128, True), (8, 4096, 32, 32, 128, False), (4, 4096, 32, 8, 128, True),
          (16, 1024, 16, 16, 128, True), (4, 2048, 16, 16, 64, True),
          (4, 4096, 32, 32, 64, False), (4, 4096, 32, 32, 64, True), (16, 1024, 32, 32, 64, True),   # 5, 6, 7
          (32, 512, 16, 16, 128, True), (32, 512, 32, 32, 64, False),                                 # 8, 9
          (4, 4096, 16, 16, 256, True), (4, 4096, 16, 16, 256, False),                                # 10, 11
          (2, 8192, 32, 32, 128, True), (1, 16384, 32, 32, 128, True), (2, 16384, 16, 16, 128, True),  # 12, 13, 14
          (8, 4096, 32, 32, 96, True), (8, 4096, 32, 32, 96, False), (32, 512, 32, 32, 96, True), (8, 4096, 32, 32, 80, True),  # 15-18
          (4, 4096, 16, 16, 192, True), (4, 4096, 16, 16, 192, False), (4, 4096, 16, 16, 160, True), (4, 4096, 16, 16, 160, False)]  # 19-22


def main():
    args = sys.argv[1:]
    kv = lambda a: tuple(int(x) for x in a.split("=", 1)[1].split(":"))
    tune = [kv(a) for a in args if a.startswith("--tune=")]
    tune_last = [kv(a) for a in args if a.startswith("--tune-last=")]   # the last library only
    args = [a for a in args if not a.startswith("--tune")]
    libs = [a for a in args if a.endswith(".so")]
    idx = [int(a) for a in args if a.isdigit()] or [0, 1]
    hip = bench.Hip()
    cabi.lib()
    handles = []
    for path in libs:
        h = C.CDLL(os.path.abspath(path))
        h.vfa_mha_bwd.argtypes = [C.POINTER(cabi.VfaParams)]
        for k, v in tune:
            h.vfa_debug_set_tuning(k, v)
        handles.append(h)
    for k, v in tune_last:
        handles[-1].vfa_debug_set_tuning(k, v)
    stream = torch.cuda.current_stream().cuda_stream
    for si in idx:
        b, s, h, hk, d, causal = SHAPES[si]
        torch.manual_seed(0)
        q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
        v = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
        do = torch.randn_like(q)
        out = torch.empty_like(q)
        lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sd = torch.empty(b, h, (s + 127) // 128 * 128, device="cuda", dtype=torch.float32)
        acc = torch.empty(1, device="cuda", dtype=torch.float32)
        cabi.call("vfa_mha_fwd", bench.build_params(cabi, q, k, v, out, lse, causal, stream))
        Pb = bench.build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, acc)
        times = [[] for _ in handles]
        grads = []
        for hd in handles:   # the builds must agree bit for bit
            Pb.stage_events = None
            for t in (dq, dk, dv):
                t.fill_(float("nan"))
            assert hd.vfa_mha_bwd(C.byref(Pb)) == 0
            grads.append((dq.clone(), dk.clone(), dv.clone()))
        if not all(all(torch.equal(a, b_) for a, b_ in zip(grads[0], g)) for g in grads[1:]):
            print("   GRADIENTS DIFFER", flush=True)
        for rnd in range(9):
            for i, hd in enumerate(handles):
                ev = (C.c_void_p * 4)(*[hip.event() for _ in range(4)])
                Pb.stage_events = C.cast(ev, C.c_void_p)
                assert hd.vfa_mha_bwd(C.byref(Pb)) == 0
                torch.cuda.synchronize()
                if rnd >= 2:
                    t = [hip.elapsed_ms(C.c_void_p(ev[j]), C.c_void_p(ev[j + 1])) for j in range(3)]
                    times[i].append(t + [hip.elapsed_ms(C.c_void_p(ev[0]), C.c_void_p(ev[3]))])
        med = lambda xs: sorted(xs)[len(xs) // 2]
        print(f"b{b} s{s} h{h}/{hk} d{d} causal={causal}:", flush=True)
        for i, path in enumerate(libs):
            n1, n2 = "dkdv", "dq"
            m = [med([x[j] for x in times[i]]) for j in range(4)]
            print(f"   {os.path.basename(path):28s} pre {m[0]:.3f}   {n1} {m[1]:.3f}   {n2} {m[2]:.3f}   all {m[3]:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
