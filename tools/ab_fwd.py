#!/usr/bin/env python3
"""Interleaved timing, in one process, of the forward kernel from several builds of the library on
random and on all-zero inputs (development tool):  ab_fwd.py lib1.so lib2.so ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import cabi
import bench

SHAPES = [(8, 4096, 32, 32, 128, True), (8, 4096, 32, 32, 128, False), (4, 8192, 32, 8, 128, True),
          (4, 2048, 16, 16, 64, True), (32, 512, 16, 16, 128, True), (16, 1024, 16, 16, 128, False),
          (8, 2048, 32, 32, 64, False), (32, 512, 32, 32, 64, True), (4, 4096, 32, 32, 64, True),   # 7, 8
          (4, 4096, 16, 16, 256, True), (4, 4096, 16, 16, 256, False), (16, 1024, 8, 8, 256, True),   # 9, 10, 11
          (16, 2048, 64, 8, 128, False), (16, 2048, 64, 8, 128, True),   # 12, 13: the shape of the programming guide's forward example
          (8, 4096, 32, 32, 96, True), (8, 4096, 32, 32, 96, False), (32, 512, 32, 32, 96, True), (8, 4096, 32, 32, 80, True),   # 14-17
          (4, 4096, 16, 16, 192, True), (4, 4096, 16, 16, 192, False), (4, 4096, 16, 16, 160, True), (4, 4096, 16, 16, 160, False),   # 18-21
          (16, 2048, 32, 32, 96, True), (32, 1024, 32, 32, 96, True), (16, 2048, 32, 32, 96, False), (4, 8192, 32, 32, 96, True),   # 22-25
          (1, 16384, 32, 32, 128, True), (2, 8192, 32, 32, 128, False), (1, 16384, 32, 32, 128, False), (16, 2048, 32, 32, 128, True)]   # 26-29


def main():
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    hip = bench.Hip()
    handles = [cabi.load(os.path.abspath(p)) for p in libs]
    # --tune-last=K:V applies a development switch to the last library only (load a COPY of the
    # library file to compare two settings of one build)
    for a in sys.argv[1:]:
        if a.startswith("--tune="):
            k, v = a.split("=", 1)[1].split(":")
            for h in handles:
                h.vfa_debug_set_tuning(int(k), int(v))
        if a.startswith("--tune-last="):
            k, v = a.split("=", 1)[1].split(":")
            handles[-1].vfa_debug_set_tuning(int(k), int(v))
    stream = torch.cuda.current_stream().cuda_stream
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    datas = ("random",) if "--random-only" in sys.argv else ("random", "zeros")
    for si, (b, s, h, hk, d, causal) in enumerate(SHAPES):
        if only and si not in only:
            continue
        for data in datas:
            torch.manual_seed(0)
            mk = torch.randn if data == "random" else torch.zeros
            q = mk(b, s, h, d, device="cuda", dtype=torch.bfloat16)
            k = mk(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
            v = mk(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
            out = torch.empty_like(q)
            lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
            P = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
            times = [[] for _ in handles]
            outs = []
            for hd in handles:   # the builds must agree bit for bit
                out.fill_(float("nan"))
                assert hd.vfa_mha_fwd(C.byref(P)) == 0
                outs.append(out.clone())
            if not all(torch.equal(outs[0], o) for o in outs[1:]):
                print("   OUTPUTS DIFFER, max", max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:]), flush=True)
            for rnd in range(14):
                for i, hd in enumerate(handles):
                    e0, e1 = hip.event(), hip.event()
                    hip.record(e0, stream)
                    assert hd.vfa_mha_fwd(C.byref(P)) == 0
                    hip.record(e1, stream)
                    torch.cuda.synchronize()
                    if rnd >= 4:
                        times[i].append(hip.elapsed_ms(e0, e1))
            flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
            for i, p in enumerate(libs):
                t = sorted(times[i])[len(times[i]) // 2]
                print(f"b{b} s{s} h{h}/{hk} d{d} causal={causal} {data:6s} {os.path.basename(p):26s} {t:.3f} ms {flops / t / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
