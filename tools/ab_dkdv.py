#!/usr/bin/env python3
"""Interleaved A/B timing, in one process, of the two dK/dV kernels (8 waves x 32 keys vs 4 waves x
64 keys) through vfa_debug_set_tuning (development tool).  Prints the median dK/dV and dQ kernel
times per shape and variant, and checks that both variants give the same dk / dv to 1 ulp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import cabi
import bench


def main():
    for a in list(sys.argv[1:]):   # --lib=path: another build of the library
        if a.startswith("--lib="):
            cabi.LIB_PATH = os.path.abspath(a.split("=", 1)[1])
            sys.argv.remove(a)
    hip = bench.Hip()
    lib = cabi.lib()
    torch.manual_seed(0)
    shapes = [(8, 4096, 32, 32, 128, True, "bf16"), (8, 4096, 32, 32, 128, False, "bf16"),
              (4, 4096, 32, 8, 128, True, "bf16"), (2, 8192, 16, 16, 128, True, "bf16"),
              (16, 1024, 16, 16, 128, True, "bf16"), (32, 512, 16, 16, 128, True, "bf16"),
              (4, 2048, 16, 16, 64, True, "bf16"), (8, 4096, 32, 32, 64, True, "bf16"),
              (4, 4096, 16, 16, 256, True, "bf16"), (4, 4096, 16, 16, 256, False, "bf16"),
              (4, 4096, 16, 16, 192, True, "bf16"),
              (32, 512, 16, 16, 128, False, "bf16"), (16, 1024, 16, 16, 128, False, "bf16"),     # 11, 12
              (8, 2048, 16, 16, 128, True, "bf16"), (8, 2048, 16, 16, 128, False, "bf16"),       # 13, 14
              (32, 512, 32, 32, 64, True, "bf16"), (16, 1024, 32, 32, 64, True, "bf16"),         # 15, 16
              (32, 512, 32, 32, 64, False, "bf16"), (8, 2048, 32, 32, 64, True, "bf16"),         # 17, 18
              (2, 8192, 32, 32, 64, True, "bf16"), (1, 16384, 32, 32, 64, True, "bf16"),         # 19, 20
              (4, 4096, 32, 32, 64, False, "bf16"), (8, 2048, 32, 32, 64, False, "bf16"),        # 21, 22
              (1, 16384, 16, 16, 128, True, "bf16"), (4, 6144, 16, 16, 128, True, "bf16")]       # 23, 24
    variants = [1, 2]   # development switch 0: 1 = 8 waves x 32 keys, 2 = 4 waves x 64 keys, 3 = two-phase 8 waves, 5 = 4 waves x 32 keys
    args = sys.argv[1:]
    zeros = "--zeros" in args   # all-zero inputs: same instruction stream, no data-dependent power
    if zeros:
        args.remove("--zeros")
    for a in list(args):
        if a.startswith("--variants="):
            variants = [int(x) for x in a.split("=")[1].split(",")]
            args.remove(a)
    switch = 0
    for a in list(args):
        if a.startswith("--switch="):   # which development switch the variants are values of
            switch = int(a.split("=")[1])
            args.remove(a)
    if args:
        shapes = [shapes[int(a)] for a in args]
    stream = torch.cuda.current_stream().cuda_stream
    for (b, s, h, hk, d, causal, dt) in shapes:
        tdt = torch.bfloat16 if dt == "bf16" else torch.float16
        q = torch.randn(b, s, h, d, device="cuda", dtype=tdt)
        k = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        v = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        do = torch.randn_like(q)
        if zeros:
            q.zero_(); k.zero_(); v.zero_(); do.zero_()
        out = torch.empty_like(q)
        lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
        dq = torch.empty_like(q)
        sd = torch.empty(b, h, (s + 127) // 128 * 128, device="cuda", dtype=torch.float32)
        acc = torch.empty(1, device="cuda", dtype=torch.float32)
        Pf = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
        cabi.call("vfa_mha_fwd", Pf)
        res, times = {}, {}
        for var in variants:
            dk, dv = torch.empty_like(k), torch.empty_like(v)
            res[var] = (dk, dv, bench.build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, acc))
            times[var] = []
        n = 7
        for rnd in range(n + 2):
            for var in variants:
                lib.vfa_debug_set_tuning(switch, var)
                Pb = res[var][2]
                ev = (C.c_void_p * 4)(*[hip.event() for _ in range(4)])
                Pb.stage_events = C.cast(ev, C.c_void_p)
                cabi.call("vfa_mha_bwd", Pb)
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[var].append((hip.elapsed_ms(C.c_void_p(ev[1]), C.c_void_p(ev[2])),
                                       hip.elapsed_ms(C.c_void_p(ev[2]), C.c_void_p(ev[3]))))
        lib.vfa_debug_set_tuning(switch, 0)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        f = bench.fwd_flops(b, s, s, h, d, causal)
        line = f"b{b} s{s} h{h}/{hk} d{d} causal={causal}:"
        for var in variants:
            t = med([x[0] for x in times[var]])
            t2 = med([x[1] for x in times[var]])
            line += f"  v{var} dkdv {t:.3f} ms ({2 * f / t / 1e9:.0f} TF exec) dq {t2:.3f} ms"
        dk1, dv1 = res[variants[0]][0].float(), res[variants[0]][1].float()
        dk2, dv2 = res[variants[-1]][0].float(), res[variants[-1]][1].float()
        ek = (dk1 - dk2).abs().max().item() / max(1.0, dk1.abs().max().item())
        ev_ = (dv1 - dv2).abs().max().item() / max(1.0, dv1.abs().max().item())
        line += f"  | rel diff dk {ek:.2e} dv {ev_:.2e}  nan {int(torch.isnan(dk2).any() or torch.isnan(dv2).any())}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
