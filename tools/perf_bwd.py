#!/usr/bin/env python3
"""Per-kernel timing of forward + backward on one GPU (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import cabi
if os.environ.get("VFA_LIB"):  # A/B runs of two builds on one box
    cabi.LIB_PATH = os.environ["VFA_LIB"]
import bench


def main():
    hip = bench.Hip()
    torch.manual_seed(0)
    shapes = [(8, 4096, 32, 32, 128, True, "bf16"), (8, 4096, 32, 32, 128, False, "bf16"),
              (4, 2048, 16, 16, 64, True, "bf16"), (4, 4096, 32, 8, 128, True, "bf16"),
              (2, 8192, 16, 16, 128, True, "bf16"), (16, 1024, 16, 16, 128, True, "bf16"),
              (32, 512, 16, 16, 128, True, "bf16"), (32, 512, 16, 16, 128, False, "bf16")]
    if len(sys.argv) > 1:
        shapes = [shapes[int(a)] for a in sys.argv[1:]]
    stream = torch.cuda.current_stream().cuda_stream
    for (b, s, h, hk, d, causal, dt) in shapes:
        tdt = torch.bfloat16 if dt == "bf16" else torch.float16
        q = torch.randn(b, s, h, d, device="cuda", dtype=tdt)
        k = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        v = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        do = torch.randn_like(q)
        out = torch.empty_like(q)
        lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sd = torch.empty(b, h, (s + 127) // 128 * 128, device="cuda", dtype=torch.float32)
        acc = torch.empty(1, device="cuda", dtype=torch.float32)
        Pf = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
        Pb = bench.build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, acc)
        pdrop = float(os.environ.get("VFA_PDROP", "0"))  # dropout variants of the three kernels
        Pf.p_dropout = Pb.p_dropout = pdrop
        Pf.philox_seed = Pb.philox_seed = 1234
        n = 10
        for _ in range(3):
            cabi.call("vfa_mha_fwd", Pf); cabi.call("vfa_mha_bwd", Pb)
        evs = [(C.c_void_p * 4)(*[hip.event() for _ in range(4)]) for _ in range(n)]
        fe = [(hip.event(), hip.event()) for _ in range(n)]
        for i in range(n):
            hip.record(fe[i][0], stream); cabi.call("vfa_mha_fwd", Pf); hip.record(fe[i][1], stream)
            Pb.stage_events = C.cast(evs[i], C.c_void_p)
            cabi.call("vfa_mha_bwd", Pb)
        torch.cuda.synchronize()
        med = lambda xs: sorted(xs)[len(xs) // 2]
        tf = med([hip.elapsed_ms(a, b_) for a, b_ in fe])
        t0 = med([hip.elapsed_ms(C.c_void_p(e[0]), C.c_void_p(e[1])) for e in evs])
        t1 = med([hip.elapsed_ms(C.c_void_p(e[1]), C.c_void_p(e[2])) for e in evs])
        t2 = med([hip.elapsed_ms(C.c_void_p(e[2]), C.c_void_p(e[3])) for e in evs])
        f = bench.fwd_flops(b, s, s, h, d, causal)
        tot = tf + t0 + t1 + t2
        print(f"b{b} s{s} h{h}/{hk} d{d} causal={causal}: fwd {tf:.3f} ms ({f/tf/1e9:.0f} TF) | pre {t0:.3f} "
              f"dkdv {t1:.3f} ms ({2*f/t1/1e9:.0f} TF-alg4/5) dq {t2:.3f} ms | bwd {2.5*f/(t0+t1+t2)/1e9:.0f} TF "
              f"| fwd+bwd {3.5*f/tot/1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
