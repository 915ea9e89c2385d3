#!/usr/bin/env python3
"""Every kernel geometry on a grid of shapes (development tool): for the forward, the dK/dV and the
dQ kernel, time each candidate (development library, kernel-variant switches) interleaved in one
process and print / store the medians, the best candidate and what the shipped dispatch takes.

  dispatch_sweep.py [out.json]
Shapes: (batch x heads) in {1x8, 2x16, 8x32} x seqlen in {1024 ... 16384}, d = 128 bf16, causal and
full; batch is capped so that q, k, v, do and the outputs stay below a few GiB."""
import ctypes as C
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
import bench

FWD = {"persist8": {3: 2}, "block8": {3: 1, 6: 1}, "block4": {3: 1, 6: 2}, "shipped": {}}
DQ = {"persist8": {3: 2}, "block8": {3: 1, 7: 1}, "block4": {3: 1, 7: 2}, "shipped": {}}
DKDV = {"block8": {0: 1}, "block4": {0: 5}, "wave64": {0: 2}, "shipped": {}}


def main():
    lib = cabi.dev_lib()
    cabi._lib = lib
    hip = bench.Hip()
    stream = torch.cuda.current_stream().cuda_stream

    def tuned(sw):
        for i in range(8):
            lib.vfa_debug_set_tuning(i, 0)
        for k, v in sw.items():
            lib.vfa_debug_set_tuning(k, v)

    res = []
    for b, h in ((1, 8), (2, 16), (8, 32)):
        for s in (1024, 2048, 4096, 8192, 16384):
            if b * h * s > 8 * 32 * 4096 * 2:
                continue
            for causal in (True, False):
                torch.manual_seed(0)
                q, k, v, do = (torch.randn(b, s, h, 128, device="cuda", dtype=torch.bfloat16) for _ in range(4))
                out = torch.empty_like(q)
                lse = torch.empty(b, h, s, device="cuda", dtype=torch.float32)
                dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                sd = torch.empty(b, h, (s + 127) // 128 * 128, device="cuda", dtype=torch.float32)
                acc = torch.empty(1, device="cuda", dtype=torch.float32)
                Pf = bench.build_params(cabi, q, k, v, out, lse, causal, stream)
                Pb = bench.build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, acc)
                tuned({})
                cabi.call("vfa_mha_fwd", Pf)
                row = {"b": b, "h": h, "s": s, "causal": causal}
                reps = 9 if s <= 4096 else 5
                # forward
                t = {n: [] for n in FWD}
                for r in range(reps):
                    for n, sw in FWD.items():
                        tuned(sw)
                        e0, e1 = hip.event(), hip.event()
                        hip.record(e0, stream); cabi.call("vfa_mha_fwd", Pf); hip.record(e1, stream)
                        torch.cuda.synchronize()
                        if r >= 2:
                            t[n].append(hip.elapsed_ms(e0, e1))
                row["fwd"] = {n: sorted(x)[len(x) // 2] for n, x in t.items()}
                # backward: dK/dV (events 1..2) and dQ (2..3) candidates
                for name, cands, sl in (("dkdv", DKDV, (1, 2)), ("dq", DQ, (2, 3))):
                    t = {n: [] for n in cands}
                    for r in range(reps):
                        for n, sw in cands.items():
                            tuned(sw)
                            ev = (C.c_void_p * 4)(*[hip.event() for _ in range(4)])
                            Pb.stage_events = C.cast(ev, C.c_void_p)
                            cabi.call("vfa_mha_bwd", Pb)
                            torch.cuda.synchronize()
                            if r >= 2:
                                t[n].append(hip.elapsed_ms(C.c_void_p(ev[sl[0]]), C.c_void_p(ev[sl[1]])))
                    Pb.stage_events = None
                    row[name] = {n: sorted(x)[len(x) // 2] for n, x in t.items()}
                for name in ("fwd", "dkdv", "dq"):
                    cand = {n: x for n, x in row[name].items() if n != "shipped"}
                    best = min(cand, key=cand.get)
                    row[name + "_best"] = best
                    row[name + "_shipped_behind_best_pct"] = 100.0 * (row[name]["shipped"] / cand[best] - 1.0)
                res.append(row)
                print(json.dumps(row), flush=True)
                del q, k, v, do, out, dq, dk, dv
    tuned({})
    if len(sys.argv) > 1:
        json.dump({"_note": "tools/dispatch_sweep.py: median kernel times in ms (HIP events), d = 128 bf16; "
                            "'shipped' = what the library's dispatch takes; *_behind_best_pct = shipped vs the best candidate",
                   "rows": res}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
