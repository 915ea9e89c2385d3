#!/usr/bin/env python3
"""Instruction-mix summary of the attention kernels from rocprofv3 --pmc counter_collection CSVs.

usage: pmc_issue.py OUT.json LABEL=DIR [LABEL=DIR ...]
Every DIR holds the CSVs of several counter passes over the same command (each pass a run of its
own: at most 8 SQ counters per run).  Output per label and kernel: every counter as its average
per launch (SQ counters are summed over the chip's shader engines by rocprofv3), plus
  valu_per_mfma      vector-ALU instructions issued per MFMA instruction (MFMAs excluded)
  lds_per_mfma       LDS instructions per MFMA
  salu_per_mfma      scalar-ALU instructions per MFMA
  trans_per_mfma     transcendental (exp2) instructions per MFMA
  valu_active_share  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a wave's life in which it has a
                     vector instruction (MFMA included) in flight
  lds_wait_share     SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES: share of a wave's life spent waiting for LDS
(SQ_INSTS_VALU counts MFMAs too on gfx950: they are subtracted for valu_per_mfma.)
"""
import csv, glob, json, os, sys


def collect(d):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").replace("vfa::", "").strip()
                if not k.startswith(("fwd_", "bwd_")):
                    continue
                a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add((path, row["Dispatch_Id"]))
    out = {}
    for k, cs in acc.items():
        c = {n: v[0] / max(1, len(v[1])) for n, v in cs.items()}
        o = {"counters_per_launch": c}
        mf = c.get("SQ_INSTS_MFMA") or 0.0
        if mf:
            if "SQ_INSTS_VALU" in c:
                o["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - mf) / mf
            for name, key in (("SQ_INSTS_LDS", "lds_per_mfma"), ("SQ_INSTS_SALU", "salu_per_mfma"),
                              ("SQ_INSTS_VALU_TRANS_F32", "trans_per_mfma"), ("SQ_INSTS_VALU_CVT", "cvt_per_mfma"),
                              ("SQ_INSTS_VMEM", "vmem_per_mfma")):
                if name in c:
                    o[key] = c[name] / mf
        wc = c.get("SQ_WAVE_CYCLES") or 0.0
        if wc:
            for name, key in (("SQ_ACTIVE_INST_VALU", "valu_active_share"), ("SQ_ACTIVE_INST_LDS", "lds_active_share"),
                              ("SQ_ACTIVE_INST_SCA", "scalar_active_share"), ("SQ_WAIT_INST_LDS", "lds_wait_share"),
                              ("SQ_WAIT_INST_ANY", "any_inst_wait_share"), ("SQ_WAIT_ANY", "any_wait_share")):
                if name in c:
                    o[key] = c[name] / wc
        out[k] = o
    return out


def main():
    res = {"_note": "rocprofv3 --pmc passes (each a run of its own, <= 8 SQ counters) over python3 bench.py "
                    "--workload c3 --steps 3 --warmup 1; per kernel, average per launch; see tools/pmc_issue.py "
                    "for the derived ratios"}
    for arg in sys.argv[2:]:
        label, d = arg.split("=", 1)
        res[label] = collect(d)
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
    for label, ks in res.items():
        if label == "_note":
            continue
        for k, o in ks.items():
            print(label, k, {a: round(b, 3) for a, b in o.items() if a != "counters_per_launch"})


if __name__ == "__main__":
    main()
