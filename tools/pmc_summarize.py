#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) into the
per-kernel, per-launch HBM traffic summary bench.py reads.

usage: pmc_summarize.py OUT.json WORKLOAD=DIR_WITH_CSVS [WORKLOAD=DIR ...]
FETCH_SIZE and WRITE_SIZE count KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of wide
coalesced reads as 64 bytes).
"""
import csv, glob, json, os, sys


def collect(d):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").replace("vfa::", "").strip()
                c = row["Counter_Name"]
                a = acc.setdefault(k, {}).setdefault(c, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row["Dispatch_Id"])
    out = {}
    for k, cs in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            n = max(1, len(cs["GRBM_GUI_ACTIVE"][1]))
            mf, gui = cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n, cs["GRBM_GUI_ACTIVE"][0] / n
            out.setdefault(k, {})["mfma"] = {
                "sq_valu_mfma_busy_cycles": mf, "grbm_gui_active": gui,
                "sq_busy_cycles": cs.get("SQ_BUSY_CYCLES", [0.0])[0] / n,
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (it equals 32 cycles x
                # the number of 32x32x16 MFMAs issued); GRBM_GUI_ACTIVE is summed over the 8 XCDs,
                # so one XCD's wall-clock cycles are gui / 8 and the busy share is
                # mfma / (gui / 8 * 1024).
                "mfma_busy_frac": mf / (gui * 128.0) if gui else None}
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        f = cs["FETCH_SIZE"][0] / max(1, len(cs["FETCH_SIZE"][1]))
        w = cs["WRITE_SIZE"][0] / max(1, len(cs["WRITE_SIZE"][1]))
        out.setdefault(k, {}).update({"fetch_size_kb_raw": f, "read_bytes": f * 1024 * 2, "write_size_kb_raw": w,
                  "write_bytes": w * 1024, "hbm_bytes": f * 2048 + w * 1024})
    return out


def main():
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch; units KB -> bytes "
                    "(x1024); FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads (counter "
                    "tallies 128-B requests at 64 B); command: rocprofv3 --pmc <C> -- python3 bench.py "
                    "--workload <w> --steps 3 --warmup 1"}
    for arg in sys.argv[2:]:
        w, d = arg.split("=", 1)
        res[w] = {k: v for k, v in collect(d).items() if k.startswith(("fwd_", "bwd_"))}
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({w: sorted(v) for w, v in res.items() if w != "_note"}))


if __name__ == "__main__":
    main()
