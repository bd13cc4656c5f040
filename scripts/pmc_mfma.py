#!/usr/bin/env python3
"""MFMA utilisation / effective clock per kernel from a rocprofv3 SQ+GRBM counter pass.

  python scripts/pmc_mfma.py <counter_collection.csv> [<kernel_trace.csv>] > profiles/rNN_mfma_util.md

Per MI355X_MICROARCH.md ('DVFS give-back', cycle constants): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
effective shader clock of a dispatch = GRBM_GUI_ACTIVE / 8 / wall time; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
per SIMD-issue (64 per v_mfma_f32_32x32x2_f32), so
    MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles and are reported as shares of SQ_WAVE_CYCLES.
"""
import collections
import csv
import sys


def main():
    cc = sys.argv[1]
    rows = list(csv.DictReader(open(cc)))
    dur = {}
    if len(sys.argv) > 2:
        for r in csv.DictReader(open(sys.argv[2])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    per = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch -> counter -> value
    name = {}
    for r in rows:
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        name[d] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if d not in dur and "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d, c in per.items():
        a = agg[name[d]]
        for k, v in c.items():
            a[k] += v
        a["_ns"] += dur.get(d, 0)
        a["_n"] += 1
    print("| kernel | launches | wall ms (profiled) | eff. clock GHz | MFMA pipe util | executed MFMA TF/s | fp32-equivalent TF/s | wait_any | wait_inst | active_inst | LDS conflict / LDS active |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
        if a["_ns"] < 2e5:
            continue
        gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clk = gui / a["_ns"] if a["_ns"] else 0.0
        util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else 0.0
        wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        # MODE template argument (6th) of conv_igemm_kernel: 1 = bf16 split operands (1024 FLOP/clk/SIMD, 6 MFMA FLOP per
        # algorithmic FLOP), 0 = fp32 operands (64 FLOP/clk/SIMD)
        targs = k[k.find("<") + 1:k.rfind(">")].split(",") if "<" in k else []
        split = ("conv_igemm_kernel" in k and len(targs) >= 6 and targs[5].strip() == "1") or "resunit192_kernel" in k or "stack_tail_kernel" in k
        tf = util * 1024 * (1024 if split else 64) * clk / 1e3
        tf_eq = tf / 6.0 if split else tf
        ldsr = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        print(f"| `{k[:60]}` | {int(a['_n'])} | {a['_ns'] / 1e6:.2f} | {clk:.2f} | {100 * util:.1f} % | {tf:.0f} | {tf_eq:.1f} | "
              f"{100 * a.get('SQ_WAIT_ANY', 0) / wc:.1f} % | {100 * a.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} % | "
              f"{100 * a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} % | {ldsr:.3f} |")


if __name__ == "__main__":
    main()
