#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/.

  python scripts/pmc_summary.py <round-tag> <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <n_forward_passes_in_pmc_run>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE
reads exactly half of the bytes of a wide coalesced (16 B/lane) stream on gfx950, so reads are doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.  The counters sit on the L2's fabric side, so Infinity-Cache
hits are included: this is traffic beyond L2, an upper bound on HBM bytes.
"""
import collections
import csv
import json
import os
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
    return agg


def main():
    tag, stats, fetch, write, passes = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    # the convolution family = the implicit-GEMM kernel and the fused residual-unit kernel (three convolutions per launch)
    fam = {"conv_igemm": ("conv_igemm_kernel", "resunit192_kernel", "stack_tail_kernel"), "win_attn": ("win_attn_kernel", "win_attn8_mfma_kernel"),
           "variance_mask": ("variance_mask_kernel",), "gauss_tail": ("gauss_tail_kernel",)}
    res = {}
    for name, pats in fam.items():
        hit = lambda k: any(p_ in k for p_ in pats)
        fk = sum(v[0] for k, v in f.items() if hit(k))
        wk = sum(v[0] for k, v in w.items() if hit(k))
        n = sum(v[1] for k, v in f.items() if hit(k))
        res[name] = {"launches_per_step": n // passes,
                     "fetch_bytes_per_step": 2 * fk * 1024 / passes,      # x2: gfx950 FETCH_SIZE correction
                     "write_bytes_per_step": wk * 1024 / passes,
                     "traffic_bytes_per_step": (2 * fk + wk) * 1024 / passes}
    json.dump(res, open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as g:
        g.write(open(stats).read())
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(out_dir, f"{tag}_summary.md"), "w") as g:
        g.write(f"# rocprofv3 summary {tag}\n\n`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16`\n\n")
        g.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:20]:
            g.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
        g.write("\n## traffic beyond L2 per forward step (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)\n\n")
        g.write("| family | launches/step | read GB (2xFETCH_SIZE) | written GB | total GB |\n|---|---|---|---|---|\n")
        for k, v in res.items():
            g.write(f"| {k} | {v['launches_per_step']} | {v['fetch_bytes_per_step']/1e9:.2f} | {v['write_bytes_per_step']/1e9:.2f} | {v['traffic_bytes_per_step']/1e9:.2f} |\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
