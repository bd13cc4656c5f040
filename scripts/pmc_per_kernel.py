#!/usr/bin/env python3
"""Fabric traffic and rate per kernel configuration from the two traffic counter passes.

  python scripts/pmc_per_kernel.py <fetch_counter_collection.csv> <write_counter_collection.csv> <fetch_kernel_trace.csv>

Groups dispatches by (kernel, grid size): launches, average duration, bytes read beyond L2 (2 x FETCH_SIZE KiB: the gfx950
correction of MI355X_MICROARCH.md), bytes written (WRITE_SIZE KiB) and the resulting rate.  Durations come from the
FETCH pass's kernel trace (counter collection serialises dispatches; durations are within a few % of an unprofiled run)."""
import collections
import csv
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    dur = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(sys.argv[3])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size_X"])
        dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dur[k][1] += 1
    rows = []
    for k, (fv, n) in f.items():
        d = dur[k][0] / max(dur[k][1], 1) * 1e-9
        rd, wr = 2 * fv * 1024 / n, w[k][0] * 1024 / max(w[k][1], 1)
        rows.append((dur[k][0], k, n, d, rd, wr))
    rows.sort(reverse=True)
    print("| kernel | grid | launches | avg us | read MB | written MB | TB/s beyond L2 |\n|---|---|---|---|---|---|---|")
    for _, k, n, d, rd, wr in rows[:60]:
        print(f"| `{k[0]}` | {k[1]} | {n} | {d*1e6:.1f} | {rd/1e6:.1f} | {wr/1e6:.1f} | {(rd+wr)/d/1e12 if d else 0:.2f} |")


if __name__ == "__main__":
    main()
