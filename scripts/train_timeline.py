#!/usr/bin/env python3
"""Concurrency summary of one first_train step from a rocprofv3 kernel trace of scripts/bench_train.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o tt -- python3 scripts/bench_train.py --steps 3 --warmup 2
    python scripts/train_timeline.py gpurun_out/tt/.../tt_kernel_trace.csv > profiles/<tag>_first_train_timeline.md

A step starts with `s2d_input_kernel` (first launch of the forward plan).  For the LAST complete graph-replayed step: wall,
sum of kernel durations, union of busy intervals, time with 0 / 1 / 2 / >= 3 kernels in flight, busy time per hardware
queue, and — the part that says where the step still waits — the time during which exactly ONE kernel is in flight,
summed by kernel name with that kernel's workgroup count (a kernel alone with fewer workgroups than the chip has CUs
leaves the rest idle)."""
import csv
import sys
from collections import defaultdict


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "s2d_input_kernel" in r["Kernel_Name"]]
    steps = [(a, b) for a, b in zip(starts, starts[1:])]
    # graph replays use more than one hardware queue; take the last such step that is complete
    pick = None
    for a, b in steps:
        if len({rows[i]["Queue_Id"] for i in range(a, b)}) >= 2:
            pick = (a, b)
    if pick is None:
        pick = steps[-1]
    ks = rows[pick[0]:pick[1]]
    t0 = int(ks[0]["Start_Timestamp"])
    ev = []
    for i, r in enumerate(ks):
        ev.append((int(r["Start_Timestamp"]), 1, i))
        ev.append((int(r["End_Timestamp"]), -1, i))
    ev.sort(key=lambda e: (e[0], e[1]))
    level = defaultdict(int)
    solo = defaultdict(lambda: [0, 0, 0])          # name -> [ns alone, launches touched, workgroups]
    live = set()
    last = ev[0][0]
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            level[min(len(live), 3)] += dt
            if len(live) == 1:
                (j,) = live
                r = ks[j]
                nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
                wg = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
                s = solo[(nm, wg)]
                s[0] += dt
        last = t
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
    wall = max(int(r["End_Timestamp"]) for r in ks) - t0
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    queues = defaultdict(int)
    for r in ks:
        queues[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ms = lambda v: f"{v / 1e6:.2f}"
    print(f"# first_train step timeline ({len(ks)} launches; last complete graph-replayed step of the trace)\n")
    print("| wall ms | sum of kernel durations | nothing in flight | exactly 1 kernel | 2 kernels | >= 3 kernels | busy per HW queue |")
    print("|---|---|---|---|---|---|---|")
    print(f"| {ms(wall)} | {ms(tot)} | {ms(level[0])} | {ms(level[1])} | {ms(level[2])} | {ms(level[3])} | "
          + " / ".join(ms(v) for v in sorted(queues.values(), reverse=True)) + " |\n")
    print("## time with exactly one kernel in flight, by kernel and grid (top 25)\n")
    print("| kernel | workgroups | ms alone |")
    print("|---|---|---|")
    for (nm, wg), s in sorted(solo.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"| `{nm}` | {wg} | {ms(s[0])} |")
    small = sum(s[0] for (nm, wg), s in solo.items() if wg < 256)
    print(f"\nalone with fewer than 256 workgroups: {ms(small)} ms; alone with 256 - 511: "
          f"{ms(sum(s[0] for (nm, wg), s in solo.items() if 256 <= wg < 512))} ms")


if __name__ == "__main__":
    main()
