#!/usr/bin/env python3
"""Timeline summary of the steps in a rocprofv3 kernel trace of bench.py: how much of a step's kernel time overlaps.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    python scripts/trace_overlap.py <round-tag> gpurun_out/trace/.../tr_kernel_trace.csv

A step (one pass of forward_single_quality) starts with `s2d_input_kernel`, the first launch of the plan.  bench.py runs
the plan in two ways and the trace holds both: as hipGraph replays (warm-up + timed steps: the two branches of the graph —
base slices | progressive (mu, sigma) chain — on two hardware queues) and, afterwards, serialised on one stream for the
HIP-event measurement behind `roofline` (each launch alone).  Per step: wall (first start -> last end), the SUM of the
kernel durations, the UNION of their busy intervals, and the busy time per hardware queue.  Writes
profiles/<tag>_timeline.md."""
import csv
import os
import sys


def main():
    tag, path = sys.argv[1], sys.argv[2]
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "s2d_input_kernel" in r["Kernel_Name"]]
    bounds = starts[1:] + [len(rows)]
    # launches per step: the mode over the steps (the last step of the trace runs into whatever follows it)
    counts = sorted(b - a for a, b in zip(starts, bounds))
    per_step = counts[len(counts) // 2]
    out = []
    for a, b in zip(starts, bounds):
        ks = rows[a:min(b, a + per_step)]
        if len(ks) < per_step:
            continue
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ks)
        tot = sum(e - s for s, e in iv)
        union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        union += cur_e - cur_s
        wall = max(e for _, e in iv) - iv[0][0]
        conv = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks if "conv_igemm_kernel" in r["Kernel_Name"] or "resunit192_kernel" in r["Kernel_Name"] or "stack_tail_kernel" in r["Kernel_Name"])
        queues = {}
        for r in ks:
            q = r["Queue_Id"]
            queues[q] = queues.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy_q = [v for v in queues.values() if v > 1e6]
        out.append({"launches": len(ks), "wall_ms": wall / 1e6, "sum_ms": tot / 1e6, "union_ms": union / 1e6,
                    "conv_sum_ms": conv / 1e6, "queues": {k: v / 1e6 for k, v in queues.items()},
                    "kind": "two-branch run" if len(busy_q) >= 2 else "serialised"})
    graph = [o for o in out if o["kind"] == "two-branch run"][-5:]
    serial = [o for o in out if o["kind"] == "serialised"][-3:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", f"{tag}_timeline.md"), "w") as g:
        g.write(f"# step timeline {tag}\n\n`rocprofv3 --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16`, "
                f"from the kernel trace (`scripts/trace_overlap.py`; {len(out)} steps in the trace, {per_step} launches each).  "
                "Two-branch runs (hipGraph replays; two streams with --no-graph) = what `value` / `ms_per_step` time; serialised passes = what the HIP-event figures of `roofline` time.\n\n"
                "| step kind | launches | wall ms (first start -> last end) | sum of kernel durations ms | of which conv_igemm | union of busy intervals ms | overlapped ms (sum - union) | busy ms per HW queue |\n|---|---|---|---|---|---|---|---|\n")
        for o in graph + serial:
            g.write(f"| {o['kind']} | {o['launches']} | {o['wall_ms']:.3f} | {o['sum_ms']:.3f} | {o['conv_sum_ms']:.3f} | {o['union_ms']:.3f} | "
                    f"{o['sum_ms'] - o['union_ms']:.3f} | {', '.join(f'q{k}: {v:.2f}' for k, v in sorted(o['queues'].items()) if v > 0.005)} |\n")
        if graph and serial:
            gw = sum(o["union_ms"] for o in graph) / len(graph)
            gs = sum(o["sum_ms"] for o in graph) / len(graph)
            ss = sum(o["sum_ms"] for o in serial) / len(serial)
            g.write(f"\nReading: under the graph the two branches run side by side ({gs - gw:.1f} ms of kernel time overlapped per step), but kernels "
                    f"that share the chip slow each other down: the same launches sum to {gs:.1f} ms inside the graph against {ss:.1f} ms "
                    f"when each runs alone, so the step's busy time is {gw:.1f} ms against {ss:.1f} ms serialised.  The two-branch graph "
                    "removes the launch gaps and hides the short launches of the slice chain; it does not buy compute-bound time.\n")
    for o in graph + serial:
        print(o)


if __name__ == "__main__":
    main()
