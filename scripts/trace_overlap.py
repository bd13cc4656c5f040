#!/usr/bin/env python3
"""Timeline summary of graph-replayed steps from a rocprofv3 kernel trace: how much of a step's kernel time overlaps.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    python scripts/trace_overlap.py <round-tag> gpurun_out/trace/.../tr_kernel_trace.csv

A step (one hipGraph launch of forward_single_quality) starts with `s2d_input_kernel` (the first launch of the plan).
For the LAST `steps` steps of the trace the script reports per step: wall (first start -> last end), the SUM of the kernel
durations, the UNION of their busy intervals, and per queue the busy time — the sum exceeds the wall by exactly what the
two branches of the graph (base slices | progressive (mu, sigma) chain) run side by side.  Writes profiles/<tag>_timeline.md."""
import csv
import os
import sys


def main():
    tag, path = sys.argv[1], sys.argv[2]
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "s2d_input_kernel" in r["Kernel_Name"]]
    starts = starts[-n_steps:]
    bounds = starts[1:] + [len(rows)]
    out = []
    for a, b in zip(starts, bounds):
        ks = rows[a:b]
        # the last step runs to the end of the trace: cut it at the kernel that ends g_s (the NCHW store), i.e. keep
        # as many launches as the previous steps had
        if out and len(ks) > out[0]["launches"]:
            ks = ks[:out[0]["launches"]]
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
        conv = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks if "conv_igemm_kernel" in r["Kernel_Name"])
        queues = {}
        for r in ks:
            q = r["Queue_Id"]
            queues[q] = queues.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        out.append({"launches": len(ks), "wall_ms": wall / 1e6, "sum_ms": tot / 1e6, "union_ms": union / 1e6,
                    "conv_sum_ms": conv / 1e6, "queues": {k: v / 1e6 for k, v in queues.items()}})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", f"{tag}_timeline.md"), "w") as g:
        g.write(f"# graph-replay timeline {tag}\n\n`rocprofv3 --kernel-trace -- python3 bench.py --steps {n_steps} --warmup 2 --no-cpu-baseline`, "
                "the timed steps (one hipGraph launch each), from the kernel trace (`scripts/trace_overlap.py`).\n\n"
                "| step | launches | wall ms (first start -> last end) | sum of kernel durations ms | of which conv_igemm | union of busy intervals ms | overlapped ms (sum - union) | busy ms per HW queue |\n|---|---|---|---|---|---|---|---|\n")
        for i, o in enumerate(out):
            g.write(f"| {i} | {o['launches']} | {o['wall_ms']:.3f} | {o['sum_ms']:.3f} | {o['conv_sum_ms']:.3f} | {o['union_ms']:.3f} | "
                    f"{o['sum_ms'] - o['union_ms']:.3f} | {', '.join(f'q{k}: {v:.2f}' for k, v in sorted(o['queues'].items()))} |\n")
        g.write("\nThe sum of the kernel durations is larger than the step because the graph has two branches (base slices on one, "
                "the progressive (mu, sigma) chain on the other: `models._FsqPlan`): their kernels run side by side, which the "
                "HIP-event figure of bench.py (each launch alone, serialised) cannot show.\n")
    for o in out:
        print(o)


if __name__ == "__main__":
    main()
