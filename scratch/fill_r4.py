"""Fill the R4_* placeholders of DESIGN.md / README.md from profiles/r04_bench.json (run once, after scripts/profile_round.sh r04)."""
import json
import os
import re

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
r = d["roofline"]
g = r["g_a_g_s"]
ft = d["train"]["first_train"]
rm = d["train"]["rem_finetune"]
cb = d["cpu_baseline"]
ph = ft["phase_ms"]
cls = r["classes"]
sub = {
    "R4_STEP": f"**{d['ms_per_step']:.2f} ms = {d['value']:.1f} MP/s** (`r04_bench.json`; the boxes of the round: 26.1 – 27.3 ms with the same kernels)",
    "R4_CONV": f"{r['launches_per_step']} launches, {r['kernel_ms_per_step']['conv_igemm']:.2f} ms; {r['achieved']:.1f} TF/s algorithmic = {r['executed_mfma_tflops']:.0f} TF/s executed; "
               f"**`roofline.frac` {r['frac']:.3f}** ({r['power_limited_ceiling']['frac']:.2f} of the power-limited ceiling); per class (ms / TF/s): "
               + " · ".join(f"{k} {v['ms_per_step']:.2f} / {v['tflops']:.0f}" for k, v in cls.items())
               + f"; traffic beyond L2 {r['traffic'] / 1e6:.0f} MB per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic (`r04_pmc_traffic.json`)",
    "R4_GAGS": f"{g['ms_per_step']:.2f} ms, {g['achieved']:.1f} TF/s, **frac {g['frac']:.3f}** (north star ≥ 0.40: **not met**)",
    "R4_TRAINPHASE": f"forward {ph['train_forward']:.1f} ms, backward {ph['backward_incl_all_reduce']:.1f} ms (= {ph['backward_incl_all_reduce'] / ph['train_forward']:.2f} × forward; the judge's bar ≤ 2.1 ×), "
                     f"clip + Adam {ph['clip_adam']:.1f} ms — each phase alone between device synchronisations (the step itself overlaps the optimiser's host work with the device)",
    "R4_TRAIN": f"first_train **{ft['ms_per_step']:.1f} ms = {ft['images_per_s']:.1f} images/s** ({ft['tflops']:.1f} TF/s algorithmic over forward + backward; round 3: 159.9 ms = 200.1; the judge's bar ≤ 125 ms: **not met**); "
                f"rem_finetune {rm['ms_per_step']:.1f} ms = {rm['images_per_s']:.0f} images/s (round 3: 24.4 ms)",
    "R4_CPU": f"{cb['value']:.3f} MP/s, oracle (\"port\") on {cb['cores']} threads of the box's {cb['cpu']}; {cb['sample']}",
    "R4_FTIMG": f"{ft['images_per_s']:.1f}",
    "R4_FT": f"{ft['ms_per_step']:.1f}",
    "R4_REM": f"{rm['ms_per_step']:.1f} ms = {rm['images_per_s']:.0f} images/s (round 3: 24.4 ms = 657)",
}
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(root, name)
    s = open(p).read()
    for k in sorted(sub, key=len, reverse=True):
        s = re.sub(r"\b" + k + r"\b", sub[k].replace("\\", "\\\\"), s)
    open(p, "w").write(s)
    left = re.findall(r"R4_[A-Z]+", s)
    print(name, "placeholders left:", left)
