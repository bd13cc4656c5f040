"""Execution plans: the reference's module graph lowered to libvampic launches.

A :class:`Plan` owns every NHWC buffer it needs (allocated while the plan is built,
never while it runs) and a flat list of pre-marshalled C-ABI calls.  ``run()`` replays
them on the current HIP stream; ``graph()`` captures the replay once into a hipGraph so
a steady-state step is a single ``hipGraphLaunch``.

Lowering rules (reference lines in brackets)
  * torch.cat of supports never happens: a conv problem reads up to 4 channel windows
    [pic.py:528-529,548,598-599,635];
  * GELU / LeakyReLU / 0.5*tanh / sigmoid-gate / residual adds / GDN's x*rsqrt(.) /
    clamp(0,1) / PixelShuffle are conv epilogues [layers.py:30-86, gdn.py:62-75, rem.py:52-66];
  * structurally identical, mutually independent stacks run in lockstep as grouped
    launches: the two encoders, the four hyper-synthesis stacks, the (mean, scale) pair of a
    slice, base slices 5..9, and the ten progressive LRP stacks [SURVEY §3b].
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import contextlib
import ctypes as C
import os
import torch
import torch.nn as nn

from . import _lib as L
from . import layers as Ly
from . import ops
from .ops import View


class Plan:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.steps: List[Callable[[], None]] = []
        self.keep: List[object] = []          # buffers / ctypes arrays referenced by the steps
        self._graph: Optional[ops.Graph] = None
        self.flops = 0.0
        self.meta: List[dict] = []            # one entry per step: what it is and its algorithmic FLOP
        self.branch_of: List[int] = []        # stream branch of each step (0 = the caller's stream)
        self.cur_branch = 0
        self.class_of: List[int] = []         # launch class of each step (event profiler: L.PROF_CLASSES)
        self.cur_class = 0
        self.n_events = 0
        self._side: Dict[int, "torch.cuda.Stream"] = {}
        # bf16-storage mode (BASELINE configs[2] "bf16"): while ``act16`` is on, intermediate tensors of at least
        # ``act16_min_hw`` positions per image are bf16 (the large feature maps of g_a / g_s); everything at latent
        # resolution — the sigma stacks, the mask, the likelihoods — stays fp32
        self.act16 = False
        self.act16_min_hw = 4096        # 64 x 64 positions per image ("bf16 activations on >= 64^2 feature maps")
        # fp16x2 mode (opt-in, VAMPIC_CONV=f16x2): every conv launch needs an upper bound of max |x| of each input segment.
        # One cell per plan buffer, written by the epilogues of the conv launches that fill the buffer (vam_conv.out_amax);
        # inputs that something else wrote (or that the plan does not own) are reduced in front of the launch (vam_absmax).
        # The pool is zeroed by the plan's first step, ahead of any branch.
        self._amax_pool: Optional[torch.Tensor] = None
        self._amax_next = 0
        self._amax_bufs: List[list] = []      # [start, end, cell index, state]  state: 0 unwritten, 1 conv-written, 2 other
        if ops.f16x2_mode():
            self._amax_pool = torch.zeros(8192, dtype=torch.int32, device=self.device)
            pool = self._amax_pool
            self.call(lambda: ops.memset_zero(pool), "zero max cells")

    # ---- buffers
    def buf(self, B, H, W, C, zero=False) -> View:
        if self.act16 and not zero and H * W >= self.act16_min_hw and C % 8 == 0:
            v16 = ops.new_view16(B, H, W, C, self.device)
            self.keep.append(v16.buf)
            return v16
        v = ops.new_view(B, H, W, C, self.device, zero=zero)
        self.keep.append(v.buf)
        self._amax_register(v)
        return v

    def buf32(self, B, H, W, C) -> View:
        """fp32 whatever the storage mode (tensors the window-attention kernel reads / writes)."""
        v = ops.new_view(B, H, W, C, self.device)
        self.keep.append(v.buf)
        self._amax_register(v)
        return v

    # ---- fp16x2 mode: max-|x| cells
    def _amax_cell(self) -> int:
        i = self._amax_next
        assert i < self._amax_pool.numel(), "max-cell pool exhausted"
        self._amax_next += 1
        return i

    def _amax_register(self, v: View):
        if self._amax_pool is not None:
            p0 = v.buf.data_ptr()
            self._amax_bufs.append([p0, p0 + v.buf.numel() * 4, self._amax_cell(), 0])

    def _amax_find(self, ptr: int):
        for b in self._amax_bufs:
            if b[0] <= ptr < b[1]:
                return b
        return None

    def written_by_other(self, *views):
        """Tell the plan that something other than a conv launch writes these tensors (their max cells are not trusted)."""
        if self._amax_pool is not None:
            for v in views:
                b = self._amax_find(v.ptr)
                if b is not None:
                    b[3] = 2

    def _amax_cells(self, chunk):
        lib = L.load()
        base = self._amax_pool.data_ptr()
        fills = []
        check = os.environ.get("VAMPIC_AMAX_CHECK", "0") == "1"
        track = os.environ.get("VAMPIC_AMAX_TRACK", "1") != "0"     # 0: reduce every input in front of its launch (exact max; measurement aid)
        for c in chunk:
            if c.flags & (L.CONV_W_BF16 | L.CONV_IN_BF3):
                continue
            npix = c.B * c.H * c.W
            for k in range(c.n_seg):
                b = self._amax_find(c.seg[k].ptr)
                if b is not None and b[3] == 1 and track:
                    c.in_amax[k] = base + 4 * b[2]
                    if check:
                        fills.append((c, k, base + 4 * self._amax_cell(), base + 4 * b[2]))
                else:
                    cell = base + 4 * self._amax_cell()
                    c.in_amax[k] = cell
                    fills.append((c, k, cell, None))
            if not (c.flags & L.CONV_OUT_NCHW):
                b = self._amax_find(c.out)
                if b is not None and b[3] != 2:
                    c.out_amax = base + 4 * b[2]
                    b[3] = 1
        if fills:
            self.keep.append(chunk)

            def fill():
                s = ops.stream_ptr()
                for c, k, cell, tracked in fills:
                    L.check(lib.vam_absmax(C.byref(c.seg[k]), 1, c.B * c.H * c.W, cell, s), "vam_absmax")
                    if tracked is not None:      # VAMPIC_AMAX_CHECK=1: the tracked cell must bound what the tensor holds now
                        torch.cuda.current_stream().synchronize()
                        off_t, off_c = (tracked - base) // 4, (cell - base) // 4
                        t, m = int(self._amax_pool[off_t]), int(self._amax_pool[off_c])
                        if t < m:
                            raise RuntimeError(f"fp16x2: tracked max cell {t:#x} below the tensor's max {m:#x} "
                                               f"(segment {k} of a [{c.N} ch, k{c.kh}] problem): an untracked writer")
            self.call(fill, "absmax")

    def pk(self, m, ins, out, aux=()):
        """Packed weights of ``m`` for a problem reading ``ins`` and writing ``out``: bf16 weights as soon as one of
        the tensors it touches is bf16-stored."""
        any16 = any(isinstance(v, ops.View16) for v in list(ins) + [out] + [a for a in aux if a is not None])
        return m.packed(True) if any16 else m.packed()

    def buf3(self, B, H, W, C) -> "ops.View3":
        """bf16x3-plane buffer for a tensor whose only consumer is another convolution."""
        v = ops.new_view3(B, H, W, C, self.device)
        self.keep.append(v.buf)
        return v

    # ---- recorded launches
    def conv(self, problems: Sequence[L.VamConv]):
        lib = L.load()
        for i in range(0, len(problems), L.VAM_MAX_GROUP):
            chunk = list(problems[i:i + L.VAM_MAX_GROUP])
            if self._amax_pool is not None:
                self._amax_cells(chunk)
            arr = (L.VamConv * len(chunk))(*chunk)
            n = len(chunk)
            self.keep.append(arr)
            fl = 0.0
            for c in chunk:
                cin = sum(c.seg[k].C for k in range(c.n_seg))
                fl += 2.0 * c.B * c.Ho * c.Wo * c.N * cin * c.kh * c.kw
            self.flops += fl
            c0 = chunk[0]
            self.meta.append({"kind": "conv", "flops": fl, "desc": f"{n}x[{sum(c0.seg[k].C for k in range(c0.n_seg))}->{c0.N} "
                              f"k{c0.kh}x{c0.kw} s{c0.stride} P={c0.B * c0.Ho * c0.Wo}]"})
            self.branch_of.append(self.cur_branch)
            self.class_of.append(self.cur_class)
            self.steps.append(lambda arr=arr, n=n: L.check(lib.vam_conv_group(arr, n, ops.stream_ptr()), "vam_conv_group"))

    def stack_tail(self, problems: Sequence["L.VamStackTail"]):
        """The last two layers of K slice stacks as one launch per VAM_MAX_TAIL_GROUP stacks (csrc/stack_tail.hip)."""
        lib = L.load()
        arr = (L.VamStackTail * len(problems))(*problems)
        n = len(problems)
        self.keep += [arr] + [p._keep for p in problems]
        c0 = problems[0]
        fl = 2.0 * n * c0.B * c0.H * c0.W * 9.0 * (128 * 64 + 64 * 32)
        self.flops += fl
        self.meta.append({"kind": "conv", "flops": fl, "desc": f"{n}x[128->64->32 k3x3 fused tail P={c0.B * c0.H * c0.W}]"})
        self.branch_of.append(self.cur_branch)
        self.class_of.append(self.cur_class)
        self.steps.append(lambda: L.check(lib.vam_stack_tail_group(arr, n, ops.stream_ptr()), "vam_stack_tail_group"))

    def resunit(self, problems: Sequence["L.VamResunit"]):
        """Fused residual units (one launch per group of up to VAM_MAX_GROUP units, csrc/resunit.hip)."""
        lib = L.load()
        for i in range(0, len(problems), L.VAM_MAX_GROUP):
            chunk = list(problems[i:i + L.VAM_MAX_GROUP])
            arr = (L.VamResunit * len(chunk))(*chunk)
            n = len(chunk)
            self.keep.append(arr)
            fl = sum(2.0 * c.B * c.H * c.W * (c.C * (c.C // 2) * 2 + (c.C // 2) ** 2 * 9) for c in chunk)
            self.flops += fl
            c0 = chunk[0]
            self.meta.append({"kind": "conv", "flops": fl, "desc": f"{n}x[resunit {c0.C} P={c0.B * c0.H * c0.W}]"})
            self.branch_of.append(self.cur_branch)
            self.class_of.append(self.cur_class)
            self.steps.append(lambda arr=arr, n=n: L.check(lib.vam_resunit_group(arr, n, ops.stream_ptr()), "vam_resunit_group"))

    # Weight gradients have no consumer inside a backward plan (they land in the flat gradient buffer), and inside one
    # transform every tensor they read — taped activations, output gradients — is written once.  Between defer_wgrad() and
    # flush_wgrad() their problems are only collected; the flush issues them grouped by shape (up to 16 per launch): the
    # six residual units of an attention block then share three launches instead of eighteen, each problem planning its
    # pixel splits against a sixth of the chip (fewer splits, fewer reduce launches, fuller grids).
    _wgrad_pending: Optional[list] = None

    def defer_wgrad(self):
        # OFF by default: measured SLOWER (interleaved on one box, profiles/r04_first_train_ab.txt: first_train 141.9 vs 137.7 ms).
        # A weight gradient launched right behind the data-gradient launch that produced its dY finds dY in the 256 MB
        # last-level cache; deferred to the end of the transform, both operands come from HBM.  VAMPIC_WGRAD_DEFER=1 opts in.
        if self._wgrad_pending is None and os.environ.get("VAMPIC_WGRAD_DEFER", "0") == "1":
            self._wgrad_pending = []

    def flush_wgrad(self):
        pend, self._wgrad_pending = self._wgrad_pending, None
        if pend:
            pend.sort(key=lambda c: (c.kh, c.stride == 2, c.N, c.C, c.B * c.H * c.W))
            self.wgrad(pend, now=True)

    # Weight gradients on their own branch.  Nothing downstream of a layer's weight gradient is on the backward's critical
    # path (the chain is data gradient -> data gradient), and the slice stacks' launches leave most of the chip idle (8192
    # pixels: 128 - 256 workgroups): with ``wgrad_branch`` set, a weight-gradient launch that no later step reads (not
    # ``now``) is recorded on that branch behind an event of the launch that produced its dY, so under capture it becomes a
    # parallel branch of the hipGraph.  Both operands are buffers that nothing writes after that event (fresh buffers of the
    # backward, the forward's tape, channel windows of the accumulators that are final by then) and every problem owns its
    # pixel-split scratch, so the branch needs no other ordering; :meth:`join` (and the end of every :meth:`run_range`)
    # orders the main branch behind it.
    wgrad_branch: Optional[int] = None
    wgrad_branches = 1        # > 1: consecutive weight-gradient launches alternate over this many branches (experiment)
    _wgrad_rr = 0

    def _next_wgrad_branch(self) -> int:
        b = self.wgrad_branch + (self._wgrad_rr % max(1, self.wgrad_branches))
        self._wgrad_rr += 1
        return b

    def join_wgrad(self):
        """The current branch waits for every weight-gradient branch."""
        if self.wgrad_branch is not None:
            for k in range(max(1, self.wgrad_branches)):
                self.join(self.wgrad_branch + k)

    def _on_wgrad_branch(self) -> bool:
        return self.wgrad_branch is not None and self.wgrad_branch <= self.cur_branch < self.wgrad_branch + max(1, self.wgrad_branches)

    def join(self, b: Optional[int]):
        """The current branch waits for everything recorded on branch ``b`` so far."""
        if b is None or b == self.cur_branch or b not in self.branch_of:
            return
        b0 = self.cur_branch
        self.branch(b)
        ev = self.record()
        self.branch(b0)
        self.wait(ev)

    wgrad_units = False       # whole gradient sub-chains (launch + re-parametrisation / un-padding) on the branch as well

    @contextlib.contextmanager
    def off_path(self):
        """Steps recorded inside go to the weight-gradient branch (when there is one and ``wgrad_units`` is on), behind
        everything recorded so far on the current branch: for a sub-chain that ENDS in parameter gradients — a ``now``
        weight-gradient launch into a temporary plus the step that re-indexes it, a bias-gradient column sum — which nothing
        later on the current branch reads."""
        if self.wgrad_branch is None or not self.wgrad_units or self._on_wgrad_branch():
            yield
            return
        ev, b0 = self.record(), self.cur_branch
        self.branch(self._next_wgrad_branch())
        self.wait(ev)
        try:
            yield
        finally:
            self.branch(b0)

    def wgrad(self, problems: Sequence[L.VamWgrad], now: bool = False):
        """Grouped weight-gradient launches (pre-marshalled like :meth:`conv`).  ``now``: a later step of the plan reads the
        result (a temporary that is re-indexed / re-parametrised into the gradient): never deferred, never on the side branch."""
        if self._wgrad_pending is not None and not now:
            self._wgrad_pending += list(problems)
            return
        if self.wgrad_branch is not None and not now and not self._on_wgrad_branch() and problems:
            ev, b0 = self.record(), self.cur_branch
            self.branch(self._next_wgrad_branch())
            self.wait(ev)
            try:
                self.wgrad(problems, now=True)
            finally:
                self.branch(b0)
            return
        lib = L.load()
        for i in range(0, len(problems), L.VAM_MAX_WGRAD_GROUP):
            chunk = list(problems[i:i + L.VAM_MAX_WGRAD_GROUP])
            if len(chunk) > 1:
                ops.wgrad_plan(chunk)         # pixel splits against each problem's share of the grouped launch
            arr = (L.VamWgrad * len(chunk))(*chunk)
            n = len(chunk)
            self.keep.append(arr)
            self.keep += [c._ws for c in chunk if getattr(c, "_ws", None) is not None]     # pixel-split scratch (ops.wgrad_plan)
            fl = sum(2.0 * c.B * c.H * c.W * c.C * c.N * c.kh * c.kw for c in chunk)
            self.flops += fl
            self.meta.append({"kind": "wgrad", "flops": fl, "desc": f"{n}x wgrad [{chunk[0].C}->{chunk[0].N} k{chunk[0].kh}]"})
            self.branch_of.append(self.cur_branch)
            self.class_of.append(self.cur_class)
            self.steps.append(lambda arr=arr, n=n: L.check(lib.vam_conv_wgrad_group(arr, n, ops.stream_ptr()),
                                                           "vam_conv_wgrad_group"))

    def call(self, fn: Callable[[], None], desc: str = "op"):
        self.meta.append({"kind": "op", "flops": 0.0, "desc": desc})
        self.branch_of.append(self.cur_branch)
        self.class_of.append(self.cur_class)
        self.steps.append(fn)

    # ---- concurrency: independent chains are recorded on different branches (HIP streams) and
    # ordered by events; under capture they become parallel branches of the hipGraph.
    def branch(self, b: int):
        self.cur_branch = b

    def set_class(self, name: str):
        """Launch class of the steps recorded from here on (what the event profiler sums convolution time by)."""
        self.cur_class = L.PROF_CLASSES.index(name)

    def record(self) -> int:
        """Mark "everything recorded so far on the current branch is done"; returns an event id."""
        idx = self.n_events
        self.n_events += 1
        self.meta.append({"kind": "event", "flops": 0.0, "desc": f"record {idx}"})
        self.branch_of.append(self.cur_branch)
        self.class_of.append(self.cur_class)
        self.steps.append(("rec", idx))
        return idx

    def wait(self, idx: int):
        self.meta.append({"kind": "event", "flops": 0.0, "desc": f"wait {idx}"})
        self.branch_of.append(self.cur_branch)
        self.class_of.append(self.cur_class)
        self.steps.append(("wait", idx))

    def profile(self, reps: int = 3):
        """Per-step time (ms, best of reps) measured with events on the current stream."""
        best = [float("inf")] * len(self.steps)
        for _ in range(reps):
            evs = []
            for s in self.steps:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if not isinstance(s, tuple):      # single-stream replay: events are no-ops
                    s()
                b.record()
                evs.append((a, b))
            torch.cuda.current_stream().synchronize()
            best = [min(t, a.elapsed_time(b)) for t, (a, b) in zip(best, evs)]
        return [dict(m, ms=t) for m, t in zip(self.meta, best)]

    # ---- execution
    def run(self):
        self.run_range(0, len(self.steps))

    def run_range(self, a: int, b: int):
        """Steps [a, b) on the current stream and, for the other branches, on side streams ordered by the recorded events.
        A range is self-contained (it can be captured as one hipGraph): side streams are forked from the current stream
        at their first use and joined back at the end, so a wait on an event recorded before ``a`` is already satisfied."""
        main = torch.cuda.current_stream(self.device)
        prof = ops.prof_on()
        if self.n_events == 0 or prof:
            # (event profiler on: one stream, in recording order — a valid topological order — so that a launch's event
            # pair brackets that launch alone and not whatever the other branch runs beside it)
            for c, s in zip(self.class_of[a:b], self.steps[a:b]):
                if isinstance(s, tuple):
                    continue
                if prof:
                    ops.prof_set_class(c)
                s()
            return
        streams = {0: main}
        events: Dict[int, "torch.cuda.Event"] = {}
        start = None
        cur = 0
        try:
            for i in range(a, b):
                br, s = self.branch_of[i], self.steps[i]
                if br not in streams:
                    if br not in self._side:
                        self._side[br] = torch.cuda.Stream(device=self.device)
                    streams[br] = self._side[br]
                    if start is None:
                        start = torch.cuda.Event()
                        start.record(main)       # (main has only advanced since: a later fork point would do as well)
                    streams[br].wait_event(start)
                if br != cur:
                    torch.cuda.set_stream(streams[br])
                    cur = br
                if isinstance(s, tuple):
                    if s[0] == "rec":
                        events[s[1]] = torch.cuda.Event()
                        events[s[1]].record(streams[br])
                    elif s[1] in events:
                        streams[br].wait_event(events[s[1]])
                else:
                    s()
            for br, st in streams.items():
                if br != 0:
                    e = torch.cuda.Event()
                    e.record(st)
                    main.wait_event(e)
        finally:
            torch.cuda.set_stream(main)

    def graph_run(self):
        if self._graph is None:
            self.run()                       # warm every kernel (lazy code-object load) before capture
            torch.cuda.current_stream().synchronize()
            g = ops.Graph()
            g.capture(self.run)
            self._graph = g
        self._graph.launch()


# =============================================================================
# building blocks (each appends launches to a plan)
# =============================================================================
def _act_after(seq: nn.Sequential, i: int) -> int:
    nxt = seq[i + 1] if i + 1 < len(seq) else None
    if isinstance(nxt, Ly.GELU):
        return L.ACT_GELU
    if isinstance(nxt, Ly.LeakyReLU):
        return L.ACT_LEAKY
    return L.ACT_NONE


def conv_layers(seq: nn.Sequential):
    """[(layer, act)] for a conv/GELU/subpel Sequential."""
    out = []
    for i, m in enumerate(seq):
        if isinstance(m, (Ly.Conv2d, Ly.SubpelConv)):
            out.append((m, _act_after(seq, i)))
        elif not isinstance(m, Ly._Marker):
            raise TypeError(f"unexpected {type(m).__name__} in a conv stack")
    return out


def _out_extent(m, v: View):
    if isinstance(m, Ly.SubpelConv):
        return 2 * v.H, 2 * v.W, m.out_ch
    if m.is_rgb_s2d:                       # runs as a 3x3 s1 problem on the space-to-depth grid
        return v.H, v.W, m.out_channels
    s = m.stride
    k = m.kernel_size
    if s == 1:
        return v.H, v.W, m.out_channels
    return (v.H + 2 * (k // 2) - k) // s + 1, (v.W + 2 * (k // 2) - k) // s + 1, m.out_channels


# Plane tensors pay off where the consumer's tiles are small (latent-resolution stacks: the split is a sizeable share
# of their per-chunk work); on the big feature maps the 8-byte plane stores and 1.5x bytes cost more than they save.
import os as _os
P3_MAX_PIXELS = int(_os.environ.get("VAMPIC_P3_MAX_PIXELS", "16384"))       # (environment override: A/B measurements)


def lower_stack_heads(plan: Plan, stacks: Sequence[nn.Sequential], hyper_inputs: Sequence[View],
                      has_support: Sequence[bool]) -> Dict[int, tuple]:
    """The first layer of every slice stack reads cat(hyperprior tensor [c_head ch], supports...) (pic.py:528-529,
    598-599,635).  Its hyperprior part does not depend on the slice loop, so it is computed for ALL stacks up front in
    large grouped launches:  head_k = bias_k + conv(hyper; W_k[:, :c_head]);  inside the loop the first layer becomes
    act(head_k + conv(supports; W_k[:, c_head:])) — the same sum in a different association (the split is the same
    for encoder and decoder, so their bits still agree).  A stack without supports gets its complete first layer.
    Returns {id(stack): (view, complete)}."""
    heads: Dict[int, tuple] = {}
    probs = []
    for st, hv, sup in zip(stacks, hyper_inputs, has_support):
        m, act = conv_layers(st)[0]
        head, tail = m.packed_split(hv.C)
        assert (tail is not None) == bool(sup), "stack input width does not match its supports"
        o = plan.buf(hv.B, hv.H, hv.W, m.out_channels)
        probs.append(ops.conv_problem(head, [hv], o, L.ACT_NONE if sup else act))
        heads[id(st)] = (o, not sup)
    plan.conv(probs)
    return heads


def lower_stacks(plan: Plan, stacks: Sequence[nn.Sequential], inputs: Sequence[Sequence[View]],
                 outs: Sequence[Optional[View]], final: Optional[Sequence[dict]] = None,
                 heads: Optional[Dict[int, tuple]] = None) -> List[View]:
    """Run K structurally identical conv stacks in lockstep: layer d of every stack is one
    grouped launch.  ``final[k]`` = extra epilogue kwargs (act/pre/mul/post/post2) of the last
    layer of stack k.  ``heads`` (from :func:`lower_stack_heads`): the first layer's hyperprior part is already
    there — ``inputs[k]`` then lists the supports only.  Returns the output views."""
    K = len(stacks)
    lay = [conv_layers(s) for s in stacks]
    depth = len(lay[0])
    assert all(len(l) == depth for l in lay)
    cur: List[Sequence[View]] = [list(i) for i in inputs]
    res: List[View] = []
    # intermediates of a stack are read by the next layer only: with the split-operand kernel they are written as
    # bf16x3 planes by the producer (one split per element) instead of being re-split per tap and N tile downstream
    p3 = ops.split_mode()
    for d in range(depth):
        probs = []
        nxt = []
        if d == depth - 2 and depth >= 3 and plan._amax_pool is None and not plan.act16 and all(len(c) == 1 for c in cur):
            # the last two layers as one launch (csrc/stack_tail.hip) where every stack of the group qualifies
            fkw = [dict(final[k]) if (final is not None and final[k]) else {} for k in range(K)]
            if all(lay[k][d][1] == L.ACT_GELU and
                   ops.stack_tail_ok(cur[k][0], lay[k][d][0], lay[k][d + 1][0], outs[k],
                                     dict(fkw[k], act=fkw[k].get("act", lay[k][d + 1][1]))) for k in range(K)):
                tails = []
                for k in range(K):
                    x = cur[k][0]
                    o = outs[k] if outs[k] is not None else plan.buf(x.B, x.H, x.W, 32)
                    tails.append(ops.stack_tail_problem(x, lay[k][d][0].packed(), lay[k][d + 1][0].packed(), o,
                                                        fkw[k].get("act", lay[k][d + 1][1]), fkw[k].get("post"), fkw[k].get("post2")))
                    res.append(o)
                plan.stack_tail(tails)
                return res
        for k in range(K):
            m, act = lay[k][d]
            hd = heads.get(id(stacks[k])) if (heads is not None and d == 0) else None
            if hd is not None and hd[1]:             # complete first layer came with the heads
                assert not cur[k]
                nxt.append([hd[0]])
                continue
            v0 = cur[k][0]
            Ho, Wo, Co = _out_extent(m, v0)
            last = d == depth - 1
            if last and outs[k] is not None:
                o = outs[k]
            elif p3 and not last and Co % 8 == 0 and m.packed().ps2_cq == 0 and v0.B * Ho * Wo <= P3_MAX_PIXELS and \
                    not (plan.act16 and Ho * Wo >= plan.act16_min_hw) and not isinstance(v0, ops.View16):
                o = plan.buf3(v0.B, Ho, Wo, Co)
            else:
                o = plan.buf(v0.B, Ho, Wo, Co)
            kw = {}
            if last and final is not None and final[k]:
                kw = dict(final[k])
                act = kw.pop("act", act)
            if hd is not None:                        # supports only; the hyperprior part (with the bias) enters as `pre`
                assert "pre" not in kw
                c_head = m.in_channels - sum(v.C for v in cur[k])
                probs.append(ops.conv_problem(m.packed_split(c_head)[1], cur[k], o, act, pre=hd[0], **kw))
            else:
                probs.append(ops.conv_problem(plan.pk(m, cur[k], o, [kw.get(a_) for a_ in ("pre", "mul", "post", "post2")]),
                                              cur[k], o, act, **kw))
            nxt.append([o])
            if last:
                res.append(o)
        if probs:
            plan.conv(probs)
        cur = nxt
    return res


def lower_gdn(plan: Plan, mods: Sequence[Ly.GDN], xs: Sequence[View], outs: Sequence[Optional[View]]) -> List[View]:
    probs, res = [], []
    for m, x, o in zip(mods, xs, outs):
        o = o if o is not None else plan.buf(x.B, x.H, x.W, x.C)
        probs.append(ops.conv_problem(plan.pk(m, [x], o), [x], o, L.ACT_SQRT if m.inverse else L.ACT_RSQRT, mul=x,
                                      flags=L.CONV_SQUARE_IN))
        res.append(o)
    plan.conv(probs)
    return res


def lower_residual_units(plan: Plan, mods: Sequence[Ly.ResidualUnit], xs: Sequence[View]) -> List[View]:
    """out = GELU(conv(x) + x)  (layers/layers.py:43-48), K units in lockstep.  Shapes with a fused kernel
    (csrc/resunit.hip) take ONE launch per group of units, both C/2-channel intermediates staying in LDS; the rest
    run as three grouped conv launches — the same arithmetic, bit for bit."""
    if all(ops.resunit_supported(x) for x in xs) and len({type(x) for x in xs}) == 1:
        b16 = isinstance(xs[0], ops.View16)           # bf16-storage mode: bf16 tensors, bf16-packed weights
        outs = [plan.buf(x.B, x.H, x.W, x.C) for x in xs]
        if all(type(o) is type(x) for o, x in zip(outs, xs)):
            plan.resunit([ops.resunit_problem(m.conv[0].packed(b16), m.conv[2].packed(b16), m.conv[4].packed(b16), x, o)
                          for m, x, o in zip(mods, xs, outs)])
            return outs
    fin = [dict(act=L.ACT_GELU, pre=x) for x in xs]
    return lower_stacks(plan, [m.conv for m in mods], [[x] for x in xs], [None] * len(mods), fin)


def lower_win_attention(plan: Plan, mods: Sequence[Ly.WinBasedAttention], xs: Sequence[View]) -> List[View]:
    """x + proj(window_attention(qkv(x)))  (layers/win_attention.py:153-207)."""
    qkvs = [plan.buf32(x.B, x.H, x.W, 3 * x.C) for x in xs]                  # q, k, v and the attention output are fp32
    plan.conv([ops.conv_problem(plan.pk(m.attn.qkv, [x], q), [x], q) for m, x, q in zip(mods, xs, qkvs)])
    atts = [plan.buf32(x.B, x.H, x.W, x.C) for x in xs]
    for m, q, a, x in zip(mods, qkvs, atts, xs):
        tab = m.attn.relative_position_bias_table
        plan.keep.append(tab)
        plan.call(lambda q=q, a=a, tab=tab, C_=x.C, h=m.num_heads, ws=m.window_size, sh=m.shift_size:
                  ops.win_attention(q, a, tab, C_, h, ws, sh))
    outs = [plan.buf(x.B, x.H, x.W, x.C) for x in xs]
    plan.conv([ops.conv_problem(plan.pk(m.attn.proj, [a], o, [x]), [a], o, post=x) for m, a, o, x in zip(mods, atts, outs, xs)])
    return outs


def lower_attention_blocks(plan: Plan, mods: Sequence[Ly.Win_noShift_Attention], xs: Sequence[View],
                           outs: Sequence[Optional[View]]) -> List[View]:
    """out = a * sigmoid(b) + x  (layers/layers.py:68-74); branch a and branch b's residual
    units are independent, so they share launches."""
    K = len(mods)
    b = lower_win_attention(plan, [m.conv_b[0] for m in mods], xs)
    a = list(xs)
    for i in range(3):
        r = lower_residual_units(plan, [m.conv_a[i] for m in mods] + [m.conv_b[i + 1] for m in mods], a + b)
        a, b = r[:K], r[K:]
    res, probs = [], []
    for m, av, bv, x, o in zip(mods, a, b, xs, outs):
        o = o if o is not None else plan.buf(x.B, x.H, x.W, x.C)
        probs.append(ops.conv_problem(plan.pk(m.conv_b[4], [bv], o, [av, x]), [bv], o, L.ACT_SIGMOID, mul=av, post=x))
        res.append(o)
    plan.conv(probs)
    return res


def lower_deconv(plan: Plan, mods: Sequence[Ly.ConvTranspose2d], xs: Sequence[View], outs: Sequence[Optional[View]],
                 act: int = L.ACT_NONE, out_nchw: Optional[Sequence[torch.Tensor]] = None) -> List[Optional[View]]:
    probs, res = [], []
    for i, (m, x) in enumerate(zip(mods, xs)):
        o = outs[i]
        nchw = out_nchw[i] if out_nchw is not None else None
        if o is None and nchw is None:
            o = plan.buf(x.B, 2 * x.H, 2 * x.W, m.out_channels)
        any16 = isinstance(x, ops.View16) or isinstance(o, ops.View16)
        for pk in m.packed(any16):
            probs.append(ops.conv_problem(pk, [x], o, act, out_nchw=nchw))
        res.append(o)
    plan.conv(probs)
    return res


def lower_g_a(plan: Plan, encs: Sequence[nn.Sequential], x_s2d: View, ys: Sequence[View]):
    """models/builder.py:43-53, both encoders in lockstep; ys = 320-channel windows of y."""
    K = len(encs)
    t = lower_stacks(plan, [nn.Sequential(e[0]) for e in encs], [[x_s2d]] * K, [None] * K)
    t = lower_gdn(plan, [e[1] for e in encs], t, [None] * K)
    t = lower_stacks(plan, [nn.Sequential(e[2]) for e in encs], [[v] for v in t], [None] * K)
    t = lower_gdn(plan, [e[3] for e in encs], t, [None] * K)
    t = lower_attention_blocks(plan, [e[4] for e in encs], t, [None] * K)
    t = lower_stacks(plan, [nn.Sequential(e[5]) for e in encs], [[v] for v in t], [None] * K)
    t = lower_gdn(plan, [e[6] for e in encs], t, [None] * K)
    t = lower_stacks(plan, [nn.Sequential(e[7]) for e in encs], [[v] for v in t], [None] * K)
    return lower_attention_blocks(plan, [e[8] for e in encs], t, ys)


def lower_g_s(plan: Plan, decs: Sequence[nn.Sequential], ys: Sequence[View], x_hat: Sequence[torch.Tensor],
              clamp: bool = True):
    """models/builder.py:8-18 (+ clamp_(0,1) of pic.py:558,651), result stored NCHW."""
    K = len(decs)
    t = lower_attention_blocks(plan, [d[0] for d in decs], ys, [None] * K)
    t = lower_deconv(plan, [d[1] for d in decs], t, [None] * K)
    t = lower_gdn(plan, [d[2] for d in decs], t, [None] * K)
    t = lower_deconv(plan, [d[3] for d in decs], t, [None] * K)
    t = lower_gdn(plan, [d[4] for d in decs], t, [None] * K)
    t = lower_attention_blocks(plan, [d[5] for d in decs], t, [None] * K)
    t = lower_deconv(plan, [d[6] for d in decs], t, [None] * K)
    t = lower_gdn(plan, [d[7] for d in decs], t, [None] * K)
    lower_deconv(plan, [d[8] for d in decs], t, [None] * K, L.ACT_CLAMP01 if clamp else L.ACT_NONE, out_nchw=x_hat)


def lower_rem_resblocks(plan: Plan, blocks: Sequence[Ly.ResidualBlock], ins: Sequence[Sequence[View]],
                        outs: Sequence[Optional[View]], final: Optional[Sequence[dict]] = None) -> List[View]:
    """LeakyReLU(conv2(LeakyReLU(conv1(x)))) + skip(x)   (layers/rem.py:52-66), K blocks in lockstep.
    ``final``: extra mul/post applied after the residual add is NOT expressible in one
    epilogue, so callers needing it (the REM tail) pass it to :func:`lower_rem_block`."""
    K = len(blocks)
    x0 = [i[0] for i in ins]
    h = [plan.buf(v.B, v.H, v.W, b.conv1.out_channels) for b, v in zip(blocks, x0)]
    probs = [ops.conv_problem(b.conv1.packed(), i, o, L.ACT_LEAKY) for b, i, o in zip(blocks, ins, h)]
    idn: List[View] = []
    for b, i in zip(blocks, ins):
        if b.skip is not None:
            s = plan.buf(i[0].B, i[0].H, i[0].W, b.skip.out_channels)
            probs.append(ops.conv_problem(b.skip.packed(), i, s))
            idn.append(s)
        else:
            assert len(i) == 1
            idn.append(i[0])
    plan.conv(probs)
    res, probs = [], []
    for k, (b, hv) in enumerate(zip(blocks, h)):
        o = outs[k] if outs[k] is not None else plan.buf(hv.B, hv.H, hv.W, b.conv2.out_channels)
        probs.append(ops.conv_problem(b.conv2.packed(), [hv], o, L.ACT_LEAKY, post=idn[k]))
        res.append(o)
    plan.conv(probs)
    return res


def lower_rem_blocks(plan: Plan, mods: Sequence[Ly.LatentRateReduction], y_cks: Sequence[View],
                     ep_bases: Sequence[Sequence[View]], ep_progs: Sequence[Sequence[View]], atts: Sequence[View],
                     outs: Sequence[Sequence[View]]):
    """K REM blocks in lockstep:  res = identity + enc(cat(f_latent, f_ent_base, f_ent_prog)) * att_mask
    (layers/rem.py:130-141).  ep_progs[k] = [mu, sigma] windows with ``mu_std`` (identity = their concat), [sigma]
    without; atts[k] = the N-channel mask, applied to every N-channel part (rem_pic.py:194-195);
    outs[k] = [mu', sigma'] (or [sigma']) windows."""
    K = len(mods)
    m0 = mods[0]
    assert all(m.mu_std == m0.mu_std for m in mods)
    parts = 2 if m0.mu_std else 1
    assert all(len(e) == parts for e in ep_progs) and all(len(o) == parts for o in outs)
    names = ["enc_base_rep", "enc_progressive_entropy_params", "enc_base_entropy_params"]
    cur: List[Sequence[View]] = [[y] for y in y_cks] + [list(e) for e in ep_progs] + [list(e) for e in ep_bases]
    for d in range(len(m0.enc_base_rep)):
        blocks = [getattr(m, n)[d] for n in names for m in mods]
        r = lower_rem_resblocks(plan, blocks, cur, [None] * (3 * K))
        cur = [[v] for v in r]
    t: List[Sequence[View]] = [[cur[k][0], cur[2 * K + k][0], cur[K + k][0]] for k in range(K)]  # latent, base, prog
    for d in range(len(m0.enc)):
        r = lower_rem_resblocks(plan, [m.enc[d] for m in mods], t, [None] * K)
        t = [[v] for v in r]
    N = m0.dim_block
    pk = _identity_pack(N, t[0][0].buf.device)
    probs = []
    for k in range(K):
        ret = t[k][0]
        assert ret.C == parts * N
        for half, (idv, o) in enumerate(zip(ep_progs[k], outs[k])):
            probs.append(ops.conv_problem(pk, [ret.window(half * N, N)], o, L.ACT_NONE, mul=atts[k], post=idv))
    plan.conv(probs)


# =============================================================================
# REM fine-tune: taped forward + backward lowering            (train.py:223-226, training/step.py:56-95)
# =============================================================================
class TrainPacks:
    """Persistent packed weights of the TRAINED convolutions (forward form and data-gradient form), refreshed
    in place at the head of every step: the optimiser changes the parameters between steps while the plan's
    pre-marshalled launches keep their pointers."""

    def __init__(self, convs: Sequence[Ly.Conv2d], need_dgrad: Sequence[Ly.Conv2d]):
        self.convs = list(convs)
        self.f = {id(c): ops.pack_conv(c.weight, c.bias, 1) for c in self.convs}
        self.d = {id(c): ops.pack_conv_dgrad(c.weight) for c in need_dgrad}
        self._dg = list(need_dgrad)

    def fwd(self, c) -> ops.Packed:
        return self.f[id(c)]

    def dgrad(self, c) -> ops.Packed:
        return self.d[id(c)]

    def record_refresh(self, plan: Plan):
        def refresh():
            with ops.pack_batch():               # one grouped launch per 32 repacks instead of one launch each
                for c in self.convs:
                    ops.repack_conv(c.weight, c.bias, self.f[id(c)])
                for c in self._dg:
                    ops.repack_conv(c.weight, None, self.d[id(c)], dgrad=True)
        plan.call(refresh, f"repack {len(self.convs)}+{len(self._dg)} trained convs")


def rem_trained_convs(mods: Sequence[Ly.LatentRateReduction]):
    """(all convs, convs whose data gradient is needed) of K REM blocks.  The first ResidualBlock of each of the
    three input branches feeds from frozen tensors, so its conv1 / skip need no data gradient."""
    allc, dg = [], []
    for m in mods:
        for name in ("enc_base_rep", "enc_progressive_entropy_params", "enc_base_entropy_params", "enc"):
            for d, rb in enumerate(getattr(m, name)):
                cs = [rb.conv1, rb.conv2] + ([rb.skip] if rb.skip is not None else [])
                allc += cs
                dg.append(rb.conv2)
                if name == "enc" or d > 0:
                    dg += [rb.conv1] + ([rb.skip] if rb.skip is not None else [])
    return allc, dg


def _rb_forward_taped(plan: Plan, blocks, ins, packs: TrainPacks):
    """ResidualBlock forward keeping what the backward needs: h1a = LeakyReLU(conv1 x), o2 = LeakyReLU(conv2 h1a)
    (the sign of a LeakyReLU's output is the sign of its input), out = o2 + skip(x)."""
    h = [plan.buf(i[0].B, i[0].H, i[0].W, b.conv1.out_channels) for b, i in zip(blocks, ins)]
    probs = [ops.conv_problem(packs.fwd(b.conv1), i, o, L.ACT_LEAKY) for b, i, o in zip(blocks, ins, h)]
    idn = []
    for b, i in zip(blocks, ins):
        if b.skip is not None:
            sv = plan.buf(i[0].B, i[0].H, i[0].W, b.skip.out_channels)
            probs.append(ops.conv_problem(packs.fwd(b.skip), i, sv))
            idn.append(sv)
        else:
            assert len(i) == 1
            idn.append(i[0])
    plan.conv(probs)
    o2 = [plan.buf(v.B, v.H, v.W, b.conv2.out_channels) for b, v in zip(blocks, h)]
    plan.conv([ops.conv_problem(packs.fwd(b.conv2), [hv], o, L.ACT_LEAKY) for b, hv, o in zip(blocks, h, o2)])
    outs = [plan.buf(v.B, v.H, v.W, v.C) for v in o2]
    for a, b_, o in zip(o2, idn, outs):
        plan.call(lambda a=a, b_=b_, o=o: ops.add(b_, a, o), "rb residual add")     # same order as post + act(.)
    recs = [dict(block=b, x=list(i), h1a=hv, o2=o) for b, i, hv, o in zip(blocks, ins, h, o2)]
    return outs, recs


def lower_rem_blocks_train(plan: Plan, mods, y_cks, ep_bases, ep_progs, atts, outs, packs: TrainPacks) -> dict:
    """Forward of K REM blocks (same arithmetic as :func:`lower_rem_blocks`) recording a tape."""
    K = len(mods)
    m0 = mods[0]
    assert len({m.mu_std for m in mods}) == 1         # mu_std = False: ep_progs / outs hold the scale alone (rem.py:86,100)
    names = ["enc_base_rep", "enc_progressive_entropy_params", "enc_base_entropy_params"]
    cur = [[y] for y in y_cks] + [list(e) for e in ep_progs] + [list(e) for e in ep_bases]
    tape = {"branch": [], "enc": [], "K": K}
    for d in range(len(m0.enc_base_rep)):
        blocks = [getattr(m, n)[d] for n in names for m in mods]
        r, recs = _rb_forward_taped(plan, blocks, cur, packs)
        tape["branch"].append(recs)
        cur = [[v] for v in r]
    t = [[cur[k][0], cur[2 * K + k][0], cur[K + k][0]] for k in range(K)]       # latent, base, prog (rem.py:137)
    for d in range(len(m0.enc)):
        r, recs = _rb_forward_taped(plan, [m.enc[d] for m in mods], t, packs)
        tape["enc"].append(recs)
        t = [[v] for v in r]
    N = m0.dim_block
    pk = _identity_pack(N, t[0][0].buf.device)
    probs = []
    for k in range(K):
        ret = t[k][0]
        for half, (idv, o) in enumerate(zip(ep_progs[k], outs[k])):
            probs.append(ops.conv_problem(pk, [ret.window(half * N, N)], o, L.ACT_NONE, mul=atts[k], post=idv))
    plan.conv(probs)
    return tape


def _rb_backward(plan: Plan, recs, d_outs, need_dx: bool, packs: TrainPacks, grads: Dict[int, torch.Tensor]):
    """Backward of K ResidualBlocks in lockstep.  ``grads[id(param)]`` are the gradient buffers (parameter
    layout).  Returns dL/dx per block (one View over the concatenated input channels) or None."""
    g = lambda p: grads[id(p)]
    dh2 = [plan.buf(r["o2"].B, r["o2"].H, r["o2"].W, r["o2"].C) for r in recs]
    wg = []
    for r, do, t in zip(recs, d_outs, dh2):
        plan.call(lambda r=r, do=do, t=t: ops.leaky_bwd(r["o2"], do, t), "leaky bwd")
        b = r["block"]
        wg += ops.wgrad_problems([r["h1a"]], t, g(b.conv2.weight), g(b.conv2.bias))
    plan.wgrad(wg)
    dh1a = [plan.buf(r["h1a"].B, r["h1a"].H, r["h1a"].W, r["h1a"].C) for r in recs]
    plan.conv([ops.conv_problem(packs.dgrad(r["block"].conv2), [t], o) for r, t, o in zip(recs, dh2, dh1a)])
    dh1 = [plan.buf(v.B, v.H, v.W, v.C) for v in dh1a]
    wg = []
    for r, a, t, do in zip(recs, dh1a, dh1, d_outs):
        b = r["block"]
        plan.call(lambda r=r, a=a, t=t: ops.leaky_bwd(r["h1a"], a, t), "leaky bwd")
        wg += ops.wgrad_problems(r["x"], t, g(b.conv1.weight), g(b.conv1.bias))
        if b.skip is not None:
            wg += ops.wgrad_problems(r["x"], do, g(b.skip.weight), g(b.skip.bias))
    plan.wgrad(wg)
    if not need_dx:
        return None
    cin = [sum(v.C for v in r["x"]) for r in recs]
    dx = [plan.buf(r["o2"].B, r["o2"].H, r["o2"].W, c) for r, c in zip(recs, cin)]
    via = []
    sk = [(r, do) for r, do in zip(recs, d_outs) if r["block"].skip is not None]
    if sk:
        tmp = {id(r): plan.buf(r["o2"].B, r["o2"].H, r["o2"].W, c) for (r, _), c in zip(sk, [sum(v.C for v in r["x"]) for r, _ in sk])}
        plan.conv([ops.conv_problem(packs.dgrad(r["block"].skip), [do], tmp[id(r)]) for r, do in sk])
    for r, do in zip(recs, d_outs):
        via.append(tmp[id(r)] if r["block"].skip is not None else do)
    plan.conv([ops.conv_problem(packs.dgrad(r["block"].conv1), [t], o, post=p_)
               for r, t, o, p_ in zip(recs, dh1, dx, via)])
    return dx


def lower_rem_backward(plan: Plan, tape: dict, mods, d_mu: Sequence[View], d_sigma: Sequence[View], atts: Sequence[View],
                       packs: TrainPacks, grads: Dict[int, torch.Tensor]):
    """dL/d(REM parameters) from dL/d(mu', sigma') of K slices: res = identity + ret * att  (rem.py:139-141),
    so d ret = d res * att; nothing upstream of the REM inputs is trainable (train.py:223-226)."""
    K = tape["K"]
    N = mods[0].dim_block
    mu_std = mods[0].mu_std                         # False: the block refines the scale only, ret has N channels and dL/dmu stops here
    d_ret = [plan.buf(d_sigma[k].B, d_sigma[k].H, d_sigma[k].W, (2 if mu_std else 1) * N) for k in range(K)]
    for k in range(K):
        if mu_std:
            plan.call(lambda k=k: ops.mul(d_mu[k], atts[k], d_ret[k].window(0, N)), "d ret (mu half)")
        plan.call(lambda k=k: ops.mul(d_sigma[k], atts[k], d_ret[k].window(N if mu_std else 0, N)), "d ret (sigma half)")
    d = d_ret
    for recs in reversed(tape["enc"]):
        d = _rb_backward(plan, recs, d, True, packs, grads)
    # d[k]: gradient of cat(f_latent, f_base, f_prog); branch order in the tape: latent x K, prog x K, base x K
    d_branch = [d[k].window(0, N) for k in range(K)] + [d[k].window(2 * N, N) for k in range(K)] + \
               [d[k].window(N, N) for k in range(K)]
    depth = len(tape["branch"])
    for i, recs in enumerate(reversed(tape["branch"])):
        d_branch = _rb_backward(plan, recs, d_branch, i < depth - 1, packs, grads)


_ONE_BY_ONE: Dict[tuple, ops.Packed] = {}


def _identity_pack(C: int, device) -> ops.Packed:
    """1x1 identity weights: lets ``o = i + r * a`` reuse the conv epilogue (exact: 1.0*r plus zeros)."""
    key = (C, str(device))
    if key not in _ONE_BY_ONE:
        w = torch.eye(C, dtype=torch.float32, device=device).reshape(C, C, 1, 1)
        _ONE_BY_ONE[key] = ops.pack_conv(w, None, 1)
    return _ONE_BY_ONE[key]


# =============================================================================
# stand-alone execution of one module on NCHW tensors (reference harness surface)
# =============================================================================
def run_module(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    Ly._no_autograd(x, *list(m.parameters()))
    L.require_gpu()
    plan = Plan(x.device)
    if isinstance(m, Ly.TransformStack) and isinstance(m[0], Ly.Conv2d) and m[0].is_rgb_s2d:
        xin = ops.s2d_input(x)
        y = plan.buf(xin.B, xin.H // 8, xin.W // 8, m[7].out_channels)
        lower_g_a(plan, [m], xin, [y])
        plan.run()
        return y.torch_nchw()
    if isinstance(m, Ly.TransformStack):
        v = ops.from_nchw(x)
        out = torch.empty((v.B, m[8].out_channels, v.H * 16, v.W * 16), dtype=torch.float32, device=x.device)
        lower_g_s(plan, [m], [v], [out], clamp=False)
        plan.run()
        return out
    if isinstance(m, Ly.Conv2d) and m.is_rgb_s2d:
        xin = ops.s2d_input(x)
        o = lower_stacks(plan, [nn.Sequential(m)], [[xin]], [None])[0]
    else:
        v = ops.from_nchw(x)
        if isinstance(m, (Ly.Conv2d, Ly.SubpelConv)):
            o = lower_stacks(plan, [nn.Sequential(m)], [[v]], [None])[0]
        elif isinstance(m, Ly.ConvStack):
            c_head = getattr(m, "c_head", None)
            if c_head is not None and m[0].in_channels > c_head and v.C == m[0].in_channels:
                # slice stack called on cat(hyper, supports): same association of the first-layer sum as the fused plans
                heads = lower_stack_heads(plan, [m], [v.window(0, c_head)], [True])
                o = lower_stacks(plan, [m], [[v.window(c_head, v.C - c_head)]], [None], heads=heads)[0]
            else:
                o = lower_stacks(plan, [m], [[v]], [None])[0]
        elif isinstance(m, Ly.ConvTranspose2d):
            o = lower_deconv(plan, [m], [v], [None])[0]
        elif isinstance(m, Ly.GDN):
            o = lower_gdn(plan, [m], [v], [None])[0]
        elif isinstance(m, Ly.ResidualUnit):
            o = lower_residual_units(plan, [m], [v])[0]
        elif isinstance(m, Ly.WinBasedAttention):
            o = lower_win_attention(plan, [m], [v])[0]
        elif isinstance(m, Ly.Win_noShift_Attention):
            o = lower_attention_blocks(plan, [m], [v], [None])[0]
        else:
            raise TypeError(f"no lowering for {type(m).__name__}")
    plan.run()
    return o.torch_nchw()


def run_rem_module(m: Ly.LatentRateReduction, x_base, ep_base, ep_prog, att_mask) -> torch.Tensor:
    Ly._no_autograd(x_base, ep_base, ep_prog)
    L.require_gpu()
    plan = Plan(x_base.device)
    N = m.dim_block
    vb, ve, vp, va = (ops.from_nchw(t) for t in (x_base, ep_base, ep_prog, att_mask))
    out = plan.buf(vb.B, vb.H, vb.W, 2 * N)
    lower_rem_blocks(plan, [m], [vb], [[ve]], [[vp.window(0, N), vp.window(N, N)]], [va.window(0, N)],
                     [[out.window(0, N), out.window(N, N)]])
    plan.run()
    return out.torch_nchw()
