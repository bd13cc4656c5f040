"""Plan / hipGraph lifetime (round-2 host segfault): dropping plans between replays.  Round 3 found
the crash again, deterministically, behind hipGraphExecDestroy itself (ops.Graph docstring): dropped graphs are retired."""
import copy
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu

import vampic.synth as synth     # noqa: E402
from vampic import ops           # noqa: E402


def test_dropping_plans_between_graph_replays(gpu_model):
    """The sequence of the crash record (gpurun_out/full_gpu_r02d.log: `codec` fixture -> net.update() -> _plans.clear()
    with replays of the dropped plans' graphs possibly still running, then capture + launch of a new plan): dropped
    executable graphs are only PARKED (no HIP call from a destructor / during a capture) and retired at the next plan
    entry point after their stream has been synchronised.  Runs once — no loops."""
    net = copy.deepcopy(gpu_model[0])
    x = synth.synth_image(2, 64, 128, seed=5).cuda()
    with torch.no_grad():
        a = net.forward_single_quality(x, 2.5)            # capture + replay on the model's private stream
        a0 = net.forward_single_quality(x, 0)
        gc.collect()                                      # plans of earlier tests' models: parked by their destructors
        ops.drain_graveyard()
        assert ops.graveyard_size() == 0
        net.update()                                      # drops both plans, replays possibly in flight
        parked = ops.graveyard_size()
        assert parked >= 2, "update() must hand the plans' graphs to the deferred-destroy list explicitly"
        gc.collect()                                      # the collector finding the dropped plans' cycles calls no HIP and
        assert ops.graveyard_size() == parked             # finds nothing left to park
        b = net.forward_single_quality(x, 2.5)            # entry point: drain (sync the old stream, destroy), capture, replay
        assert ops.graveyard_size() == 0
        assert torch.equal(a["x_hat"], b["x_hat"]) and torch.equal(a["mask"], b["mask"])
        net.load_state_dict(net.state_dict())             # the other two paths that drop plans
        c = net.forward_single_quality(x, 0)
        net.float()
        d = net.forward_single_quality(x, 2.5)
        assert torch.equal(a0["x_hat"], c["x_hat"]) and torch.equal(a["x_hat"], d["x_hat"])
    torch.cuda.synchronize()
    ops.drain_graveyard()
    assert ops.graveyard_size() == 0


def test_many_dropped_graphs_then_a_new_plan(gpu_model):
    """The shape of the round-3 crash: a model that has replayed plans of several shapes and qualities (about twenty
    executable graphs) drops them all in ``update()``; the next call builds, captures and replays a new plan.  With
    hipGraphExecDestroy in ``drain_graveyard`` this sequence faulted inside hipGraphLaunch at the end of the order
    ops -> model -> golden -> config_variants -> bitstream; with the handles retired it must simply work."""
    net = copy.deepcopy(gpu_model[0])
    xs = [synth.synth_image(1, 64, 64, seed=1).cuda(), synth.synth_image(1, 64, 128, seed=2).cuda(),
          synth.synth_image(2, 64, 64, seed=3).cuda()]
    before = ops.retired_graphs()
    with torch.no_grad():
        ref = net.forward_single_quality(xs[1], 2.5)
        for x in xs:
            for q in (0, 0.25, 0.5, 1, 2.5, 5, 10):
                net.forward_single_quality(x, q)
        net.update()
        assert ops.graveyard_size() >= 15
        again = net.forward_single_quality(xs[1], 2.5)
        assert ops.graveyard_size() == 0 and ops.retired_graphs() >= before + 15
        enc = net.compress(xs[1], quality=2.5)
        dec = net.decompress(enc["strings"], enc["shape"], quality=2.5)
    assert torch.equal(ref["x_hat"], again["x_hat"]) and torch.equal(dec["x_hat"], again["x_hat"])
