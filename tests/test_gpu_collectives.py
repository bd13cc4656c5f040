"""The exchange steps of the training schedules on a REAL device: with VAMPIC_FORCE_COLLECTIVES=1 a
1-rank ``nccl`` (= RCCL) process group on the single GPU runs communicator init, the 22 bucketed all-reduces of the
first-stage step on the communication stream behind events of the backward's graph segments, ``work.wait()``, the
division and the clip after the last bucket — everything an N-rank job runs except the wire.  Sums over one rank are the
identity, so the gradients must equal the no-collective step's bit for bit."""
import argparse
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import vampic                              # noqa: E402
import vampic.synth as synth               # noqa: E402
from vampic import sharding                # noqa: E402
from conftest import README_ARGS           # noqa: E402


def _model():
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0))
    return net.cuda().train()


def _step(net, x, noise):
    from vampic.finetune import ScalableRateDistortionLoss
    net.zero_grad(set_to_none=True)
    out = net(x, quality=[0, 10], training=True, noise=noise)
    c = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")(out, x)
    c["loss"].backward()
    return float(c["loss"]), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()


@pytest.fixture
def single_rank_group(monkeypatch):
    import torch.distributed as dist
    monkeypatch.setenv("VAMPIC_FORCE_COLLECTIVES", "1")
    assert sharding.init_single_rank_group(torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_forced_collective_first_train_step_is_bit_identical(single_rank_group):
    dist = single_rank_group
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and sharding.collectives_active()
    x = synth.synth_image(2, 64, 64, seed=8).cuda()
    noise = {"y": synth.uniform((2, 640, 4, 4), 301) - 0.5, "z": synth.uniform((2, 192, 1, 1), 302) - 0.5}
    net = _model()
    net.use_graph = True
    os.environ["VAMPIC_FORCE_COLLECTIVES"] = "0"
    loss0, g0 = _step(net, x, noise)                      # no reducer, no collective
    os.environ["VAMPIC_FORCE_COLLECTIVES"] = "1"
    net.grad_reducer = sharding.BucketReducer()
    loss1, g1 = _step(net, x, noise)                      # segmented backward graphs + one RCCL all-reduce per bucket
    loss2, g2 = _step(net, x, noise)                      # and again: replay of the segmented graphs
    plan = next(p for k, p in net._plans.items() if k[0] == "full_train")
    nb = len(plan.bucket_bounds)
    assert nb >= 20 and [i for i, _ in net.grad_reducer.log] == list(range(nb)) * 2
    assert [n for _, n in net.grad_reducer.log[:nb]] == [hi - lo for lo, hi in plan.bucket_bounds]
    assert net.grad_reducer.comm is not None and not net.grad_reducer.pending
    assert loss0 == loss1 == loss2
    assert torch.equal(g0, g1) and torch.equal(g0, g2)
    # the other exchange helpers on the same communicator
    assert sharding.max_over_ranks(1.5, "cuda") == 1.5
    assert sharding.sum_over_ranks([2.0, 3.0], "cuda") == [2.0, 3.0]
    import random
    assert 0 <= sharding.broadcast_choice(7, random.Random(3), "cuda") < 7
    ps = [torch.nn.Parameter(torch.randn(5, 3, device="cuda")), torch.nn.Parameter(torch.randn(4, device="cuda"))]
    ps[0].grad = torch.randn(5, 3, device="cuda")
    want = ps[0].grad.clone()
    assert sharding.all_reduce_gradients(ps) == 4 * (15 + 4 + 2)
    assert torch.equal(ps[0].grad, want) and ps[1].grad is None


def test_flat_clip_matches_torch_clip():
    """finetune.clip_grad_norm_: one reduction over the flat buffer the first-stage backward hands its gradients out of
    == torch.nn.utils.clip_grad_norm_ over the 1065 tensors (training/step.py:98)."""
    from vampic import finetune as ft
    x = synth.synth_image(2, 64, 64, seed=8).cuda()
    noise = {"y": synth.uniform((2, 640, 4, 4), 301) - 0.5, "z": synth.uniform((2, 192, 1, 1), 302) - 0.5}
    net = _model()
    _step(net, x, noise)
    plan = next(p for k, p in net._plans.items() if k[0] == "full_train")
    flat = plan.handout[0]
    assert all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    ref = [p.grad.detach().clone() for p in net.parameters()]
    want_norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.double()) for g in ref]))
    got_norm = ft.clip_grad_norm_(net, 1.0)
    assert abs(float(got_norm) - float(want_norm)) <= 1e-5 * float(want_norm)
    scale = min(1.0, 1.0 / (float(want_norm) + 1e-6))
    for p, g in zip(net.parameters(), ref):
        assert torch.allclose(p.grad, g * scale, rtol=1e-5, atol=1e-12)
    # a frozen subset takes torch's own routine and still clips
    list(net.parameters())[0].grad = None
    assert float(ft.clip_grad_norm_(net, 1.0)) <= 1.0 + 1e-4
