"""Library-level half of the graph-destroy probe: the shape of the round-3 crash in one short
process — a model replays plans of three shapes x seven qualities (~21 executable graphs), ``update()`` drops them, the
next call drains the graveyard (VAMPIC_GRAPH_DESTROY=1: hipGraphExecDestroy of each), captures and launches a new plan,
then compress / decompress build more.  Three rounds.  Environment toggles the ingredients:
    VAMPIC_GRAPH_DESTROY=1          destroy instead of retire (the call that preceded the fault)
    VAMPIC_GRAPH_KEEP_TEMPLATE=1    keep the template hipGraph_t until its executable graph is destroyed
Run under rocgdb for a native backtrace:  rocgdb -batch -ex run -ex bt --args python scratch/probe/graph_destroy_lib.py
"""
import argparse
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import vampic                     # noqa: E402
import vampic.synth as synth      # noqa: E402
from vampic import ops            # noqa: E402

args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True,
                          dim_chunk=32, division_dimension=[320, 640], mask_policy="point-based-std",
                          support_progressive_slices=5, delta_encode=True, total_mu_rep=True, all_scalable=True)
net = vampic.get_model(args, "cpu").eval()
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0))
net = net.cuda()
net.update()
xs = [synth.synth_image(1, 64, 64, seed=1).cuda(), synth.synth_image(1, 64, 128, seed=2).cuda(),
      synth.synth_image(2, 64, 64, seed=3).cuda()]
print("mode: destroy =", os.environ.get("VAMPIC_GRAPH_DESTROY", "0"), " keep template =", os.environ.get("VAMPIC_GRAPH_KEEP_TEMPLATE", "0"),
      flush=True)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    ref = net.forward_single_quality(xs[1], 2.5)
    for r in range(rounds):
        for x in xs:
            for q in (0, 0.25, 0.5, 1, 2.5, 5, 10):
                net.forward_single_quality(x, q)
        net.update()
        print(f"round {r}: {ops.graveyard_size()} dropped graphs parked, {ops.retired_graphs()} retired so far", flush=True)
        again = net.forward_single_quality(xs[1], 2.5)
        enc = net.compress(xs[1], quality=2.5)
        dec = net.decompress(enc["strings"], enc["shape"], quality=2.5)
        torch.cuda.synchronize()
        assert torch.equal(ref["x_hat"], again["x_hat"]) and torch.equal(dec["x_hat"], again["x_hat"])
        print(f"round {r}: new plans captured and launched, results identical", flush=True)
print("done: no fault", flush=True)
