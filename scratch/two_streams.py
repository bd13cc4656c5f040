"""Throughput with TWO batches in flight (two model replicas, two streams) against one: do the latency-bound slice-chain
launches of one step hide under the other step's convolutions?"""
import copy, sys, time, torch
sys.path.insert(0, "/root/repo")
sys.argv = sys.argv[:1]
import vampic
from bench import build_model
dev = torch.device("cuda")
netA, sd = build_model(dev)
netB = copy.deepcopy(netA)
xs = [vampic.synth.synth_image(32, 256, 256, 100 + i).to(dev) for i in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            if two:
                with torch.cuda.stream(sA if i % 2 == 0 else sB):
                    (netA if i % 2 == 0 else netB).forward_single_quality(xs[i % 2], 2.5, clone=False)
            else:
                netA.forward_single_quality(xs[0], 2.5, clone=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    run(6, two)
    print("two in flight" if two else "one in flight", f"{run(20, two):.3f} ms per step")
