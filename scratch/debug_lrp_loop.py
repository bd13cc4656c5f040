import sys, copy, argparse, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import vampic, vampic.synth as synth
from vampic import finetune as FT
from conftest import README_ARGS
net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=0)); net = net.cuda()
x = synth.synth_image(2, 64, 64, seed=8).cuda()
for trial in range(5):
    m = copy.deepcopy(net); m.use_graph = True
    params = FT.refine_gs_setup(m, lrp=True)
    crit = FT.DistortionLoss(device="cuda")
    opt = torch.optim.Adam(params, lr=1e-4)
    m.train()
    ls = []
    for it in range(14):
        opt.zero_grad()
        out = m.forward_single_quality(x, quality=2.5, training=True)
        c = crit(out, x)
        c["loss"].backward()
        for n, p in m.named_parameters():
            if p.grad is not None:
                mx = float(p.grad.abs().max())
                if not (mx < 1e2):
                    idx = int(p.grad.abs().flatten().argmax())
                    print(f"   trial {trial} step {it}: {n} max|g| {mx:.3e} at flat index {idx} of {p.numel()}", flush=True)
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        ls.append(round(float(c["loss"].detach()), 3))
    print("trial", trial, ls, flush=True)
