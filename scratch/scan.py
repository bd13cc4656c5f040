import sys, torch, argparse
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import vampic, vampic_oracle as O
args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True, dim_chunk=32, division_dimension=[320,640], mask_policy="point-based-std", support_progressive_slices=5, delta_encode=True, total_mu_rep=True, all_scalable=True)
net = vampic.get_model(args, "cpu").eval()
sd = vampic.synth.synth_state_dict(net.state_dict(), 0); net.load_state_dict(sd); net = net.cuda()
for (B,H,W) in [(1,64,64),(1,64,128),(1,128,128)]:
  for seed in range(4):
    x = vampic.synth.synth_image(B,H,W,seed)
    for q in (0, 0.5, 2.5, 10):
        ref = O.forward_single_quality(sd, x, q)
        with torch.no_grad(): out = net.forward_single_quality(x.cuda(), q)
        fl = int((torch.round(out["y_hat"].cpu()-ref["y_hat"]).abs()>=1).sum())
        mf = int((out["mask"].cpu()!=ref["mask"]).sum()) if q else 0
        print((B,H,W), "seed", seed, "q", q, "symflips", fl, "maskflips", mf, "dpsnr", abs(O.psnr(x,out["x_hat"].cpu())-O.psnr(x,ref["x_hat"])), flush=True)
