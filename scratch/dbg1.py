import sys, os, torch, argparse
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import vampic, vampic_oracle as O
args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True, dim_chunk=32, division_dimension=[320,640], mask_policy="point-based-std", support_progressive_slices=5, delta_encode=True, total_mu_rep=True, all_scalable=True)
net = vampic.get_model(args, "cpu").eval()
sd = vampic.synth.synth_state_dict(net.state_dict(), 0); net.load_state_dict(sd); net = net.cuda()
net.use_graph = False
for (B,H,W) in [(1,64,128),(2,128,192),(1,128,192),(2,64,64)]:
    x = vampic.synth.synth_image(B,H,W,1)
    ref = O.forward_single_quality(sd, x, 2.5)
    with torch.no_grad(): out = net.forward_single_quality(x.cuda(), 2.5)
    plan = [p for k,p in net._plans.items() if k[:3]==(B,H,W)][0]
    y = plan.y.torch_nchw().cpu()
    def e(a,b): return f"{(a-b).abs().max().item():.3e}/{b.abs().max().item():.2e}"
    print((B,H,W), "y", e(y, ref["y"]), "zhat", e(plan.z_hat.torch_nchw().cpu(), ref["z_hat"]),
          "mu_b", e(out["mu_base"].cpu(), ref["mu_base"]), "std_b", e(out["std_base"].cpu(), ref["std_base"]),
          "ybase", e(out["y_base"].cpu(), ref["y_base"]), "mu", e(out["mu"].cpu(), ref["mu"]), "std", e(out["std"].cpu(), ref["std"]),
          "mask flips", int((out["mask"].cpu()!=ref["mask"]).sum()), "yhat", e(out["y_hat"].cpu(), ref["y_hat"]), "x", e(out["x_hat"].cpu(), ref["x_hat"]))
    # per-slice mu_base error
    mb, rb = out["mu_base"].cpu(), ref["mu_base"]
    print("   mu_base per slice", [f"{(mb[:,32*i:32*i+32]-rb[:,32*i:32*i+32]).abs().max().item():.1e}" for i in range(10)])
    # stage check of g_a[0]
    with torch.no_grad():
        t = x; pre="g_a.0."
        r0 = O.conv_k(sd, pre+"0.", x, 2); g0 = net.g_a[0][0](x.cuda()).cpu(); print("   conv0", e(g0, r0))
        r1 = O.gdn(sd, pre+"1.", r0, False); g1 = net.g_a[0][1](r0.cuda()).cpu(); print("   gdn1", e(g1, r1))
        r2 = O.conv_k(sd, pre+"2.", r1, 2); g2 = net.g_a[0][2](r1.cuda()).cpu(); print("   conv2", e(g2, r2))
        r3 = O.gdn(sd, pre+"3.", r2, False)
        r4 = O.attention_block(sd, pre+"4.", r3, 8); g4 = net.g_a[0][4](r3.cuda()).cpu(); print("   attn4", e(g4, r4))
        r5 = O.conv_k(sd, pre+"5.", r4, 2); r6 = O.gdn(sd, pre+"6.", r5, False); r7 = O.conv_k(sd, pre+"7.", r6, 2)
        g7 = net.g_a[0][7](r6.cuda()).cpu(); print("   conv7", e(g7, r7))
        r8 = O.attention_block(sd, pre+"8.", r7, 4); g8 = net.g_a[0][8](r7.cuda()).cpu(); print("   attn8", e(g8, r8))
