"""Where does the NaN of progressive.decode(q_ind=0) at 256x256 come from?  (debug script, GPU box)"""
import sys, argparse, torch
sys.path[:0] = ["/root/repo", "/root/repo/oracle", "/root/repo/tests"]
import vampic
from vampic import progressive as P
from conftest import README_ARGS
args = argparse.Namespace(model="rem", check_levels=[0.75], mu_std=True, dimension="middle", **README_ARGS)
net = vampic.get_model(args, "cpu").eval()
sd = vampic.synth.synth_state_dict(net.state_dict(), seed=0)
torch.nn.Module.load_state_dict(net, sd)
net = net.cuda()
net.update()
def st(name, t):
    t = t.float()
    print(f"{name:28s} shape {tuple(t.shape)} nan {int(torch.isnan(t).sum())} inf {int(torch.isinf(t).sum())} absmax {float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else -1:.4g}", flush=True)
for size in (64, 128, 256):
    print("=== size", size)
    x = vampic.synth.synth_image(1, size, size, seed=0).cuda()
    with torch.no_grad():
        base = net.compress(x, quality=0)
        st("compress y_hat_base", base["y_hat_base"])
        z_hat, means_h, scales_h, ysh = P._decode_hyper(net, base["strings"][1], base["shape"], 0)
        st("z_hat", z_hat); st("means_h", means_h); st("scales_h", scales_h)
        d, gc = 320, net.gaussian_conditional
        y_hat = []
        for i in range(10):
            sup = y_hat[:min(5, i)]
            m_sup = torch.cat([means_h[:, :d]] + sup, dim=1)
            s_sup = torch.cat([scales_h[:, :d]] + sup, dim=1)
            mu, sc = net.cc_mean_transforms[i](m_sup), net.cc_scale_transforms[i](s_sup)
            st(f"slice {i} mu", mu); st(f"slice {i} sc", sc)
            idx = gc.build_indexes(sc)
            rv = gc.decompress(base["strings"][0][i], idx).reshape(mu.shape)
            st(f"slice {i} rv", rv)
            yh = gc.dequantize(rv, mu)
            l = net.lrp_transforms[i](torch.cat([m_sup, yh], dim=1))
            st(f"slice {i} lrp", l)
            yh = yh + 0.5 * torch.tanh(l)
            y_hat.append(yh)
        yb = torch.cat(y_hat, 1)
        st("y_hat_base (decoded)", yb)
        st("diff vs compress", yb - base["y_hat_base"])
        xh = net.g_s[0](yb)
        st("g_s[0](y_hat)", xh)
        xh2 = net.g_s[0](base["y_hat_base"])
        st("g_s[0](compress y_hat)", xh2)
        fw = net.forward_single_quality(x, 0)
        st("fsq x_hat", fw["x_hat"])
