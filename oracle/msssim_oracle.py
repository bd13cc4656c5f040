"""CPU restatement of MS-SSIM as the reference calls it (utility/functions.py:176-177:
``pytorch_msssim.ms_ssim(a, b, data_range=1.)``).  TEST INFRASTRUCTURE.

pytorch_msssim 0.2.1 (environment.yml) is absent offline, so this restates its published algorithm
(Wang et al. 2003 as implemented there) — **parity unpinned**: no vectors of the package are available here.
  * window: 11-tap Gaussian, sigma 1.5, normalised; applied separably, "valid" (no padding), per channel;
  * per level: mu, sigma^2, sigma_xy from the filtered x, y, x^2, y^2, xy; C1 = (0.01 L)^2, C2 = (0.03 L)^2;
    cs = (2 sigma_xy + C2) / (sigma_x^2 + sigma_y^2 + C2), ssim = (2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1) * cs,
    both averaged over the valid positions per (image, channel);
  * 5 levels, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333); between levels avg_pool2d(kernel 2, padding = size % 2,
    zeros counted); relu on cs / ssim; result = prod_l cs_l^w_l * ssim_L^w_L per (image, channel), then the mean.
"""
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gaussian_window(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    C = x.shape[1]
    w = win.reshape(1, 1, 1, -1).repeat(C, 1, 1, 1)
    x = F.conv2d(x, w.transpose(2, 3), groups=C)          # along H
    return F.conv2d(x, w, groups=C)                        # along W


def _ssim_level(x, y, win, data_range):
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    s1 = _filter(x * x, win) - mu1 * mu1
    s2 = _filter(y * y, win) - mu2 * mu2
    s12 = _filter(x * y, win) - mu1 * mu2
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = (2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> float:
    assert x.shape == y.shape and x.dim() == 4
    assert min(x.shape[2:]) > (11 - 1) * 2 ** 4, "image too small for 5 scales with an 11-tap window"
    win = gaussian_window()
    x, y = x.double(), y.double()
    win = win.double()
    mcs = []
    for lvl in range(5):
        ssim_pc, cs = _ssim_level(x, y, win, data_range)
        if lvl < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_pc)], dim=0)            # [5, B, C]
    w = torch.tensor(WEIGHTS, dtype=vals.dtype).reshape(-1, 1, 1)
    return float(torch.prod(vals ** w, dim=0).mean())
