"""Stand-in (pytorch_msssim 0.2.1 is absent offline): lets the reference's utility module import; raises on use."""


def ms_ssim(*a, **k):
    raise NotImplementedError("pytorch_msssim stand-in: MS-SSIM is not pinned by the fixtures")
