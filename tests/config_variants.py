"""Constructor-flag variants away from the README configuration (reference models/__init__.py:11-55,
models/pic.py:27-164, layers/rem.py:69-128): the table oracle/gen_golden.py section 9 ran the REFERENCE on
(tests/golden/config_variants.*).  TEST INFRASTRUCTURE."""
import argparse

from conftest import README_ARGS

CONFIG_VARIANTS = {
    "single_encoder": dict(multiple_encoder=False),
    "single_decoder": dict(multiple_decoder=False),
    "single_hyperprior": dict(multiple_hyperprior=False),
    "all_single": dict(multiple_encoder=False, multiple_decoder=False, multiple_hyperprior=False),
    "sp0": dict(support_progressive_slices=0),
    "sp2": dict(support_progressive_slices=2),
    "sp8": dict(support_progressive_slices=8),
    "no_delta_no_mu_rep": dict(delta_encode=False, total_mu_rep=False),
    "not_all_scalable": dict(all_scalable=False),
    "rem_big": dict(model="rem", dimension="big"),
    "rem_no_mu_std": dict(model="rem", mu_std=False),
    "rem_not_all_scalable": dict(model="rem", all_scalable=False, support_progressive_slices=3),
}


def variant_args(name: str) -> argparse.Namespace:
    base = dict(model="pic", check_levels=[0.75], mu_std=True, dimension="middle", **README_ARGS)
    base.update(CONFIG_VARIANTS[name])
    return argparse.Namespace(**base)


def oracle_kwargs(a: argparse.Namespace) -> dict:
    kw = dict(prog_support=a.support_progressive_slices, multiple_encoder=a.multiple_encoder,
              multiple_decoder=a.multiple_decoder, multiple_hyperprior=a.multiple_hyperprior, delta_encode=a.delta_encode,
              total_mu_rep=a.total_mu_rep, all_scalable=a.all_scalable)
    if a.model == "rem":
        kw.update(check_levels=a.check_levels, mu_std=a.mu_std)
    return kw
