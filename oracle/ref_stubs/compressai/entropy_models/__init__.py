class EntropyBottleneck:  # placeholder: only imported by the out-of-scope WACNN
    pass


class GaussianConditional:
    pass
