def pmf_to_quantized_cdf(pmf, precision):
    raise RuntimeError("compressai._CXX is not available offline (stub)")
