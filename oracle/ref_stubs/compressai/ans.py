class _Unavailable:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        raise RuntimeError("compressai.ans is not available offline (stub)")


class RansEncoder(_Unavailable):
    pass


class RansDecoder(_Unavailable):
    pass


class BufferedRansEncoder(_Unavailable):
    pass
