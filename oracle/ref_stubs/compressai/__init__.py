"""Minimal stand-in for compressai 1.2.4 (see ../README.md)."""
_entropy_coder = "ans"


def available_entropy_coders():
    return ["ans"]


def get_entropy_coder():
    return _entropy_coder
