"""Helpers with the reference's names (reference: commons.py)."""
from .functional import sequence_mask  # noqa: F401  (commons.py:128-132)


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """commons.py:14-15"""
    return int((kernel_size * dilation - dilation) / 2)
