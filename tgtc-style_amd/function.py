"""Reference-named entry point (function.py:4-12)."""
from .style2d import calc_mean_std  # noqa: F401
