"""Reference-named entry point (Style_function.py:4-24)."""
from .style2d import adaptive_instance_normalization, calc_mean_std  # noqa: F401
