"""Reference-named entry point (transformer.py): the style Transformer on the HIP library.
Unlike the reference (transformer.py:11) importing this module does not touch CUDA_VISIBLE_DEVICES."""
from .style2d import Transformer  # noqa: F401
