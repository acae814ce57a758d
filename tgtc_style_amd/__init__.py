"""Importable alias for the `tgtc-style_amd/` package directory.

The package directory carries a hyphen (repo naming convention), which is not a valid Python
identifier; this stub makes `import tgtc_style_amd.<module>` resolve to
`tgtc-style_amd/<module>.py` without symlinks (which may not survive a snapshot copy).
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tgtc-style_amd")
if not _os.path.isdir(_real):
    raise ImportError("tgtc-style_amd/ directory not found next to tgtc_style_amd/")
__path__.insert(0, _real)

__version__ = "0.1.0"
