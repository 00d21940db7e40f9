"""Importable alias of the `unet-watermark_amd/` package directory (a hyphen is not a valid Python
identifier, so the real package directory is put on this package's search path)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "unet-watermark_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
