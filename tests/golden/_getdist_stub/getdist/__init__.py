"""Import-time stand-in for the third-party `getdist` package (absent in this image).

Used ONLY by tests/golden/make_golden.py, in the build container, so that the Python
reference under /root/reference can be imported to generate golden vectors
(cobaya/collection.py imports 4 getdist names at import time). It implements nothing.
"""
from . import chains  # noqa: F401


class MCSamples:  # pragma: no cover - placeholder type only
    pass


class IniFile:  # pragma: no cover - placeholder type only
    """cobaya/likelihoods/base_classes/cmblikes.py:15 imports the name at import time (reached
    when G13 imports planck_pliklite); never instantiated: G13 hands `init_params` its own
    key/value object (make_golden._Ini)."""

    def __init__(self, *a, **k):
        raise RuntimeError("getdist is not installed: this is an import-time placeholder")


class ParamNames(IniFile):  # pragma: no cover - placeholder type only
    pass
