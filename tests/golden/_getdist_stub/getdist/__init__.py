"""Import-time stand-in for the third-party `getdist` package (absent in this image).

Used ONLY by tests/golden/make_golden.py, in the build container, so that the Python
reference under /root/reference can be imported to generate golden vectors
(cobaya/collection.py imports 4 getdist names at import time). It implements nothing.
"""
from . import chains  # noqa: F401


class MCSamples:  # pragma: no cover - placeholder type only
    pass
