"""Import shim: the package sources live in ``rfw-rs_amd/`` (a name Python cannot import),
so this package only extends its search path to that directory and re-exports the API."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rfw-rs_amd")
__path__.insert(0, _real)

from .api import *  # noqa: E402,F401,F403
