"""Import alias for the package directory the build contract names
`pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd/` (not a valid Python identifier).

`import pime_amd` resolves sub-modules from that directory through ``__path__``.
"""
import os as _os

_IMPL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd")
if not _os.path.isdir(_IMPL):  # pragma: no cover
    raise ImportError(f"pime_amd: implementation directory missing: {_IMPL}")
__path__.insert(0, _IMPL)

from ._pkg import *  # noqa: E402,F401,F403
from ._pkg import __version__  # noqa: E402,F401
