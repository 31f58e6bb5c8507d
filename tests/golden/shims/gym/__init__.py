"""Minimal stand-in for the `gym==0.18.0` API surface the reference imports.

Container-only test infrastructure: it lets `tests/golden/make_golden.py` import
the *unmodified* reference from /root/reference to emit golden vectors.  Not
shipped, not used by the product path.  Written from the gym 0.18 public
behaviour (Env / Wrapper attribute forwarding / Box bound casting / TimeLimit /
seeding.np_random); gym itself is not installable here (no network).
"""
from . import error, logger, spaces, utils  # noqa: F401
from .core import Env, Wrapper  # noqa: F401
from .envs.registration import make, register, registry  # noqa: F401
from . import envs, wrappers  # noqa: F401

__version__ = "0.18.0-shim"
