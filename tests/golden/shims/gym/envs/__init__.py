from .registration import make, register, registry, spec  # noqa: F401
