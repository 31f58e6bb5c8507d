"""Agent side of the set-point-control stack: the interface of the reference's `elegantrl` package for the
residual-PPO / TD3 path, re-built vectorised-first on PyTorch-ROCm with the rollout,
GAE scan and value pass on the HIP kernels of libpime_hip.so."""
from . import logger  # noqa: F401  (the reference imports `from elegantrl import logger`; its own file is missing)
