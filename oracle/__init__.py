"""CPU oracle for the set-point-control hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as
the checker.  The product package never imports it.  See pime_oracle.c for the restatement and its
reference citations.  (The MT19937 stream emulation of the seed-for-seed mode is host logic of the product:
pime_amd/vec_env.py:Mt19937Draws and pime_amd/gym_compat.py; it is pinned by the seeded golden rollouts.)
"""
from .binding import (  # noqa: F401
    OraclePH, OracleWT, build, critic_forward, explore_noise, gae, lib, modular_actor_mean, philox4x32_10,
    philox_uniform_pair, ph_table, ph_zoh, plain_actor_mean, residual_action, set_threads,
)
