"""Stand-in for the two `control==0.9.1` calls on the pH path (tf2ss, c2d).

Without slycot, control 0.9.1 routes SISO tf2ss to scipy.signal.tf2ss and c2d
('zoh') to scipy.signal.cont2discrete; this shim calls scipy directly.
Container-only test infrastructure (see tests/golden/make_golden.py).
"""
import numpy as np
import scipy.signal

from . import matlab  # noqa: F401


class StateSpace(object):
    def __init__(self, A, B, C, D, dt=0):
        self.A = np.atleast_2d(np.asarray(A, dtype=float))
        self.B = np.atleast_2d(np.asarray(B, dtype=float))
        self.C = np.atleast_2d(np.asarray(C, dtype=float))
        self.D = np.atleast_2d(np.asarray(D, dtype=float))
        self.dt = dt


def ss(A, B, C, D, dt=0):
    return StateSpace(A, B, C, D, dt)


def tf2ss(num, den):
    A, B, C, D = scipy.signal.tf2ss(np.squeeze(num), np.squeeze(den))
    return StateSpace(A, B, C, D)


def c2d(sysc, Ts, method="zoh"):
    Ad, Bd, Cd, Dd, _ = scipy.signal.cont2discrete((sysc.A, sysc.B, sysc.C, sysc.D), Ts, method=method)
    return StateSpace(Ad, Bd, Cd, Dd, Ts)
