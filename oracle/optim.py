"""Flat-vector Adam and Polyak averaging (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates baselines/common/mpi_adam.py:21-35 (MpiAdam.update, after the Allreduce) and
baselines/her/ddpg.py:459-462 (target-net init / Polyak update).

NumPy promotion note.  ``a`` in mpi_adam.py:30 is an np.float64 *scalar*.  With the
NumPy 1.x the reference was written for, value-based casting keeps
``(-a) * self.m / (np.sqrt(self.v) + self.epsilon)`` in float32.  Under NumPy >= 2
(NEP 50) the same line promotes to float64 and the result is rounded to float32 once,
when it is written back into the float32 variable.  ``nep50=False`` (default) restates
the historical float32 arithmetic -- this is what the HIP kernel reproduces bit for bit;
``nep50=True`` restates what the imported reference computes under this container's
NumPy 2.2 and is what tests/golden/adam.npz was captured with.
"""
import numpy as np


def adam_alpha(stepsize, t, beta1=0.9, beta2=0.999):
    # mpi_adam.py:30 (float64 scalar math)
    return stepsize * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)


def adam_update(theta, m, v, t, globalg, stepsize, beta1=0.9, beta2=0.999, epsilon=1e-08,
                nep50=False):
    """One MpiAdam.update after the gradient all-reduce.  Returns (theta, m, v, t)."""
    globalg = globalg.astype(np.float32)
    t = t + 1                                                     # mpi_adam.py:29
    a = adam_alpha(stepsize, t, beta1, beta2)                      # mpi_adam.py:30
    f = np.float32
    m = f(beta1) * m + f(1 - beta1) * globalg                      # mpi_adam.py:31
    v = f(beta2) * v + f(1 - beta2) * (globalg * globalg)          # mpi_adam.py:32
    if nep50:
        step = np.float64(-a) * m.astype(np.float64) / (np.sqrt(v) + f(epsilon)).astype(np.float64)
        theta = (theta.astype(np.float64) + step).astype(np.float32)
    else:
        step = f(-a) * m / (np.sqrt(v) + f(epsilon))               # mpi_adam.py:33
        theta = theta + step                                       # mpi_adam.py:34
    return theta.astype(np.float32), m.astype(np.float32), v.astype(np.float32), t


def polyak_update(target, main, polyak):
    # ddpg.py:461-462: target <- polyak*target + (1-polyak)*main (float32 TF ops)
    f = np.float32
    return (f(polyak) * target + f(1. - polyak) * main).astype(np.float32)
