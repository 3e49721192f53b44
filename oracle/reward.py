"""Sparse per-task reward (oracle side).  TEST INFRASTRUCTURE ONLY.

The reference obtains rewards from ``env.unwrapped.compute_reward(achieved_goal=,
goal=, task_descr=, info=)`` (baselines/her/experiment/config.py:158-159), which lives in
the un-vendored ``gym_flowers`` package -> "parity unpinned" upstream.  The build fixes
the reward to the form described in the CURIOUS paper and exercised by
baselines/her/experiment/test_env.py:14-26:

    task  = index of the 1 in task_descr
    d     = || ag_2[tasks_ag_id[task][:len(tasks_g_id[task])]] - g[tasks_g_id[task]] ||_2
    r     = -1 if d > eps else 0            (eps = 0.05), shape [n, 1]

Arithmetic contract (what the HIP kernel reproduces bit for bit): the inputs are
float32-representable values promoted to float64; the squared differences are summed
sequentially in index order in float64 without fused multiply-add; sqrt is correctly
rounded; the comparison is done in float64.
"""
import numpy as np

DEFAULT_EPS = 0.05


def make_reward_fun(tasks_ag_id, tasks_g_id, eps=DEFAULT_EPS, flat=False):
    """Returns reward_fun(ag_2, g, task_descr, info) -> float32 [n, 1]."""
    nb_tasks = len(tasks_g_id)
    ag_ids = [list(tasks_ag_id[j][:len(tasks_g_id[j])]) for j in range(nb_tasks)]
    g_ids = [list(tasks_g_id[j]) for j in range(nb_tasks)]

    def reward_fun(ag_2, g, task_descr=None, info=None):
        ag_2 = np.asarray(ag_2, dtype=np.float64)
        g = np.asarray(g, dtype=np.float64)
        single = ag_2.ndim == 1
        if single:
            ag_2, g = ag_2[None], g[None]
            if task_descr is not None:
                task_descr = np.asarray(task_descr)[None]
        n = ag_2.shape[0]
        r = np.zeros([n, 1], dtype=np.float32)
        if task_descr is None:
            # flat structure: one reward over every goal slot (her.py:57-58 passes None)
            a_idx = sum(ag_ids, [])
            g_idx = sum(g_ids, [])
            d2 = np.zeros(n)
            for a, b in zip(a_idx, g_idx):
                diff = ag_2[:, a] - g[:, b]
                d2 = d2 + diff * diff
            r[:, 0] = -(np.sqrt(d2) > eps).astype(np.float32)
        else:
            tasks = np.argmax(np.asarray(task_descr), axis=1)
            for j in range(nb_tasks):
                rows = np.where(tasks == j)[0]
                if rows.size == 0:
                    continue
                d2 = np.zeros(rows.size)
                for a, b in zip(ag_ids[j], g_ids[j]):
                    diff = ag_2[rows, a] - g[rows, b]
                    d2 = d2 + diff * diff
                r[rows, 0] = -(np.sqrt(d2) > eps).astype(np.float32)
        return r[0] if single else r

    reward_fun.spec = dict(kind='sparse_l2', eps=float(eps))
    return reward_fun
