"""Competence / learning-progress queues and the epsilon-proportional task probabilities (host logic).

Mirrors CompetenceQueue baselines/her/queues.py:7-36 and the task-probability update of
RolloutWorker.generate_rollouts rollout.py:374-393.
"""
from collections import deque

import numpy as np


class CompetenceQueue:
    """Sliding window over the last 2*window exploit outcomes of one task.
    C = mean of the newest half, CP = |sum(newest half) - sum(older half)| / (2*w) (queues.py:16-21)."""

    def __init__(self, window=100):
        self.window = window
        self.successes = deque(maxlen=2 * window)
        self.CP = 0.
        self.C = 0.

    def update(self, success_list):
        self.successes.extend(success_list)
        n = self.size
        if n > 2:
            w = min(n // 2, self.window)
            recent = np.fromiter(self.successes, dtype=np.float64, count=n)
            newest = np.sum(recent[n - w:].tolist())
            older = np.sum(recent[n - 2 * w:n - w].tolist())
            self.CP = np.abs(newest - older) / (2 * w)
            self.C = newest / w

    @property
    def size(self):
        return len(self.successes)

    @property
    def full(self):
        return self.size == self.successes.maxlen

    def clear_queue(self):
        self.successes = deque(maxlen=2 * self.window)
        self.CP = 0
        self.C = 0.


def task_probabilities(CP, nb_tasks, epsilon=0.4):
    """p = eps/N + (1-eps)*CP/sum(CP), uniform when sum(CP) == 0, then the sum-to-one fix-up (rollout.py:383-393)."""
    CP = np.asarray(CP, dtype=np.float64)
    total = CP.sum()
    if total == 0:
        p = (1 / nb_tasks) * np.ones([nb_tasks])
    else:
        p = epsilon * (1 / nb_tasks) * np.ones([nb_tasks]) + (1 - epsilon) * CP / total
    s = p.sum()
    if s > 1:
        p[np.argmax(p)] -= p.sum() - 1
    elif s < 1:
        p[-1] = 1 - p[:-1].sum()
    return p
