"""Competence / learning-progress queue (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates baselines/her/queues.py:7-36 and the epsilon-proportional task probabilities
of baselines/her/rollout.py:374-393.
"""
from collections import deque
import numpy as np


class CompetenceQueue:
    def __init__(self, window=100):
        self.window = window
        self.successes = deque(maxlen=2 * window)
        self.CP = 0.
        self.C = 0.

    def update(self, success_list):
        self.successes.extend(success_list)                       # queues.py:13-14
        n = len(self.successes)
        if n > 2:                                                 # queues.py:16
            w = min(n // 2, self.window)
            s = list(self.successes)
            q1 = s[n - w:]
            q2 = s[n - 2 * w:n - w]
            self.CP = np.abs(np.sum(q1) - np.sum(q2)) / (2 * w)   # queues.py:20
            self.C = np.sum(q1) / w                               # queues.py:21

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
    """rollout.py:383-393 (epsilon hard-coded to 0.4 there)."""
    CP = np.asarray(CP, dtype=np.float64)
    if CP.sum() == 0:
        p = (1 / nb_tasks) * np.ones([nb_tasks])
    else:
        p = epsilon * (1 / nb_tasks) * np.ones([nb_tasks]) + (1 - epsilon) * CP / CP.sum()
    if p.sum() > 1:
        p[np.argmax(p)] -= p.sum() - 1
    elif p.sum() < 1:
        p[-1] = 1 - p[:-1].sum()
    return p
