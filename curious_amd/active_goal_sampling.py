"""SAGG-RIAC goal sampling (goal_selection='active').  Mirrors baselines/her/active_goal_sampling.py:8-194.

Host-side logic only (a few hundred goals per task): the goal space of one task is split recursively into axis-aligned
regions; goals are drawn from regions in proportion to exp(temperature * |competence progress|).  The class keeps the
reference's interface (`update(goals, binary_competence, continuous_competence=None) -> (new_split, all_order)`,
`sample_goal()`, `nb_regions`, `get_regions`, `.probas`, `.interest`, `.region_bounds`, `.regions`) and its arithmetic
and order of random draws, including the parts that look accidental (noted inline), so that a seeded run makes the same
decisions.  Differences: no dependency on gym (a minimal `Box` with the four members the algorithm uses; `sample()`
draws from the NumPy global stream), no debug printing.
"""
from collections import deque

import numpy as np


class Box:
    """The subset of gym.spaces.Box used by SAGG-RIAC: private float32 copies of the bounds, closed-interval
    `contains`, uniform `sample`."""

    def __init__(self, low, high, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.array(low, dtype=self.dtype)
        self.high = np.array(high, dtype=self.dtype)
        self.shape = self.low.shape

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def sample(self):
        return np.random.uniform(low=self.low, high=self.high).astype(self.dtype)


def _progress(outcomes, window_cp, continuous):
    """|competence progress| of a list of outcomes (active_goal_sampling.py:92-99,163-170)."""
    c = np.array(outcomes)
    if continuous:
        w = min(len(c), window_cp)
        return np.abs(c[-w].mean())                     # (sic) a single element: reference indexes, not slices
    w = min(len(c) // 2, window_cp)
    return np.abs(c[-2 * w:-w].mean() - c[-w:].mean())


class SAGG_RIAC:
    def __init__(self, min, max, continuous_competence=False):
        assert len(min) == len(max)
        self.maxlen = 200
        self.regions = [[deque(maxlen=self.maxlen + 1), deque(maxlen=self.maxlen + 1)]]   # [outcomes, goals]
        self.region_bounds = [Box(min, max, dtype=np.float32)]
        self.interest = [0.]
        self.probas = [1.]
        self.nb_dims = len(min)
        self.window_cp = 100
        self.temperature = 20
        self.nb_split_attempts = 50
        self.continuous_competence = continuous_competence
        self.max_difference = 0.3
        self.init_size = max - min

    # ------------------------------------------------------------------ one candidate split of a full region
    def _draw_split(self, reg):
        """Redraw (dimension, threshold) until both halves hold at least maxlen / 4 goals (:61-87)."""
        parent = self.region_bounds[reg]
        outcomes, goals = self.regions[reg]
        while True:
            dim = np.random.choice(range(self.nb_dims))
            threshold = parent.sample()[dim]
            lower, upper = Box(parent.low, parent.high), Box(parent.low, parent.high)
            lower.high[dim] = threshold
            upper.low[dim] = threshold
            valid = not (np.any(lower.high - lower.low < self.init_size / 5) or
                         np.any(upper.high - upper.low < self.init_size / 5))
            halves = [[deque(), deque()], [deque(), deque()]]
            for outcome, goal in zip(outcomes, goals):
                side = 0 if lower.contains(goal) else 1
                halves[side][0].append(outcome)
                halves[side][1].append(goal)
            if len(halves[0][0]) >= self.maxlen / 4 and len(halves[1][0]) >= self.maxlen / 4:
                return [lower, upper], halves, valid

    def update(self, goals, binary_competence, continuous_competence=None):
        if len(goals) == 0:
            return False, None
        new_split, all_order = False, None
        outcomes = continuous_competence if self.continuous_competence else binary_competence
        # file every goal under the first region that contains it (:31-45); a goal outside every region is an error
        for goal, outcome in zip(goals, outcomes):
            home = next((j for j, rb in enumerate(self.region_bounds) if rb.contains(goal)), None)
            if home is None:
                raise TypeError('goal %r lies in no region' % (goal,))
            self.regions[home][0].append(outcome)
            self.regions[home][1].append(goal)

        # regions that overflowed try nb_split_attempts random splits and keep the best-scoring admissible one (:48-128)
        split_regions, kept_bounds, kept_halves = [], [], []
        order = None
        for reg in range(self.nb_regions):
            if len(self.regions[reg][0]) <= self.maxlen:
                continue
            best_score, best_diff, best_bounds, best_halves, found = 0, 0, None, None, False
            for _ in range(self.nb_split_attempts):
                bounds, halves, valid = self._draw_split(reg)
                interest = [_progress(h[0], self.window_cp, self.continuous_competence) for h in halves]
                diff = np.abs(interest[0] - interest[1])
                score = 2 * 2 * diff                   # (sic) reference multiplies len([outcomes, goals]) of both halves
                if score >= best_score and diff >= self.max_difference / 2 and valid:
                    best_score, best_diff, best_bounds, best_halves, found = score, diff, bounds, halves, True
                    order = [1, -1] if interest[0] >= interest[1] else [-1, 1]
            if found:
                split_regions.append(reg)
                if best_diff > self.max_difference:
                    self.max_difference = best_diff
            else:                                       # forget the oldest quarter (:120-122)
                for k in range(2):
                    kept = np.array(self.regions[reg][k])[-int(3 * len(self.regions[reg][k]) / 4):]
                    self.regions[reg][k] = deque(kept, maxlen=self.maxlen + 1)
            kept_bounds.append(best_bounds)             # (sic) one entry per OVERFLOWED region ...
            kept_halves.append(best_halves)

        for i, reg in enumerate(split_regions):         # ... indexed per SPLIT region below, as in the reference (:131-152)
            all_order = [0] * self.nb_regions
            all_order.pop(reg)
            all_order.insert(reg, order[0])
            all_order.insert(reg, order[1])
            new_split = True
            for seq, pair in ((self.region_bounds, kept_bounds[i]), (self.regions, kept_halves[i]),
                              (self.interest, (0, 0)), (self.probas, (0, 0))):
                seq.pop(reg)
                seq.insert(reg, pair[0])
                seq.insert(reg, pair[1])

        for i in range(self.nb_regions):                # :156-171
            cp = _progress(self.regions[i][0], self.window_cp, self.continuous_competence) \
                if len(self.regions[i][0]) > 10 else 0
            self.interest[i] = np.abs(cp)
        weights = np.exp(self.temperature * np.array(self.interest))
        self.probas = (weights / weights.sum()).tolist()
        assert len(self.probas) == len(self.regions)
        return new_split, all_order

    def sample_goal(self):
        if np.random.rand() < 0.2:                      # :180-183
            region_id = np.random.choice(range(self.nb_regions))
        else:
            region_id = np.random.choice(range(self.nb_regions), p=np.array(self.probas))
        return self.region_bounds[region_id].sample()

    @property
    def nb_regions(self):
        return len(self.regions)

    @property
    def get_regions(self):
        return self.region_bounds
