"""Running mean/std normaliser (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates baselines/her/normalizer.py:10-118.  ``comm_size``/``allreduce`` stand in for
MPI.COMM_WORLD (normalizer.py:84-94): sums are *averaged* over ranks, and the running
count starts at 1 while the running sums start at 0 (normalizer.py:31-39).
"""
import numpy as np


class Normalizer:
    def __init__(self, size, eps=1e-2, default_clip_range=np.inf, allreduce=None, comm_size=1):
        self.size = size
        self.eps = eps
        self.default_clip_range = default_clip_range
        self.local_sum = np.zeros(size, np.float32)                # normalizer.py:27-29
        self.local_sumsq = np.zeros(size, np.float32)
        self.local_count = np.zeros(1, np.float32)
        self.sum = np.zeros(size, np.float32)                      # sum_tf   (zeros, :31-33)
        self.sumsq = np.zeros(size, np.float32)                    # sumsq_tf (zeros, :34-36)
        self.count = np.ones(1, np.float32)                        # count_tf (ones,  :37-39)
        self.mean = np.zeros(size, np.float32)                     # :40-42
        self.std = np.ones(size, np.float32)                       # :43-45
        self._allreduce = allreduce
        self._comm_size = comm_size

    def update(self, v):
        v = v.reshape(-1, self.size)                               # normalizer.py:65
        self.local_sum += v.sum(axis=0)                            # :68
        self.local_sumsq += (np.square(v)).sum(axis=0)             # :69
        self.local_count[0] += v.shape[0]                          # :70

    def _mpi_average(self, x):
        buf = x.copy() if self._allreduce is None else self._allreduce(x)   # :85-86
        buf /= self._comm_size                                     # :87
        return buf

    def recompute_stats(self):
        local_count = self.local_count.copy()
        local_sum = self.local_sum.copy()
        local_sumsq = self.local_sumsq.copy()
        self.local_count[...] = 0
        self.local_sum[...] = 0
        self.local_sumsq[...] = 0
        synced_sum = self._mpi_average(local_sum)                  # :90-94
        synced_sumsq = self._mpi_average(local_sumsq)
        synced_count = self._mpi_average(local_count)
        self.count = self.count + synced_count                     # update_op :50-54 (f32)
        self.sum = self.sum + synced_sum
        self.sumsq = self.sumsq + synced_sumsq
        f = np.float32
        mean = self.sum / self.count                               # recompute_op :55-61 (f32)
        self.mean = mean.astype(f)
        var = self.sumsq / self.count - np.square(self.sum / self.count)
        self.std = np.sqrt(np.maximum(np.square(f(self.eps)), var)).astype(f)

    def normalize(self, v, clip_range=None):
        if clip_range is None:
            clip_range = self.default_clip_range
        v = np.asarray(v, dtype=np.float32)
        return np.clip((v - self.mean) / self.std, -clip_range, clip_range).astype(np.float32)  # :72-77

    def denormalize(self, v):
        return self.mean + np.asarray(v, dtype=np.float32) * self.std
