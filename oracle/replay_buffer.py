"""Episode ring buffer (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates baselines/her/replay_buffer.py:6-109.  Storage is float64 like the
reference (replay_buffer.py:23 ``np.empty`` default dtype); slot selection is
"append until full, then uniformly random slots" (replay_buffer.py:90-109).
"""
import numpy as np


class ReplayBuffer:
    def __init__(self, buffer_shapes, size_in_transitions, T, sample_transitions, rng=None):
        self.buffer_shapes = buffer_shapes
        self.size = size_in_transitions // T                      # replay_buffer.py:18
        self.T = T
        self.sample_transitions = sample_transitions
        self.buffers = {k: np.empty([self.size, *s]) for k, s in buffer_shapes.items()}  # :23-24
        self.current_size = 0
        self.n_transitions_stored = 0
        self.rng = rng

    @property
    def full(self):
        return self.current_size == self.size

    def sample(self, batch_size, task_to_replay=None, cp_proba=None):
        assert self.current_size > 0                              # replay_buffer.py:43
        buffers = {k: v[:self.current_size] for k, v in self.buffers.items()}
        buffers['o_2'] = buffers['o'][:, 1:, :]                   # replay_buffer.py:47
        buffers['ag_2'] = buffers['ag'][:, 1:, :]                 # replay_buffer.py:48
        transitions = self.sample_transitions(buffers, batch_size, task_to_replay=task_to_replay,
                                              cp_proba=cp_proba)
        for key in (['r', 'o_2', 'ag_2'] + list(self.buffers.keys())):
            assert key in transitions, "key %s missing from transitions" % key
        return transitions

    def store_episode(self, episode_batch):
        batch_sizes = [len(episode_batch[k]) for k in episode_batch.keys()]
        assert np.all(np.array(batch_sizes) == batch_sizes[0])
        batch_size = batch_sizes[0]
        idxs = self._get_storage_idx(batch_size)
        for key in self.buffers.keys():
            self.buffers[key][idxs] = episode_batch[key]          # replay_buffer.py:69-70
        self.n_transitions_stored += batch_size * self.T
        return idxs

    def get_current_episode_size(self):
        return self.current_size

    def get_current_size(self):
        return self.current_size * self.T

    def get_transitions_stored(self):
        return self.n_transitions_stored

    def clear_buffer(self):
        self.current_size = 0

    def _get_storage_idx(self, inc=None):
        r = np.random if self.rng is None else self.rng
        inc = inc or 1
        assert inc <= self.size, "Batch committed to replay is too large!"
        if self.current_size + inc <= self.size:                  # replay_buffer.py:94-95
            idx = np.arange(self.current_size, self.current_size + inc)
        elif self.current_size < self.size:                       # replay_buffer.py:96-100
            overflow = inc - (self.size - self.current_size)
            idx_a = np.arange(self.current_size, self.size)
            idx_b = r.randint(0, self.current_size, overflow)
            idx = np.concatenate([idx_a, idx_b])
        else:                                                     # replay_buffer.py:101-102
            idx = r.randint(0, self.size, inc)
        self.current_size = min(self.size, self.current_size + inc)
        if inc == 1:
            idx = idx[0]
        return idx
