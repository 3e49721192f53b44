"""GPU-resident episode ring buffer.  Mirrors ReplayBuffer baselines/her/replay_buffer.py:6-109.

Same constructor, methods and attributes as the reference; differences that matter to a caller:
storage is one float32 record block in HBM (curious_amd/layout.py) instead of a dict of float64 host arrays;
`buffers` is a dict of strided GPU views onto it; `sample` returns GPU tensors.  Slot selection keeps the
reference's arithmetic and its NumPy legacy stream ("append until full, then uniformly random slots",
replay_buffer.py:90-109).
"""
import threading
from collections import OrderedDict

import numpy as np
import torch

from curious_amd import ops
from curious_amd.layout import RecordLayout, pack_episodes


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('curious_amd needs a GPU (MI355X): replay storage lives in HBM, there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


class ReplayPool:
    """One HBM allocation [n_buffers, capacity, T+1, row_stride] shared by the per-task buffers of an agent, so
    that a multi-buffer minibatch (ddpg.py:326-345) is gathered by a single kernel launch."""

    def __init__(self, layout, capacity, n_buffers):
        self.layout = layout
        self.capacity = int(capacity)
        self.n_buffers = int(n_buffers)
        self.storage = torch.empty([n_buffers, self.capacity, layout.T + 1, layout.row_stride],
                                   dtype=torch.float32, device=_device())
        self.buf_stride = self.capacity * layout.rec_floats
        self.version = 0          # bumped by every store / clear: agents sharing the pool re-draw their staged batches


class EpisodeViews(OrderedDict):
    """{key: GPU view [E, T or T+1, dim]} plus the record block the views are cut from."""

    def __init__(self, records, layout, with_next=True):
        super().__init__(layout.record_views(records))
        if with_next:
            self['o_2'] = self['o'][:, 1:, :]                       # replay_buffer.py:47
            self['ag_2'] = self['ag'][:, 1:, :]                     # replay_buffer.py:48
        self.records = records                                      # [E, T+1, row_stride]
        self.layout = layout


def as_records(episode_batch, layout):
    """Record block [E, T+1, row_stride] on the GPU for an EpisodeViews or a plain dict of arrays."""
    if isinstance(episode_batch, EpisodeViews):
        return episode_batch.records
    host = pack_episodes(layout, episode_batch)
    return torch.from_numpy(host).to(_device())


class ReplayBuffer:
    def __init__(self, buffer_shapes, size_in_transitions, T, sample_transitions, pool=None, pool_index=0):
        """buffer_shapes: {key: (T or T+1, dim)}; size_in_transitions: capacity in transitions; T: horizon;
        sample_transitions: a sampler made by curious_amd.her (replay_buffer.py:7-16)."""
        self.buffer_shapes = buffer_shapes
        self.size = size_in_transitions // T                        # replay_buffer.py:18
        self.T = T
        self.sample_transitions = sample_transitions
        self.layout = RecordLayout(buffer_shapes, T)
        if pool is None:
            pool = ReplayPool(self.layout, self.size, 1)
            pool_index = 0
        assert pool.capacity == self.size and pool.layout.same_as(self.layout)
        self.pool = pool
        self.pool_index = pool_index
        self.current_size = 0
        self.n_transitions_stored = 0
        self.lock = threading.Lock()                                # vestigial in the reference as well

    # ------------------------------------------------------------------ views
    @property
    def records(self):
        return self.pool.storage[self.pool_index]

    @property
    def buffers(self):
        """{key: GPU view [size, T or T+1, dim]} (replay_buffer.py:23-24)."""
        return self.layout.record_views(self.records)

    @property
    def full(self):
        return self.current_size == self.size

    # ------------------------------------------------------------------ sampling
    def sample(self, batch_size, task_to_replay=None, cp_proba=None):
        """Returns {key: GPU tensor [batch_size, dim]} (replay_buffer.py:37-55)."""
        assert self.current_size > 0
        views = EpisodeViews(self.records[:self.current_size], self.layout)
        transitions = self.sample_transitions(views, batch_size, task_to_replay=task_to_replay, cp_proba=cp_proba)
        for key in (['r', 'o_2', 'ag_2'] + list(self.buffer_shapes.keys())):
            assert key in transitions, "key %s missing from transitions" % key
        return transitions

    # ------------------------------------------------------------------ storing
    def store_episode(self, episode_batch):
        """episode_batch: {key: array [n, T or T+1, dim]} (NumPy / tensors) or an EpisodeViews of a device staging
        block (replay_buffer.py:57-72)."""
        if isinstance(episode_batch, EpisodeViews):
            n = episode_batch.records.shape[0]
        else:
            sizes = [len(episode_batch[k]) for k in episode_batch.keys()]
            assert np.all(np.array(sizes) == sizes[0])
            n = sizes[0]
        idxs = np.atleast_1d(self._get_storage_idx(n))
        staging = as_records(episode_batch, self.layout)
        self.store_records(staging, np.arange(n, dtype=np.int32), idxs)
        self.n_transitions_stored += n * self.T
        return idxs

    def store_records(self, staging, src_episodes, slots):
        """Copy staging[src_episodes[i]] into slot slots[i] of this buffer (device-side copy kernel)."""
        last = {}                                                   # later writer of a slot wins (sequential semantics)
        for s_ep, slot in zip(np.asarray(src_episodes).reshape(-1), np.asarray(slots).reshape(-1)):
            last[int(slot) + self.pool_index * self.pool.capacity] = int(s_ep)
        src = torch.as_tensor(np.fromiter(last.values(), dtype=np.int32, count=len(last))).to(staging.device)
        dst = torch.as_tensor(np.fromiter(last.keys(), dtype=np.int64, count=len(last))).to(staging.device)
        ops.store_episodes(self.pool.storage, staging, self.layout, src, dst)
        self.pool.version += 1

    def get_current_episode_size(self):
        return self.current_size

    def get_current_size(self):
        return self.current_size * self.T

    def get_transitions_stored(self):
        return self.n_transitions_stored

    def clear_buffer(self):
        self.current_size = 0
        self.pool.version += 1

    def _get_storage_idx(self, inc=None):
        inc = inc or 1
        assert inc <= self.size, "Batch committed to replay is too large!"
        free = self.size - self.current_size
        if inc <= free:                                             # replay_buffer.py:94-95
            idx = np.arange(self.current_size, self.current_size + inc)
        elif free > 0:                                              # replay_buffer.py:96-100
            head = np.arange(self.current_size, self.size)
            tail = np.random.randint(0, self.current_size, inc - free)
            idx = np.concatenate([head, tail])
        else:                                                       # replay_buffer.py:101-102
            idx = np.random.randint(0, self.size, inc)
        self.current_size = min(self.size, self.current_size + inc)
        return idx[0] if inc == 1 else idx


class RankBuffers(list):
    """The buffer list of virtual rank 0 (what `policy.buffer` is in the reference) + `.ranks`: the lists of ALL the
    virtual ranks of this process (config.py:210-214 runs once per MPI process: every rank owns a full set)."""
    ranks = None


def make_pooled_buffers(buffer_shapes, size_in_transitions, T, sample_transitions, n_logical, alias_from=None,
                        n_ranks=1):
    """The reference builds nb_tasks+1 independent buffers (config.py:210-212) and later aliases buffers 6.. to
    buffer 5 (ddpg.py:106-110).  This builds them on ONE pool; aliased logical buffers share a physical slot.
    n_ranks > 1 (DDPG virtual_ranks): one such set per virtual rank, all on the one pool -- a RankBuffers list."""
    layout = RecordLayout(buffer_shapes, T)
    n_phys = n_logical if alias_from is None else min(n_logical, alias_from + 1)
    pool = ReplayPool(layout, size_in_transitions // T, n_phys * n_ranks)
    ranks = [[ReplayBuffer(buffer_shapes, size_in_transitions, T, sample_transitions, pool=pool,
                           pool_index=v * n_phys + i) for i in range(n_phys)] + [None] * (n_logical - n_phys)
             for v in range(n_ranks)]
    if n_ranks == 1:
        return ranks[0]
    out = RankBuffers(ranks[0])
    out.ranks = [out] + ranks[1:]
    return out
