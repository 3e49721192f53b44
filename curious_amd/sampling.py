"""Sampling half of the agent (DDPG.sample_batch / stage_batch baselines/her/ddpg.py:251-366): replay proportions per
buffer (ddpg.py:255-318), the device tables the Philox-drawn HER gather reads (one row per virtual rank), the NumPy-stream
plan of parity mode, the packed staging tensor.  Mixed into curious_amd.ddpg.DDPG."""
import numpy as np
import torch

from curious_amd import _lib, ops
from curious_amd.dist import RANK_SEED_STRIDE
from curious_amd.her import upload_plan


class SamplingMixin:
    def _proportions(self, bufs=None):
        """ddpg.py:255-286 (curious, multi-buffer) and ddpg.py:303-318 (task_experts).  bufs: the buffers of one virtual
        rank (default: self.buffer) -- every rank splits ITS minibatch by the sizes of ITS buffers."""
        nb1 = self.nb_tasks + 1
        bufs = self.buffer if bufs is None else bufs
        sizes = np.array([bufs[i].current_size * self.T for i in range(nb1)])
        prop = np.zeros([nb1])
        if self.structure == 'curious':
            if sizes[1:].sum() < self.T:
                valid = np.array([0])
                n_valid = 1
                prop = sizes / sizes.sum() * self.batch_size
            else:
                valid = np.argwhere(sizes[1:] > 0).reshape(-1)
                n_valid = len(valid)
                if self.task_replay == 'replay_task_random_buffer':
                    proba = 1 / valid.size * np.ones([n_valid])
                elif self.task_replay == 'replay_task_cp_buffer':
                    CP = np.asarray(self.cp)[valid]
                    if CP.sum() == 0:
                        proba = (1 / n_valid) * np.ones([n_valid])
                    else:
                        proba = self.eps_task * (1 / n_valid) * np.ones([n_valid]) + \
                            (1 - self.eps_task) * CP / CP.sum()
                    proba[-1] = 1 - proba[:-1].sum()
                else:
                    raise NotImplementedError(self.task_replay)
                prop[valid + 1] = proba * self.batch_size
            prop = prop.astype(int)
            for i in range(self.batch_size - prop.sum()):
                prop[valid[i % n_valid] + 1] += 1
        else:
            valid = np.argwhere(sizes > 0).reshape(-1)
            n_valid = len(valid)
            if sizes[self.t_id + 1] > 0:
                prop[self.t_id + 1] = 1
            else:
                prop[valid] = 1 / len(valid)
            prop *= self.batch_size
            prop = prop.astype(int)
            for i in range(self.batch_size - prop.sum()):
                prop[valid[i % n_valid]] += 1
        return prop.astype(int)

    def _task_of_buffer(self, i):
        if self.structure == 'curious':
            return i - 1 if i > 0 else None                          # ddpg.py:329-333
        return self.t_id                                             # ddpg.py:335

    def _sizes_key(self):
        return (self._pool.version,) + tuple(bl[i].current_size for bl in self._rank_buffers
                                             for i in range(self.nb_tasks + 1))

    def _tables_stale(self):
        """The device sampling tables follow the buffers: task experts share their buffers, so an episode stored through
        ANOTHER expert (train.py:99) must be seen here too, like the reference's sample_batch reading current_size."""
        return self._tables_dirty or getattr(self, '_tables_sizes', None) != self._sizes_key()

    def _prealloc_device_loop(self):
        """Allocate what the device-resident update loop otherwise allocates lazily (ExpertBank: identical slab layouts)."""
        if self._pp is None:
            shape = [self._Bt, self._layout.batch_stride]
            self._pp = [self._new(shape) for _ in range(2)]
            self._cur = 0
        if getattr(self, '_tables', None) is None:
            n = (4 * (self.nb_tasks + 1) + 1) * self.V
            self._tables = self._new([n], torch.int32)
            self._tables_host = torch.zeros(n, dtype=torch.int32).pin_memory()

    def _refresh_device_tables(self):
        self.settle()
        nb1 = self.nb_tasks + 1
        task = np.array([-1 if self._task_of_buffer(i) is None else self._task_of_buffer(i) for i in range(nb1)],
                        np.int32)
        rows = []
        for v, bufs in enumerate(self._rank_buffers):                # one table row per virtual rank
            prop = self._proportions(bufs)
            assert prop.sum() == self.batch_size                     # ddpg.py:323
            prefix = np.concatenate([[0], np.cumsum(prop)]).astype(np.int32)
            alias = np.array([bufs[i].pool_index for i in range(nb1)], np.int32)
            cur = np.array([bufs[i].current_size for i in range(nb1)], np.int32)       # per LOGICAL buffer
            for i in range(nb1):
                assert prop[i] == 0 or bufs[i].current_size > 0      # replay_buffer.py:43
            rows += [prefix, alias, task, cur]
            if v == 0:
                self.proportions = prop
        host = np.concatenate(rows)
        if getattr(self, '_tables', None) is None or self._tables.numel() != host.size:
            self._tables = self._new([host.size], torch.int32)
            self._tables_host = torch.zeros(host.size, dtype=torch.int32).pin_memory()
        # pinned + asynchronous: the previous upload from this buffer finished long ago (every cycle has a D2H sync)
        self._tables_host.numpy()[:] = host
        self._tables.copy_(self._tables_host, non_blocking=True)
        n0 = nb1 + 1
        r = _lib.SampleRng()
        r.seed = (self.seed * 104729 + 7 + self._grank0() * RANK_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF
        r.step_ctr = self._step_ctr.data_ptr()
        r.step_host = 0
        r.prop_prefix = self._tables[:n0].data_ptr()
        r.buf_alias = self._tables[n0:n0 + nb1].data_ptr()
        r.buf_task = self._tables[n0 + nb1:n0 + 2 * nb1].data_ptr()
        r.cur_size = self._tables[n0 + 2 * nb1:].data_ptr()
        r.nbuf = nb1
        if self.V > 1:
            r.rank_rows, r.rank_tab_stride, r.rank_seed_stride = self.batch_size, 4 * nb1 + 1, RANK_SEED_STRIDE
        self._rng_desc = r
        self._tables_dirty = False
        self._tables_sizes = self._sizes_key()

    def _multi_buffer(self):
        return self.structure in ('curious', 'task_experts') and \
            ('buffer' in self.task_replay or self.task_replay == 'hand_designed')

    def _host_reward_fixup(self, packed, layout):
        """Host-evaluated reward (real-env parity-audit mode, her.py:166-176): the batch was gathered un-clipped so that
        the reward sees the sampler's goals; the clip of ddpg.py:350-353 follows (torch, off the throughput path)."""
        S = self.sample_transitions
        S.apply_host_reward(packed, layout)
        for key in ('o', 'g', 'o_2', 'g_2'):
            off, dim = layout.batch_cols[key]
            packed[:, off:off + dim].clamp_(-self.clip_obs, self.clip_obs)

    def _sample_packed(self):
        """One packed, clipped, permuted minibatch [batch_size, stride] on the GPU."""
        S = self.sample_transitions
        host_r = getattr(S, 'host_reward', None) is not None
        if host_r and self.relative_goals:
            raise NotImplementedError('a host-evaluated reward with relative_goals is not supported')
        P = S.params(np.inf if host_r else self.clip_obs, self.relative_goals)
        B = self._Bt                                                 # (virtual ranks: V minibatches of batch_size rows)
        if self._multi_buffer():
            layout = self._layout
            if self._staged is None or self._staged.shape != (B, layout.batch_stride):
                self._staged = torch.zeros([B, layout.batch_stride], dtype=torch.float32, device=self.device)
            if self.rng_mode == 'device':
                if self._tables_stale():
                    self._refresh_device_tables()
                ops.her_sample(self._pool.storage, self._pool.buf_stride, layout, S.tasks, P, B, self._staged,
                               rng=self._rng_desc)
            else:
                self.proportions = self._proportions()
                assert self.proportions.sum() == B                   # ddpg.py:323
                ep, t, uh, uo, bufi, ttr = [], [], [], [], [], []
                for i in range(self.nb_tasks + 1):                   # ddpg.py:327-336
                    n_i = int(self.proportions[i])
                    if n_i > 0:
                        buf = self.buffer[i]
                        assert buf.current_size > 0                  # replay_buffer.py:43
                        d = S.draw(buf.current_size, self.T, n_i)
                        ep.append(d[0]); t.append(d[1]); uh.append(d[2]); uo.append(d[3])
                        bufi.append(np.full(n_i, buf.pool_index, np.int32))
                        task = self._task_of_buffer(i)
                        ttr.append(np.full(n_i, -1 if task is None else task, np.int32))
                shuffle_inds = np.arange(B)
                np.random.shuffle(shuffle_inds)                      # ddpg.py:338-339
                out_row = np.empty(B, np.int32)
                out_row[shuffle_inds] = np.arange(B)                 # out[j] = tmp[shuffle_inds[j]] (ddpg.py:345)
                plan = upload_plan(B, np.concatenate(ep), np.concatenate(t), np.concatenate(uh), np.concatenate(uo),
                                   buf=np.concatenate(bufi), ttr=np.concatenate(ttr), out_row=out_row)
                ops.her_sample(self._pool.storage, self._pool.buf_stride, layout, S.tasks, P, B, self._staged,
                               plan=plan)
            if host_r:
                self._host_reward_fixup(self._staged, layout)
            self._layout_for_batch = layout
            return self._staged
        # single buffer (flat, or the *_task_transition replay modes): ddpg.py:288-299,320,348
        buf = self.buffer
        layout = buf.layout
        cp_proba = None
        if self.structure == 'curious' and self.task_replay == 'replay_cp_task_transition':
            CP = np.asarray(self.cp, dtype=np.float64).copy()
            if CP.sum() == 0:
                cp_proba = (1 / self.nb_tasks) * np.ones([self.nb_tasks])
            else:
                cp_proba = self.eps_task * (1 / self.nb_tasks) * np.ones([self.nb_tasks]) + \
                    (1 - self.eps_task) * CP / CP.sum()
            cp_proba[-1] = 1 - cp_proba[:-1].sum()
        assert buf.current_size > 0
        ep, t, uh, uo, given = S.draw(buf.current_size, self.T, B, cp_proba)
        plan = upload_plan(B, ep, t, uh, uo, buf=np.full(B, buf.pool_index, np.int32), ttr=given)
        if self._staged is None or self._staged.shape != (B, layout.batch_stride):
            self._staged = torch.zeros([B, layout.batch_stride], dtype=torch.float32, device=self.device)
        ops.her_sample(buf.pool.storage, buf.pool.buf_stride, layout, S.tasks, P, B, self._staged, plan=plan)
        if host_r:
            self._host_reward_fixup(self._staged, layout)
        self._layout_for_batch = layout
        return self._staged

    def sample_batch(self):
        """Returns the staged arrays in the reference's order (ddpg.py:251-360) as GPU views:
        ag, g, o, task_descr, u, o_2, g_2, r for the multi-task structures."""
        packed = self._sample_packed()
        views = self._layout_for_batch.batch_views(packed)
        return [views[key] for key in self.stage_shapes.keys()]

    def stage_batch(self, batch=None):
        """ddpg.py:362-366.  With batch=None a fresh minibatch is sampled straight into the staging tensor."""
        if batch is None:
            self._sample_packed()
            return
        assert len(self.stage_shapes) == len(batch)
        layout = self._layout
        host = np.zeros([self._Bt, layout.batch_stride], np.float32)
        for key, arr in zip(self.stage_shapes.keys(), batch):
            off, dim = layout.batch_cols[key]
            a = arr.detach().cpu().numpy() if isinstance(arr, torch.Tensor) else np.asarray(arr)
            host[:, off:off + dim] = a.reshape(self._Bt, dim)
        self._staged = torch.from_numpy(host).to(self.device)
        self._layout_for_batch = layout
