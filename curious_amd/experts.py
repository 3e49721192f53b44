"""Batched multi-policy update for structure='task_experts' (BASELINE configs[4]).

The reference keeps one DDPG per task on shared replay buffers (train.py:285-291) and trains them one after the other,
one expert per epoch (train.py:65-121); expert t_id samples its minibatch from buffer t_id + 1 and relabels to its own
task (ddpg.py:302-318,335).  The experts are independent given the buffers, so here all N of them go through ONE launch
sequence per update (curious_ddpg_update_experts: the 2 launches of a single-agent update with the expert on grid.z /
grid.y).

Layout: every expert's update state (parameters, target, gradient, Adam moments, step counter, step-size ring,
workspace, the two staged batches, sampling tables, loss outputs) is carved, in the same order, out of row e of one
[N, stride] float32 tensor, so that expert e's copy of anything is `stride` floats behind expert 0's.
"""
import numpy as np
import torch

from curious_amd import _lib, dist, ops
from curious_amd.ddpg import CAPTURE_MODE, CHAIN, LONG_CHAIN

SEED_STRIDE_SAMPLER = 104729            # DDPG._refresh_device_tables: sampler key = seed * 104729 + ...


class _Carver:
    """Allocator handed to DDPG(_alloc=...): consecutive 256-byte aligned pieces of one slab row (or, in measuring mode,
    stand-alone tensors while the total is added up).  The piece named 'grad' comes from `grad_row` instead: the
    experts' gradient vectors form one contiguous [N, P] block, so that several ranks sum all of them in ONE all-reduce."""

    def __init__(self, device, row=None, grad_row=None):
        self.device, self.row, self.grad_row, self.off = device, row, grad_row, 0

    def __call__(self, shape, dtype, name=None):
        n = int(np.prod(shape))
        if name == 'grad' and dtype == torch.float32:
            if self.row is None:
                return torch.zeros(shape, dtype=dtype, device=self.device)
            assert n <= self.grad_row.numel()
            piece = self.grad_row[:n]
            piece.zero_()
            return piece.view(*shape)
        words = n * (2 if dtype == torch.int64 else 1)
        start = self.off
        self.off += (words + 63) & ~63
        if self.row is None:
            return torch.zeros(shape, dtype=dtype, device=self.device)
        if self.off > self.row.numel():
            raise RuntimeError('expert slab too small: the experts do not allocate identically')
        piece = self.row[start:start + words]
        piece.zero_()
        if dtype != torch.float32:
            piece = piece.view(dtype)
        return piece.view(*shape)


class ExpertBank:
    """N task experts updated together.

    `make_expert(t_id, **hooks)` must build the DDPG of expert t_id (config.configure_ddpg with t_id) forwarding the
    hook keyword arguments to the DDPG constructor.  Expert seeds must be consecutive (seed_0 + t_id): the batched
    launch derives expert e's sampler key from expert 0's.

    Several ranks (one process per GPU): every rank holds all N experts, its own buffers and RNG streams; per update the
    experts' gradients -- one contiguous [N, P] block -- are summed over the ranks by ONE all-reduce between the gradient
    launches and the optimiser launch (the reference: 2 N MpiAdam.update Allreduces per round of updates,
    train.py:65-121, mpi_adam.py:21-35).
    """

    def __init__(self, make_expert, n_experts):
        self.n = int(n_experts)
        dev = torch.device('cuda', torch.cuda.current_device())
        probe_alloc = _Carver(dev)
        probe = make_expert(0, _alloc=probe_alloc)
        probe._prealloc_device_loop()
        self.stride = (probe_alloc.off + 63) & ~63
        self.grad_stride = (probe.P_total + 63) & ~63
        del probe
        self.slab = torch.zeros([self.n, self.stride], dtype=torch.float32, device=dev)
        self.grads = torch.zeros([self.n, self.grad_stride], dtype=torch.float32, device=dev)
        self.experts = []
        for e in range(self.n):
            x = make_expert(e, _alloc=_Carver(dev, self.slab[e], self.grads[e]))
            x._prealloc_device_loop()
            self.experts.append(x)
        x0 = self.experts[0]
        for e, x in enumerate(self.experts):
            if x.seed != x0.seed + e:
                raise ValueError('expert seeds must be consecutive (seed_0 + t_id)')
            if not x._device_loop():
                raise ValueError("the batched update needs rng_mode='device' and per-task buffers")
            for name in ('theta', 'theta_target', '_m', '_v', '_workspace', '_losses', '_Q_pi', '_step_ctr',
                         '_alpha_tab', '_tables'):
                a, b = getattr(x, name), getattr(x0, name)
                assert a.data_ptr() - b.data_ptr() == 4 * e * self.stride, name
            for nz, nz0 in ((x.o_stats, x0.o_stats), (x.g_stats, x0.g_stats)):      # every expert has its own normalisers
                assert nz.state.data_ptr() - nz0.state.data_ptr() == 4 * e * self.stride
            assert x.grad.data_ptr() - x0.grad.data_ptr() == 4 * e * self.grad_stride
        self.use_graph = bool(x0.use_graph)
        self._graphs = {}
        self._cur = 0
        self.batched = True                     # False after the library refused the shapes (sequential updates then)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.experts[i]

    def __iter__(self):
        return iter(self.experts)

    # ------------------------------------------------------------------ updates
    def trainable(self):
        """Every expert can draw a batch once any buffer holds an episode (ddpg.py:302-318)."""
        x0 = self.experts[0]
        return sum(x0.buffer[i].current_size for i in range(x0.nb_tasks + 1)) > 0

    def _prologue(self, k):
        for x in self.experts:
            if x._cur != self._cur:
                x._cur = self._cur
                x._batch_stale = True
            x._train_device_prologue(k)
        for x in self.experts:
            if x._batch_stale:
                x._sample_packed()
                x._batch_stale = False

    def _update_all(self, p, chained=False):
        x0 = self.experts[0]
        S = x0.sample_transitions
        ops.ddpg_update_experts(x0.net_cfg, self.n, self.stride, self.grad_stride, SEED_STRIDE_SAMPLER, x0.theta,
                                x0.theta_target, x0._pp[p], x0._layout, x0._Bt, x0._workspace, x0.grad,
                                x0._losses, x0._Q_pi, x0._m, x0._v, x0._step_ctr, x0._alpha_tab, x0._alpha_base,
                                x0._pp[p ^ 1], x0._pool.storage, x0._pool.buf_stride, S.tasks,
                                S.params(x0.clip_obs, x0.relative_goals), x0._rng_desc, params_unchanged=chained,
                                o_stats=x0.o_stats.state if x0.normalize_obs else None,
                                g_stats=x0.g_stats.state if x0.normalize_obs else None)

    # the two halves of an update on several ranks: the all-reduce of the gradient block sits between them
    def _grads_all(self, p, chained=False):
        """Gradients of every expert from the batches staged in tensor p; the HER gather of every expert's NEXT batch
        into tensor p ^ 1 rides in spare workgroups of the row-local launch."""
        x0 = self.experts[0]
        S = x0.sample_transitions
        ops.ddpg_grads_experts(x0.net_cfg, self.n, self.stride, self.grad_stride, x0.theta, x0.theta_target, x0._pp[p],
                               x0._layout, x0._Bt, x0._workspace, x0.grad, x0._losses, x0._Q_pi, x0._step_ctr,
                               params_unchanged=chained, seed_stride=SEED_STRIDE_SAMPLER, next_batch=x0._pp[p ^ 1],
                               storage=x0._pool.storage, buf_stride=x0._pool.buf_stride, tasks=S.tasks,
                               params=S.params(x0.clip_obs, x0.relative_goals), rng=x0._rng_desc,
                               o_stats=x0.o_stats.state if x0.normalize_obs else None,
                               g_stats=x0.g_stats.state if x0.normalize_obs else None)

    def _adam_all(self):
        """Adam of every expert from the summed gradients (one launch, grid.y = expert)."""
        x0 = self.experts[0]
        S = x0.sample_transitions
        ops.adam_update_and_sample_experts(self.n, self.stride, self.grad_stride, SEED_STRIDE_SAMPLER, x0.theta, x0._m,
                                           x0._v, x0.grad, x0.off_pi, x0.P_total - x0.off_pi, x0._alpha_tab,
                                           x0._step_ctr, x0._alpha_base, None, 0, x0._layout, S.tasks,
                                           S.params(x0.clip_obs, x0.relative_goals), x0._rng_desc, x0._Bt, None,
                                           keep=x0._kept_copies())

    def _allreduce(self):
        dist.allreduce_sum_(self.grads)                      # ONE collective: N x P floats, SUM (mpi_adam.py:26)

    def _ranks_update_all(self, p, chained=False):
        self._grads_all(p, chained)
        self._allreduce()
        self._adam_all()

    def _capture(self, fn):
        """Capture `fn` after one eager warm-up; the slab (all experts' state) is restored afterwards."""
        saved = self.slab.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            fn()
        self.slab.copy_(saved)
        return g

    def _graph(self, key, fn):
        if key not in self._graphs:
            self._graphs[key] = self._capture(fn)
            for x in self.experts:                           # the warm-up overwrote the staged batches
                x._sample_packed()
        return self._graphs[key]

    def _train(self, k):
        """k updates of every expert (k = 1, CHAIN or LONG_CHAIN)."""
        if dist.is_distributed():
            return self._train_ranks(k)
        self._prologue(k)
        p0 = self._cur
        if self.use_graph:
            self._graph((k, p0), lambda: [self._update_all((p0 + i) & 1, i > 0) for i in range(k)]).replay()
        else:
            for i in range(k):
                self._update_all((p0 + i) & 1, i > 0)
        self._cur ^= (k & 1)
        self._advance(k)

    def _advance(self, k):
        for x in self.experts:
            x._cur = self._cur
            x._staged = x._pp[x._cur]
            x.Q_adam.t += k
            x.pi_adam.t += k
            x._keep_alpha_ahead()

    def _train_ranks(self, k):
        """k updates of every expert on several ranks (the staging tensors alternate like on one rank: the gradient
        launch of an update draws the next batch).  With the collective captured (dist.captured_allreduce_ok) a chain of
        k updates is ONE graph launch; otherwise the eager all-reduce separates graph A (gradients + next gather) from
        graph B (optimiser), software-pipelined as B(i) + A(i + 1)."""
        self._prologue(k)
        p = self._cur
        x0 = self.experts[0]
        if x0.Q_adam.t % 100 == 0:                           # C4 (mpi_adam.py:42-50): every 100 updates, parameters at
            for x in self.experts:                           # rest (chains start on multiples of their length)
                x._check_synced()
        if not self.use_graph:
            for i in range(k):
                self._ranks_update_all((p + i) & 1, i > 0)
        elif dist.captured_allreduce_ok():
            self._graph(('ranks', k, p), lambda: [self._ranks_update_all((p + i) & 1, i > 0) for i in range(k)]).replay()
        else:
            ga = [self._graph(('A', q), lambda q=q: self._grads_all(q)) for q in (0, 1)]
            gb = self._graph(('B',), self._adam_all)
            gba = [self._graph(('BA', q), lambda q=q: (self._adam_all(), self._grads_all(q, True))) for q in (0, 1)]
            ga[p].replay()
            for i in range(1, k):
                self._allreduce()
                gba[(p + i) & 1].replay()
            self._allreduce()
            gb.replay()
        self._cur ^= (k & 1)
        self._advance(k)

    def train(self):
        """One update of every expert.  Returns [(critic_loss, Q_pi)] per expert (GPU tensors)."""
        return self.train_batches(1)

    def train_batches(self, n):
        """n x [policy[e].train() for every expert e] -- bit-identical to the sequential loops."""
        if not self.batched:
            return [x.train_batches(n) for x in self.experts]
        try:
            while n > 0:
                k = 1
                if self.use_graph and self._cur == 0 and n >= CHAIN:
                    k = LONG_CHAIN if n >= LONG_CHAIN else CHAIN
                if dist.is_distributed() and k > 1:
                    # chains start on multiples of their length so that the every-100 check falls on a chain head
                    t = self.experts[0].Q_adam.t
                    k = LONG_CHAIN if (n >= LONG_CHAIN and t % LONG_CHAIN == 0) else CHAIN if t % CHAIN == 0 else 1
                self._train(k)
                n -= k
        except _lib.CuriousHipError as err:
            if 'lean' not in str(err):
                raise
            self.batched = False                             # shapes outside the lean route: sequential from now on
            return [x.train_batches(n) for x in self.experts]
        return [(x._losses[0], x._Q_pi) for x in self.experts]

    def train_batches_guarded(self, n):
        """train_batches(n) with the hand-off guard read synchronously and a faulted run replayed once from the state it
        started with (DDPG.train_batches_guarded for the bank: the slab holds every expert's parameters, moments,
        counters, step-size rings and fault words)."""
        from curious_amd.ddpg import HandoffFault
        snap = self.slab.clone()
        ts = [(x.Q_adam.t, x.pi_adam.t, x._alpha_filled) for x in self.experts]
        cur = self._cur
        out = self.train_batches(n)
        try:
            self.check_faults(wait=True)
            return out
        except HandoffFault as err:
            import warnings
            warnings.warn('%s -- replaying the %d updates of this run from the state it started with' % (err, n))
        for x in self.experts:                                # the check above stopped at the first expert that raised
            try:
                x.check_faults(wait=True)
            except HandoffFault:
                pass
        self.slab.copy_(snap)
        self._cur = cur
        for x, (tq, tp, filled) in zip(self.experts, ts):
            x.Q_adam.t, x.pi_adam.t = tq, tp
            if x._alpha_filled != filled:
                x._alpha_filled = 0
            x._cur = cur
            x._staged = x._pp[cur]
            x._batch_stale = True
        out = self.train_batches(n)
        self.check_faults(wait=True)
        return out

    def update_target_net(self):
        """Every expert's target update and fault bookkeeping, whatever one of them reports: a HandoffFault of expert e is
        raised AFTER the loop, so that the experts behind it get their Polyak step, their tick and their fault-check copy in
        the same cycle (experiment.train.FaultTolerance goes on with the job; the experts' ticks must stay in phase)."""
        self._each(lambda x: x.update_target_net())

    def check_faults(self, wait=True):
        self._each(lambda x: x.check_faults(wait))

    def _each(self, fn):
        from curious_amd.ddpg import HandoffFault
        first = None
        for x in self.experts:
            try:
                fn(x)
            except HandoffFault as err:
                first = first or err
        if first is not None:
            raise first
