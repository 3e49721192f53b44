"""How runs of updates are launched (`for _ in range(n_batches): policy.train()`, train.py:152-153): the fused single-rank
update (curious_ddpg_update: 2 launches) over two alternating staging tensors, chains of updates captured in one hipGraph,
the ring of Adam step sizes (mpi_adam.py:30), the guarded (replayable) form.  The several-rank schedules live in
curious_amd.rank_schedules.  Mixed into curious_amd.ddpg.DDPG."""
import numpy as np
import torch

from curious_amd import _lib, dist, ops
from curious_amd.faults import HandoffFault

ALPHA_TAB = 4096        # Adam step sizes precomputed per cycle for graph replay
CHAIN = 10              # updates per chained hipGraph launch in train_batches (even: the staging tensors alternate)
LONG_CHAIN = 50         # batched experts: a longer chain when that many updates are due
MAX_CHAIN = 100         # single-rank path: train_batches(n) replays ONE graph of min(n, 100) (even) updates -- every graph
                        # launch leaves the GPU idle for ~5 us and its first update rebuilds the transposed weight copies
                        # (3.6 us); the reference's n_batches = 40 is one launch instead of four
MAX_CHAIN_GRAPHS = 4    # distinct chain lengths kept captured; further lengths fall back to chains of CHAIN
# hipStreamCaptureModeThreadLocal: HIP calls of OTHER threads (the RCCL watchdog polling events) must not invalidate a
# capture that only this thread's launches take part in
CAPTURE_MODE = 'thread_local'


class UpdateSchedulesMixin:
    def _fill_alpha_table(self):
        """Adam step sizes of the next ALPHA_TAB updates (float64 on the host exactly as mpi_adam.py:30, rounded to
        float32).  The device table is a ring indexed by (step - 1) mod ALPHA_TAB (the base is baked into captured
        launches, so it never changes); the refill is stream-ordered behind the updates that still read old entries."""
        t0 = self.Q_adam.t
        n = ALPHA_TAB
        ts = np.arange(t0 + 1, t0 + n + 1)
        tab = np.empty([n, 2], np.float32)
        pos = (ts - 1) % n
        tab[pos, 0] = self.Q_adam.alpha_table(self.Q_lr, ts)
        tab[pos, 1] = self.pi_adam.alpha_table(self.pi_lr, ts)
        # pinned + asynchronous: from pageable memory this H2D copy would block the host until the run of updates that
        # was just enqueued has finished (_keep_alpha_ahead refills BEHIND a run).  Two pinned blocks alternate: the
        # previous refill's copy was enqueued >= ALPHA_TAB - 2 * MAX_CHAIN updates ago
        pins = getattr(self, '_alpha_pins', None)
        if pins is None:
            pins = self._alpha_pins = [torch.empty([n, 2], dtype=torch.float32).pin_memory() for _ in range(2)]
            self._alpha_pin_k = 0
        pin = pins[self._alpha_pin_k]
        self._alpha_pin_k ^= 1
        pin.numpy()[:] = tab
        self._alpha_tab.copy_(pin, non_blocking=True)
        self._alpha_base = 0
        self._alpha_filled = t0 + n
        self._step_ctr.fill_(t0)

    def _keep_alpha_ahead(self):
        """Called right AFTER a run of updates was enqueued: refill the step-size ring while the GPU is busy with that
        run (the refill is ~3 ms of host arithmetic; in front of a run, as _train_device_prologue does it when it has
        to, the GPU waits for it)."""
        if self.Q_adam.t + 2 * MAX_CHAIN > self._alpha_filled:
            self._fill_alpha_table()

    def _device_loop(self):
        """The device-resident update loop applies: device-drawn batches from the pooled per-task buffers."""
        return self.rng_mode == 'device' and self._multi_buffer() and \
            getattr(self.sample_transitions, 'host_reward', None) is None

    def train_batches(self, n):
        """`for _ in range(n): policy.train()` (the inner loop of train.py:152-153) -- same updates, same order, same
        result.  On the single-rank hipGraph path runs of CHAIN updates are replayed as ONE graph launch: a graph
        boundary costs ~5 us of idle GPU on this stack (tools/graph_chain_probe.py), 7 % of an update.  With several
        ranks the eager all-reduce splits every update; the loop is then software-pipelined so that Adam of update k
        and the gradients of update k+1 share one graph launch."""
        out = None
        if self._device_loop() and dist.is_distributed() and self._allreduce == 'ipc':
            return self._train_ranks_ipc(n)
        if self._device_loop() and self.use_graph and dist.is_distributed() and not self._graph_allreduce():
            while n > 0:
                k = min(n, 1000)
                out = self._train_ranks_pipelined(k)
                n -= k
            return out
        while n > 0:
            k = 1
            chainable = not dist.is_distributed() or (self._graph_allreduce() and self.Q_adam.t % CHAIN == 0)
            if self._device_loop() and self.use_graph and chainable and n >= CHAIN and self._cur == 0:
                k = CHAIN
                if dist.is_distributed() and n >= MAX_CHAIN and self.Q_adam.t % MAX_CHAIN == 0:
                    k = MAX_CHAIN                                    # (the every-100 check falls on the chain's head)
                if not dist.is_distributed():
                    want = min(n, MAX_CHAIN) & ~1
                    chains = getattr(self, '_chains', None) or {}
                    if want in chains or len(chains) < MAX_CHAIN_GRAPHS:
                        k = want
            elif self._device_loop() and not self.use_graph and not dist.is_distributed() and n >= 2:
                k = min(n, MAX_CHAIN)                                # eager launches: one run, copies kept between updates
            out = self._train_device(k) if self._device_loop() else self.train()
            n -= k
        return out

    def train_batches_guarded(self, n):
        """train_batches(n) with the hand-off guard read SYNCHRONOUSLY (the host waits for these n updates): when an
        update of the run faulted, the parameters, moments and counters are put back to where the run started and the
        run is replayed once -- same batches (the sampler is keyed by the step counter), same step sizes, so the job
        ends bit-identical to one that never faulted.  A fault in the replay is raised.  The price is the host no longer
        running ahead of the GPU across this call (experiment.train: --fault_check sync); the default (asynchronous)
        form reads the verdict cycles later, keeps the job alive on the last good parameters and loses the frozen
        updates.  With several ranks every rank sees the fault (collective flag) and every rank replays."""
        snap = (self.theta.clone(), self._m.clone(), self._v.clone(), self._step_ctr.clone(), self.Q_adam.t,
                self.pi_adam.t, self._alpha_filled)
        out = self.train_batches(n)
        try:
            self.check_faults(wait=True)
            return out
        except HandoffFault as err:
            import warnings
            warnings.warn('%s -- replaying the %d updates of this run from the parameters it started with' % (err, n))
        self.theta.copy_(snap[0]); self._m.copy_(snap[1]); self._v.copy_(snap[2]); self._step_ctr.copy_(snap[3])
        self.Q_adam.t, self.pi_adam.t = snap[4], snap[5]
        if self._alpha_filled != snap[6]:
            self._alpha_filled = 0                                   # the ring was refilled past the run: fill it again
        self._batch_stale = True                                     # the first batch of the run is drawn again
        out = self.train_batches(n)
        self.check_faults(wait=True)                                 # a repeat is raised
        return out

    def _train_device(self, k):
        """k updates of the device-resident loop.  Single rank: each update is curious_ddpg_update -- gradients, Adam in
        the weight-gradient launch and the HER gather of the NEXT batch riding on that launch -- over two staging
        tensors used alternately, replayed from hipGraphs when use_graph is set (one graph per parity, plus one for a
        chain of CHAIN updates).  Several ranks: the gradient all-reduce splits every update into graph A (gradients)
        and graph B (Adam + next gather).  An explicit gather is issued whenever the buffers or the sampling tables
        changed since the last one, so every batch is still drawn after the latest store_episode."""
        if dist.is_distributed() and self._allreduce == 'ipc':
            return self._train_ranks_ipc(k)                          # (every update of this agent: the moments are sliced)
        self._train_device_prologue(k)
        if dist.is_distributed():
            assert k == 1 or (k in (CHAIN, MAX_CHAIN) and self._graph_allreduce() and 100 % CHAIN == 0 and MAX_CHAIN == 100)
            return self._train_device_ranks(k)
        graph = None
        if self.use_graph:
            # capturing runs the launches once for real: parameters and counter are restored by _capture, the staged
            # batch (overwritten by the gathers of a chain) is simply drawn again below
            if k == 1:
                if self._graphs[self._cur] is None:
                    cur = self._cur
                    self._graphs[cur] = self._capture(lambda: self._update_fused(cur))
                    self._batch_stale = True
                graph = self._graphs[self._cur]
            else:
                assert CHAIN <= k <= MAX_CHAIN and k % 2 == 0 and self._cur == 0
                if getattr(self, '_chains', None) is None:
                    self._chains = {}
                if k not in self._chains:
                    self._chains[k] = self._capture(lambda: [self._update_fused(i & 1, i > 0) for i in range(k)])
                    self._batch_stale = True
                graph = self._chains[k]
        if self._batch_stale:
            self._sample_packed()
            self._batch_stale = False
        if graph is not None:
            graph.replay()
            self._cur ^= (k & 1)
        else:
            for i in range(k):
                self._update_fused(self._cur, i > 0)
                self._cur ^= 1
        self._staged = self._pp[self._cur]
        self.Q_adam.t += k
        self.pi_adam.t += k
        self._keep_alpha_ahead()
        return self._losses[0], self._Q_pi

    def _train_device_prologue(self, k):
        if self._tables_stale():
            self._refresh_device_tables()
            self._batch_stale = True
        if self.Q_adam.t + k > self._alpha_filled or self._alpha_filled == 0:
            self._fill_alpha_table()
        if self._pp is None:
            shape = [self._Bt, self._layout.batch_stride]
            self._pp = [self._new(shape) for _ in range(2)]
            self._cur = 0
            self._batch_stale = True
        if self._staged is not self._pp[self._cur]:
            self._staged = self._pp[self._cur]
            self._batch_stale = True
        self._layout_for_batch = self._layout

    def _update_fused(self, p, chained=False):
        """chained: the previous launch on this stream was this very update (inside a captured chain) -- the only case in
        which the library may trust the transposed copies it keeps in the workspace; everything else (a single update, the
        head of a chain) has them rebuilt from theta first."""
        S = self.sample_transitions
        ops.ddpg_update(self.net_cfg, self.theta, self.theta_target, self._pp[p], self._layout, self._Bt,
                        self._workspace, self.grad, self._losses, self._Q_pi, self._m, self._v,
                        step_ctr=self._step_ctr, alpha_tab=self._alpha_tab, tab_base=self._alpha_base,
                        o_stats=self.o_stats.state if self.normalize_obs else None,
                        g_stats=self.g_stats.state if self.normalize_obs else None,
                        next_batch=self._pp[p ^ 1], storage=self._pool.storage, buf_stride=self._pool.buf_stride,
                        tasks=S.tasks, params=S.params(self.clip_obs, self.relative_goals), rng=self._rng_desc,
                        params_unchanged=chained)

    def _kept_copies(self):
        """curious_transposed_t of this agent's workspace: handed to the stand-alone optimiser launch of the multi-rank
        path so that the gradient launch that follows it inside the same graph need not rebuild the copies."""
        if getattr(self, '_kept', None) is None:
            self._kept = ops.ddpg_transposed(self.net_cfg, self._Bt, self._workspace)
        return self._kept

    def _fault_guard(self):
        """The description of _kept_copies() without the copies: an optimiser call given it only honours the fault word of
        the gradient workspace and the collective fault flag in the gradient vector (the one-at-a-time train() path)."""
        if getattr(self, '_guard', None) is None:
            g = _lib.Transposed()
            k = self._kept_copies()
            g.fault, g.fault_flag = k.fault, k.fault_flag
            self._guard = g
        return self._guard

    def _capture(self, fn):
        """Capture `fn`'s kernel launches into a hipGraph (after one eager warm-up on a side stream)."""
        ctr = self._step_ctr.clone()
        state = (self.theta.clone(), self._m.clone(), self._v.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            fn()
        # undo the side effects of the warm-up / capture runs
        self._step_ctr.copy_(ctr)
        self.theta.copy_(state[0]); self._m.copy_(state[1]); self._v.copy_(state[2])
        return g
