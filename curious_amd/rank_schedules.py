"""The update on several ranks (MpiAdam.update / check_synced, baselines/common/mpi_adam.py:21-50): gradients | all-reduce
(SUM) | optimiser as split or chained graphs with the RCCL collective captured or eager (software-pipelined), the fused IPC
all-reduce + Adam kernel (opt-in, csrc/ipc.hip), the asynchronous replica checksum.  Mixed into curious_amd.ddpg.DDPG."""
import torch

from curious_amd import _lib, dist, ops
from curious_amd.update_schedules import CHAIN, MAX_CHAIN


class RankSchedulesMixin:
    def _train_ranks_pipelined(self, n):
        """n updates on several ranks: A(0); [all-reduce; B(k)+A(k+1)] x (n-1); all-reduce; B(n-1), where A = the 8
        gradient launches and B = Adam + the gather of the next batch.  Same launches in the same order as n x train()."""
        self._train_device_prologue(n)
        self._rank_graphs()
        if self._batch_stale:
            self._sample_packed()
            self._batch_stale = False
        t = self.Q_adam.t
        if t % 100 == 0:
            self._check_synced()
        p = self._cur
        self._graph_a[p].replay()
        for i in range(1, n):
            dist.allreduce_sum_(self.grad)                           # C1+C2 fused; SUM, not mean (ddpg.py:452)
            p ^= 1                                                   # the gradient launch drew the next batch
            if (t + i) % 100 == 0:                                   # C4 between the two halves, parameters at rest
                self._graph_b.replay()
                self._check_synced()
                self._graph_a[p].replay()
            else:
                self._graph_ba[p].replay()
        dist.allreduce_sum_(self.grad)
        self._graph_b.replay()
        self._cur = p ^ 1
        self._staged = self._pp[self._cur]
        self.Q_adam.t += n
        self.pi_adam.t += n
        self._keep_alpha_ahead()
        return self._losses[0], self._Q_pi

    def _ipc_setup(self):
        """Map every rank's gradient vector, parameter vector and flag block into this process (once)."""
        if getattr(self, '_ipc', None) is not None:
            return
        ws = dist.world_size()
        if ws > 8 or self.P_total % ws:
            raise _lib.CuriousHipError("_allreduce='ipc' needs a world size <= 8 that divides the parameter count")
        blk = getattr(self, '_ipc_block', None)
        if blk is None:
            raise _lib.CuriousHipError("_allreduce='ipc' has to be chosen when the agent is built (its parameter and "
                                       'gradient vectors live in a block the peers can map)')
        words = torch.zeros(3, dtype=torch.int32, device=self.device)     # [blocks done, a wait gave up, epoch]
        torch.cuda.synchronize()
        blk.connect()
        peers = _lib.IpcPeers()
        peers.world, peers.rank = ws, dist.rank()
        for r in range(ws):
            peers.grad[r], peers.stage[r], peers.flags[r] = blk.peer_ptr(r, 0), blk.peer_ptr(r, 1), blk.peer_flags(r)
        self._ipc = dict(peers=peers, words=words, err_pin=torch.zeros(1, dtype=torch.int32).pin_memory(),
                         err_ev=torch.cuda.Event(), err_pending=False)

    def _ipc_update(self, p, chained):
        """One update: gradients of the batch staged in tensor p (+ the gather of the next batch), then the one kernel
        that sums the ranks' gradients slice by slice, runs Adam on the owned slice, hands the new slices round and
        rebuilds the transposed copies (so the next gradient launch is told params_unchanged)."""
        if self.use_graph:
            g = self._ipc.setdefault('graphs', {})
            if (p, chained) not in g:
                g[(p, chained)] = self._capture(lambda: self._grads_next(p, chained))
            g[(p, chained)].replay()
        else:
            self._grads_next(p, chained)
        w = self._ipc['words']
        ops.allreduce_adam_ipc(self._ipc['peers'], self.theta, self._m, self._v, self.off_pi, self.P_total - self.off_pi,
                               self._alpha_tab, self._step_ctr, self._alpha_base, w[2:3], w[0:1], w[1:2],
                               self._kept_copies())

    def _train_ranks_ipc(self, n):
        """n updates on several ranks through curious_allreduce_adam_ipc (DDPG(_allreduce='ipc')).  Same gradients, same
        step sizes, same order as the RCCL path; sums in rank order."""
        self._train_device_prologue(n)
        self._ipc_setup()
        if self._batch_stale:
            self._sample_packed()
            self._batch_stale = False
        # the verdict of the previous run (its copy arrived long ago), agreed between the ranks: a rank whose wait gave up
        # skipped an epoch and its peers copied a stale slice -- every rank raises, at the same point of the job
        self._ipc_verdict(collective=True)
        p = self._cur
        for i in range(n):
            if (self.Q_adam.t + i) % 100 == 0:
                self._check_synced()
            self._ipc_update(p, chained=i > 0)
            p ^= 1
        # "a wait for a peer gave up" is read at the end of EVERY run: the rank skipped that epoch's arithmetic, the
        # replicas may have parted.  The copy is asynchronous; the verdict is taken when the next run begins (or by
        # check_faults(wait=True))
        ipc = self._ipc
        ipc['err_pin'].copy_(ipc['words'][1:2], non_blocking=True)
        ipc['err_ev'].record()
        ipc['err_pending'] = True
        self._cur = p
        self._staged = self._pp[self._cur]
        self.Q_adam.t += n
        self.pi_adam.t += n
        self._keep_alpha_ahead()
        return self._losses[0], self._Q_pi

    def _ipc_verdict(self, wait=True, collective=False):
        """collective (every rank calls at the same point: the head of a run of updates): the ranks agree on the verdict
        over the host-side group -- ADVICE r5: the timed-out rank alone used to raise, its peers went on with a stale slice
        until the next check_synced."""
        ipc = getattr(self, '_ipc', None)
        if ipc is None:
            return
        mine = False
        if ipc['err_pending']:
            if wait:
                ipc['err_ev'].synchronize()
            elif not ipc['err_ev'].query():
                return
            ipc['err_pending'] = False
            mine = bool(int(ipc['err_pin'][0]))
        anyone = dist.host_any(mine) if collective else mine
        if anyone:
            who = ('rank %d' % dist.rank()) if mine else 'a peer rank'
            raise _lib.CuriousHipError('curious_allreduce_adam_ipc: a wait for a peer rank gave up (%s): that epoch was '
                                       'skipped there, the replicas may differ' % who)

    def _rank_graphs(self):
        """The split update graphs of the several-rank path with an eager collective: A[p] = gradients of the batch in
        staging tensor p (+ the gather of the next batch into the other one), B = the optimiser, BA[p] = B then A[p]."""
        if getattr(self, '_graph_a', None) is None:
            self._graph_a = [self._capture(lambda p=p: self._grads_next(p)) for p in (0, 1)]
            self._graph_b = self._capture(self._adam_only)
            self._graph_ba = [self._capture(lambda p=p: (self._adam_only(), self._grads_next(p, True))) for p in (0, 1)]
            self._batch_stale = True

    def _check_synced(self, wait=False):
        """mpi_adam.py:42-50 (every 100 updates) off the critical path: a 128-bit checksum of the fused parameter vector
        and rank 0's copy of it go to pinned host memory asynchronously; the comparison happens at the NEXT check (or at
        finish_sync_checks()), when the copy has long completed -- no host wait inside the update loop."""
        if not dist.is_distributed():
            return
        self.finish_sync_checks()
        if getattr(self, '_sync_buf', None) is None:
            self._sync_buf = (torch.zeros(2, dtype=torch.int64, device=self.device),
                              torch.zeros(2, dtype=torch.int64, device=self.device),
                              torch.zeros(4, dtype=torch.int64).pin_memory(), torch.cuda.Event())
        mine, root, host, ev = self._sync_buf
        ops.param_checksum(self.theta, mine)
        root.copy_(mine)
        dist.broadcast_(root, 0)
        host[:2].copy_(mine, non_blocking=True)
        host[2:].copy_(root, non_blocking=True)
        ev.record()
        self._sync_pending = self.Q_adam.t
        if wait:
            self.finish_sync_checks()

    def finish_sync_checks(self):
        t = getattr(self, '_sync_pending', None)
        if t is None:
            return
        _, _, host, ev = self._sync_buf
        ev.synchronize()
        self._sync_pending = None
        if not torch.equal(host[:2], host[2:]):                      # an exception, not an assert: survives python -O
            raise dist.RankDivergence('parameters diverged between ranks (rank %d, detected at update %d)' %
                                      (dist.rank(), t))

    @staticmethod
    def _graph_allreduce():
        """Capture the RCCL all-reduce inside the update graph (one graph launch per chain of updates instead of graph +
        eager collective + graph per update)?  Decided by curious_amd.dist.captured_allreduce_ok: forced by
        CURIOUS_GRAPH_ALLREDUCE=0/1, otherwise by a collective self-test at the first use."""
        return dist.captured_allreduce_ok()

    def _ranks_update(self, p, chained=False):
        """One update on several ranks: gradients of the batch staged in tensor p -- the HER gather of the next batch into
        tensor p ^ 1 rides in that launch --, the all-reduce, the optimiser."""
        self._grads_next(p, chained)
        dist.allreduce_sum_(self.grad)                               # C1+C2 fused; SUM, not mean (ddpg.py:452)
        self._adam_only()

    def _grads_next(self, p, chained=False):
        S = self.sample_transitions
        ops.ddpg_grads(self.net_cfg, self.theta, self.theta_target, self._pp[p], self._layout, self._Bt,
                       self._workspace, self.grad, self._losses, self._Q_pi,
                       o_stats=self.o_stats.state if self.normalize_obs else None,
                       g_stats=self.g_stats.state if self.normalize_obs else None, step_ctr=self._step_ctr,
                       params_unchanged=chained, next_batch=self._pp[p ^ 1], storage=self._pool.storage,
                       buf_stride=self._pool.buf_stride, tasks=S.tasks,
                       params=S.params(self.clip_obs, self.relative_goals), rng=self._rng_desc)

    def _adam_only(self):
        ops.adam_update(self.theta, self._m, self._v, self.grad, self.off_pi, self.P_total - self.off_pi,
                        alpha_tab=self._alpha_tab, step_ctr=self._step_ctr, tab_base=self._alpha_base,
                        keep=self._kept_copies())

    def _train_device_ranks(self, k=1):
        """k = 1: one update from the staging tensor of the current parity (which flips: the gradient launch draws the
        next batch into the other tensor); k = CHAIN or MAX_CHAIN (even, parity 0): one graph of k updates with the
        collective captured inside."""
        one_graph = self.use_graph and self._graph_allreduce()
        p = self._cur
        if self.use_graph and not one_graph:
            self._rank_graphs()
        if one_graph and k == 1 and self._graphs[p] is None:
            self._graphs[p] = self._capture(lambda: self._ranks_update(p))
            self._batch_stale = True
        if one_graph and k > 1 and k not in (self._graph_chain or {}):
            assert p == 0 and k % 2 == 0
            self._graph_chain = dict(self._graph_chain or {})
            self._graph_chain[k] = self._capture(lambda: [self._ranks_update(i & 1, i > 0) for i in range(k)])
            self._batch_stale = True
        if self._batch_stale:
            self._sample_packed()
            self._batch_stale = False
        if self.Q_adam.t % 100 == 0:
            self._check_synced()
        if one_graph:
            (self._graph_chain[k] if k > 1 else self._graphs[p]).replay()
        elif self.use_graph:
            self._graph_a[p].replay()
            dist.allreduce_sum_(self.grad)
            self._graph_b.replay()
        else:
            self._ranks_update(p)
        self._cur ^= (k & 1)
        self._staged = self._pp[self._cur]
        self.Q_adam.t += k
        self.pi_adam.t += k
        self._keep_alpha_ahead()
        return self._losses[0], self._Q_pi
