"""DDPG + HER agent on the GPU.  Mirrors DDPG baselines/her/ddpg.py:18-537 (constructor kwargs, methods, attributes).

What replaces what (details in DESIGN.md):
  TF graph + session            -> curious_ddpg_grads / curious_policy_forward (fp32 MFMA kernels), no session
  StagingArea feed (stage_op)   -> the packed batch tensor written by curious_her_sample
  per-buffer sample + concat + shuffle + clip (ddpg.py:326-353) -> ONE curious_her_sample launch
  2 x MpiAdam (Allreduce + NumPy Adam + getflat/setfromflat) -> one RCCL all-reduce + one fused Adam kernel
  18 tf.assign ops for Polyak   -> curious_polyak_update
Reference quirks are kept on purpose: gradients summed over ranks, train() returns main.Q_pi as
"actor_loss", buffer 0 is never written, only tasks j < 5 are routed when nb_tasks >= 5, buffers 6.. alias buffer 5.

rng_mode='numpy' (default): every random draw comes from the NumPy global stream in the reference's order (parity
with a seeded reference run).  rng_mode='device': Philox streams on the GPU, nothing crosses PCIe per update and the
whole train() step can be replayed from a hipGraph (use_graph=True).

The class is assembled from one module per concern (the reference's ddpg.py is one 537-line class as well; this one
had grown to 1 700):
  curious_amd.acting            get_actions, the fused act + env-step / whole-rollout launches, the evaluator's Q values
  curious_amd.storing           store_episode: routing on the host or on the device, the normaliser update
  curious_amd.sampling          sample_batch / stage_batch: proportions, device tables, the packed staging tensor
  curious_amd.update_schedules  train_batches on one rank: fused updates, chained hipGraphs, the step-size ring, replay
  curious_amd.rank_schedules    the update on several ranks: split / captured / pipelined / IPC schedules, check_synced
  curious_amd.faults            the guard of the in-kernel Q' hand-off (HandoffFault)
  curious_amd.persistence       logs, weight files
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from curious_amd import dist, ops
from curious_amd.acting import ActingMixin
from curious_amd.dist import RANK_SEED_STRIDE                        # noqa: F401  (re-exported: tests, experts)
from curious_amd.faults import FAULT_CHECK_EVERY, FaultsMixin, HandoffFault       # noqa: F401
from curious_amd.mpi_adam import MpiAdam
from curious_amd.normalizer import Normalizer
from curious_amd.persistence import PersistenceMixin
from curious_amd.rank_schedules import RankSchedulesMixin
from curious_amd.sampling import SamplingMixin
from curious_amd.storing import StoringMixin
from curious_amd.update_schedules import (ALPHA_TAB, CAPTURE_MODE, CHAIN, LONG_CHAIN, MAX_CHAIN,     # noqa: F401
                                          MAX_CHAIN_GRAPHS, UpdateSchedulesMixin)
from curious_amd.util import import_function, store_args


def dims_to_shapes(input_dims):
    return {key: tuple([val]) if val > 0 else tuple() for key, val in input_dims.items()}


class DDPG(ActingMixin, StoringMixin, SamplingMixin, UpdateSchedulesMixin, RankSchedulesMixin, FaultsMixin,
           PersistenceMixin):
    @store_args
    def __init__(self, input_dims, hidden, layers, network_class, polyak, batch_size,
                 Q_lr, pi_lr, norm_eps, norm_clip, max_u, action_l2, clip_obs, scope, T,
                 rollout_batch_size, subtract_goals, relative_goals, clip_pos_returns, clip_return,
                 normalize_obs, sample_transitions, gamma, buffers=None, reuse=False, tasks_ag_id=None,
                 tasks_g_id=None, task_replay='', t_id=None, eps_task=None, structure='curious',
                 rng_mode='numpy', seed=0, use_graph=False, async_store=False, virtual_ranks=1, rank_base=None,
                 total_ranks=None, **kwargs):
        """Same arguments as the reference (ddpg.py:20-59) plus rng_mode / seed / use_graph / async_store / virtual_ranks.

        virtual_ranks = V > 1: this process stands for V of the reference's MPI ranks (readme.md:16: the published runs use
        19; train.py:272-281).  Every virtual rank keeps what a rank keeps for itself -- its replay buffers
        (config.py:210-214), its RNG streams (train.py:242-243), its rollouts -- and ONE update consumes V minibatches of
        `batch_size` transitions, one per virtual rank, in one launch sequence: the losses are means per virtual rank and
        the gradient is their SUM, exactly what MpiAdam.update's Allreduce(SUM) makes of V processes (mpi_adam.py:26-28);
        the normaliser sums are averaged over the V x world ranks (normalizer.py:84-94).  Global rank of virtual rank v
        = dist.rank() * V + v: its streams are the ones that real rank would have."""
        if self.clip_return is None:
            self.clip_return = np.inf
        self.V = self.virtual_ranks = int(virtual_ranks or 1)
        # global rank of this process's first virtual rank / ranks of the whole job: V per process unless the job was laid
        # out unevenly (dist.virtual_layout: --num_cpu 19 on 8 processes)
        self.rank_base = int(rank_base) if rank_base is not None else dist.rank() * self.V
        self.total_ranks = int(total_ranks) if total_ranks is not None else dist.world_size() * self.V
        if self.V > 1 and not (rng_mode == 'device' and structure in ('curious', 'task_experts')):
            raise ValueError("virtual_ranks > 1 needs rng_mode='device' and structure 'curious' or 'task_experts'")
        self._Bt = self.V * int(batch_size)                          # rows of one update's joint batch
        # e.g. info, use_mpi: stored by store_args, pickled with the policy; names with a leading underscore are
        # construction hooks of this implementation (_alloc: slab allocator of curious_amd.experts.ExpertBank)
        self._extra_kwargs = tuple(k for k in kwargs.keys() if not k.startswith('_'))
        self._alloc = kwargs.get('_alloc')
        # several ranks: 'rccl' (default) = RCCL all-reduce + stand-alone optimiser launch; 'ipc' = the fused
        # reduce-scatter + Adam + all-gather kernel over peer-mapped buffers (csrc/ipc.hip; functional, opt-in)
        self._allreduce = kwargs.get('_allreduce') or os.environ.get('CURIOUS_ALLREDUCE', 'rccl')
        assert self._allreduce in ('rccl', 'ipc'), self._allreduce
        self._scope_arg = scope
        self.create_actor_critic = import_function(self.network_class)
        self.dimo, self.dimg = self.input_dims['o'], self.input_dims['g']
        self.dimag, self.dimu = self.input_dims['ag'], self.input_dims['u']
        self.dimtd = self.input_dims['task_descr'] if structure in ('curious', 'task_experts') else 0
        self.modular = bool(getattr(self.create_actor_critic, 'modular', True))
        if not self.modular:
            self.dimtd = 0
        assert rng_mode in ('numpy', 'device')
        self.device = torch.device('cuda', torch.cuda.current_device())

        # stage order of the reference: sorted non-info keys, then o_2, g_2, r (ddpg.py:75-83)
        stage_shapes = OrderedDict()
        input_shapes = dims_to_shapes(self.input_dims)
        for key in sorted(self.input_dims.keys()):
            if key.startswith('info_'):
                continue
            stage_shapes[key] = (None, *input_shapes[key])
        for key in ['o', 'g']:
            stage_shapes[key + '_2'] = stage_shapes[key]
        stage_shapes['r'] = (None, 1)
        self.stage_shapes = stage_shapes
        if t_id is not None:
            self.scope += str(t_id)

        self._create_network(reuse=reuse)

        if structure in ('curious', 'task_experts'):
            self.nb_tasks = len(tasks_g_id)
        if buffers is not None:
            self.buffer = buffers                                    # (virtual ranks: rank 0's list; .ranks has them all)
            ranks = getattr(buffers, 'ranks', None)
            if self.V > 1 and (ranks is None or len(ranks) != self.V):
                raise ValueError('virtual_ranks = %d needs the buffers of every rank on one pool '
                                 '(replay_buffer.make_pooled_buffers(..., n_ranks=%d))' % (self.V, self.V))
            self._rank_buffers = list(ranks) if ranks is not None else [self.buffer]
            for bl in self._rank_buffers:
                if isinstance(bl, list) and len(bl) > 5:
                    for i in range(6, len(bl)):                      # distractor buffers are equal (ddpg.py:106-110)
                        bl[i] = bl[5]
            self._adopt_buffers()
        self.first = True
        self.cp = np.zeros(self.nb_tasks) if hasattr(self, 'nb_tasks') else None
        self.proportions = None
        self._staged = None
        self._pp = None                                              # the two staging tensors of the device loop
        self._cur = 0
        self._graph_a = self._graph_b = self._graph_ba = self._graph_chain = None    # several ranks: split / chained graphs
        self._chains = None                                          # single rank: {length: graph of that many updates}
        self._graphs = [None, None]
        self._tables_dirty = True
        self._batch_stale = True
        self._store_pending = None                                   # async_store: routing the host has not mirrored yet
        self._async_batch = None

    def _new(self, shape, dtype=torch.float32, name=None):
        """Zero-filled device tensor holding per-agent update state.  An ExpertBank passes an allocator that carves these
        tensors out of the agent's slab (same order and sizes for every expert -> same offsets; `name` lets it place the
        gradient vector in the bank's contiguous gradient block instead)."""
        if self._alloc is not None:
            return self._alloc(shape, dtype, name)
        if name == 'grad' and self._allreduce == 'ipc' and dist.is_distributed():
            # what the peers read and write lives in a fine-grained block of its own (curious_amd.ipc.IpcBlock: grad |
            # staging vector of the new parameters | flags); theta stays ordinary local memory (csrc/ipc.hip)
            if getattr(self, '_ipc_block', None) is None:
                from curious_amd.ipc import IpcBlock
                self._ipc_block = IpcBlock([int(np.prod(shape))] * 2)
            return self._ipc_block.tensor(0).view(*shape)
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def _create_network(self, reuse=False):
        cfg = ops.make_net_cfg(self.dimo, self.dimg, self.dimu, self.dimtd, self.hidden, self.layers, self.modular,
                               self.max_u, self.gamma, self.clip_return, self.action_l2, self.clip_pos_returns,
                               self.normalize_obs, self.norm_clip,
                               loss_rows=self.batch_size if self.V > 1 else 0)
        self.net_cfg = cfg
        self.P_Q, self.P_pi, self.off_pi, self.P_total = ops.param_layout(cfg)
        dev = self.device
        # running averages; both accumulators in one buffer -> one all-reduce per cycle
        self._stats_acc = self._new([2 * self.dimo + 1 + 2 * self.dimg + 1])
        self.o_stats = Normalizer(self.dimo, self.norm_eps, self.norm_clip, _acc=self._stats_acc[:2 * self.dimo + 1],
                                  _state=self._new([4 * self.dimo + 1]))
        self.g_stats = Normalizer(self.dimg, self.norm_eps, self.norm_clip, _acc=self._stats_acc[2 * self.dimo + 1:],
                                  _state=self._new([4 * self.dimg + 1]))
        # parameters: Xavier-uniform kernels, zero biases (util.py:81,87-88,99).  TensorFlow draws them from its own
        # generator (not NumPy's), so a private RandomState is used and the NumPy global stream is left untouched.
        wrng = np.random.RandomState(self.seed)
        flat = np.concatenate([self._xavier(self._shapes(True), wrng), self._xavier(self._shapes(False), wrng)])
        self.theta = self._new([self.P_total], name='theta')
        self.theta.copy_(torch.from_numpy(ops.pad_params(cfg, flat)))
        self.theta_target = self._new([self.P_total])
        self.grad = self._new([self.P_total], name='grad')
        self._m = self._new([self.P_total])
        self._v = self._new([self.P_total])
        self.Q_adam = MpiAdam(self.theta[:self.off_pi], scale_grad_by_procs=False)       # ddpg.py:452-453
        self.pi_adam = MpiAdam(self.theta[self.off_pi:], scale_grad_by_procs=False)
        self.Q_adam.m, self.Q_adam.v = self._m[:self.off_pi], self._v[:self.off_pi]
        self.pi_adam.m, self.pi_adam.v = self._m[self.off_pi:], self._v[self.off_pi:]
        self._workspace = self._new([ops.workspace_floats(cfg, self._Bt)])
        self._act_ws = {}
        self._losses = self._new([2 * self.V])                       # [Q_loss, pi_loss] of every virtual rank
        self._Q_pi = self._new([self._Bt, 1])
        self._step_ctr = self._new([1], torch.int64)
        self._alpha_tab = self._new([ALPHA_TAB, 2])
        self._alpha_base = 0
        self._alpha_filled = 0
        self._noise_counter = 0
        self._sync_optimizers()                                      # ddpg.py:466
        self._init_target_net()                                      # ddpg.py:467

    def _shapes(self, critic):
        S = self.dimo + (self.dimtd if self.modular else self.dimg) + (self.dimu if critic else 0)
        out = self.dimu if not critic else 1
        shapes = [(S, self.hidden), (self.hidden,)]
        if self.modular:
            shapes.append((self.dimg, self.hidden))
        for _ in range(self.layers - 1):
            shapes += [(self.hidden, self.hidden), (self.hidden,)]
        shapes += [(self.hidden, out), (out,)]
        return shapes

    @staticmethod
    def _xavier(shapes, rng):
        parts = []
        for s in shapes:
            if len(s) == 2:
                lim = np.sqrt(6.0 / (s[0] + s[1]))
                parts.append(rng.uniform(-lim, lim, size=s).astype(np.float32).reshape(-1))
            else:
                parts.append(np.zeros(s, np.float32))
        return np.concatenate(parts)

    def _grank0(self):
        """Global rank of this process's first (virtual) rank: rank r of a job with V virtual ranks per process stands for
        the reference's ranks r V .. r V + V - 1 (train.py:242-243: every rank has its own seed); an uneven layout
        (dist.virtual_layout) hands the base in."""
        return self.rank_base

    def _adopt_buffers(self):
        """All per-task buffers must share one pool so that a mixed minibatch is a single gather launch."""
        bufs = [b for bl in self._rank_buffers for b in (bl if isinstance(bl, list) else [bl])]
        real = [b for b in bufs if b is not None]
        pool = real[0].pool
        for b in real:
            if b.pool is not pool:
                raise ValueError('the replay buffers of one agent must share a ReplayPool '
                                 '(use curious_amd.replay_buffer.make_pooled_buffers / config.configure_buffer)')
        self._pool = pool
        self._layout = real[0].layout

    def _sync_optimizers(self):
        dist.broadcast_(self.theta, 0)                               # C3: one broadcast for both networks

    def _grads(self, chained=False):
        """chained: the launch in front of this one on the stream was an optimiser call that keeps the transposed weight
        copies of the workspace current (_adam_only) inside the same captured graph -- see _update_fused."""
        b = self._staged
        ops.ddpg_grads(self.net_cfg, self.theta, self.theta_target, b, self._layout_for_batch, self._Bt,
                       self._workspace, self.grad, self._losses, self._Q_pi,
                       o_stats=self.o_stats.state if self.normalize_obs else None,
                       g_stats=self.g_stats.state if self.normalize_obs else None, step_ctr=self._step_ctr,
                       params_unchanged=chained)
        return self._losses[0], self._Q_pi, self.grad[:self.P_Q], self.grad[self.off_pi:self.off_pi + self.P_pi]

    def _update(self, Q_grad=None, pi_grad=None, use_table=False):
        """Both MpiAdam.update calls of ddpg.py:246-248 as one all-reduce + one kernel over [theta_Q | theta_pi]."""
        if self.Q_adam.t % 100 == 0:
            self._check_synced()                                     # C4
        dist.allreduce_sum_(self.grad)                               # C1+C2 fused; SUM, not mean (ddpg.py:452)
        self.Q_adam.t += 1
        self.pi_adam.t += 1
        if use_table:
            ops.adam_update(self.theta, self._m, self._v, self.grad, self.off_pi, self.P_total - self.off_pi,
                            alpha_tab=self._alpha_tab, step_ctr=self._step_ctr, tab_base=self._alpha_base,
                            keep=self._fault_guard())
        else:
            ops.adam_update(self.theta, self._m, self._v, self.grad, self.off_pi, self.P_total - self.off_pi,
                            self.Q_adam.alpha(self.Q_lr), self.pi_adam.alpha(self.pi_lr), keep=self._fault_guard())

    def train(self, stage=True):
        """One update (ddpg.py:368-373).  Returns (critic_loss, actor_loss) as GPU tensors (no host sync);
        actor_loss is main.Q_pi like in the reference (ddpg.py:237-243)."""
        if stage and self._device_loop():
            return self._train_device(1)
        if stage:
            self.stage_batch()
        critic_loss, actor_loss, Q_grad, pi_grad = self._grads()
        self._update(Q_grad, pi_grad)
        return critic_loss, actor_loss

    def _init_target_net(self):
        ops.polyak_update(self.theta_target, self.theta, 0.0)        # ddpg.py:459-460

    def update_target_net(self):
        # the fault word travels to the host every FAULT_CHECK_EVERY-th cycle (train.py:154: once per cycle we are here):
        # a faulted update freezes the parameters until the word is cleared, so a late report loses nothing
        self._fault_tick = getattr(self, '_fault_tick', 0) + 1
        ops.polyak_update(self.theta_target, self.theta, self.polyak)   # ddpg.py:461-462
        # verdict of an EARLIER copy, raised when the work of this call is done (the copy this call enqueues is looked at
        # by a later call: on an idle GPU it may arrive within microseconds, and whether this very call raised would
        # depend on timing).  One rank: as soon as the copy has arrived (no stall).  Several ranks: a fault on ANY rank
        # froze ALL of them (the flag element of the gradient all-reduce), and all of them read their verdict at the SAME
        # cycle count, FAULT_CHECK_EVERY - 1 cycles after the copy was enqueued (it arrived long ago) -- every rank raises,
        # clears and resumes in the same cycle, the replicas stay identical
        try:
            if not dist.is_distributed():
                self.check_faults(wait=False)
            elif self._fault_tick % FAULT_CHECK_EVERY == 0:
                self._fault_verdict(wait=True)
        finally:
            if self._fault_tick % FAULT_CHECK_EVERY == 1:
                self._enqueue_fault_check()

    def __getstate__(self):
        """Policies can be reloaded from a pickle for acting; training cannot be resumed from it (ddpg.py:511-521).
        The state is the constructor's own arguments (the reference filters __dict__ by substrings, which here would
        also drop e.g. use_graph) plus the weights in save_weights order."""
        if '_snapshot_state' in self.__dict__:                       # curious_amd.util.PolicySnapshot of this policy
            return self.__dict__['_snapshot_state']
        import inspect
        names = [n for n in inspect.signature(DDPG.__init__).parameters if n not in ('self', 'kwargs')]
        names += list(getattr(self, '_extra_kwargs', ()))
        skip = ('buffers', 'sample_transitions')                     # ddpg.py:514-516: no buffers, no sampler
        state = {k: self.__dict__[k] for k in names if k in self.__dict__ and k not in skip}
        state['scope'] = self._scope_arg
        state['weights'] = self._weights_lists()
        return state

    def __setstate__(self, state):
        weights = state.pop('weights')
        if 'sample_transitions' not in state:
            state['sample_transitions'] = None
        self.__init__(**state)
        self._set_weights_lists(weights)
