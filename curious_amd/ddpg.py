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
"""
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

from curious_amd import _lib, dist, ops
from curious_amd.her import TransitionBatch, upload_plan
from curious_amd.mpi_adam import MpiAdam
from curious_amd.normalizer import Normalizer, recompute_many
from curious_amd.replay_buffer import EpisodeViews, ReplayBuffer, as_records
from curious_amd.util import import_function, store_args, transitions_in_episode_batch

ALPHA_TAB = 4096        # Adam step sizes precomputed per cycle for graph replay
FAULT_CHECK_EVERY = 8   # cycles between asynchronous reads of the hand-off fault word (DDPG.update_target_net)
CHAIN = 10              # updates per chained hipGraph launch in train_batches (even: the staging tensors alternate)
LONG_CHAIN = 50         # batched experts: a longer chain when that many updates are due
MAX_CHAIN = 100         # single-rank path: train_batches(n) replays ONE graph of min(n, 100) (even) updates -- every graph
                        # launch leaves the GPU idle for ~5 us and its first update rebuilds the transposed weight copies
                        # (3.6 us); the reference's n_batches = 40 is one launch instead of four
MAX_CHAIN_GRAPHS = 4    # distinct chain lengths kept captured; further lengths fall back to chains of CHAIN
# hipStreamCaptureModeThreadLocal: HIP calls of OTHER threads (the RCCL watchdog polling events) must not invalidate a
# capture that only this thread's launches take part in
CAPTURE_MODE = 'thread_local'
RANK_SEED_STRIDE = 1000003   # what separates the device RNG keys of consecutive (global) ranks


class HandoffFault(_lib.CuriousHipError):
    """A consumer of the in-kernel Q' hand-off of the row-local update gave up (include/curious_hip.h,
    curious_workspace_fault_offset).  The optimiser skipped every update since: parameters, moments and target are those
    of the last good update."""


def dims_to_shapes(input_dims):
    return {key: tuple([val]) if val > 0 else tuple() for key, val in input_dims.items()}


class DDPG(object):
    @store_args
    def __init__(self, input_dims, hidden, layers, network_class, polyak, batch_size,
                 Q_lr, pi_lr, norm_eps, norm_clip, max_u, action_l2, clip_obs, scope, T,
                 rollout_batch_size, subtract_goals, relative_goals, clip_pos_returns, clip_return,
                 normalize_obs, sample_transitions, gamma, buffers=None, reuse=False, tasks_ag_id=None,
                 tasks_g_id=None, task_replay='', t_id=None, eps_task=None, structure='curious',
                 rng_mode='numpy', seed=0, use_graph=False, async_store=False, virtual_ranks=1, **kwargs):
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
        if self.V > 1 and not (rng_mode == 'device' and structure == 'curious'):
            raise ValueError("virtual_ranks > 1 needs rng_mode='device' and structure='curious'")
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

    # ------------------------------------------------------------------ construction
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
        the reference's ranks r V .. r V + V - 1 (train.py:242-243: every rank has its own seed)."""
        return dist.rank() * self.V

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

    # ------------------------------------------------------------------ acting
    def _random_action(self, n):
        return np.random.uniform(low=-self.max_u, high=self.max_u, size=(n, self.dimu))   # ddpg.py:114-115

    def get_actions(self, o, ag, g, task_descr=None, noise_eps=0., random_eps=0., use_target_net=False,
                    compute_Q=False):
        """ddpg.py:129-161.  NumPy inputs -> NumPy outputs (host envs); GPU tensors -> GPU tensors (batched env)."""
        host_io = not isinstance(o, torch.Tensor)
        dev = self.device
        if host_io:
            # host envs (rollout.py:226-232 with a Python list of envs): [o | ag | g | td] of all envs goes up in ONE
            # asynchronous copy from a pinned block, the actions (and Q) come back through a pinned block; two blocks
            # alternate so that the caller may fill the next step's inputs while this step's copies are in flight
            n = int(np.asarray(o).reshape(-1, self.dimo).shape[0])
            io = self._host_io_blocks(n)
            hin = io['hin'][io['k']]
            view = hin.numpy()
            c0, c1, c2 = self.dimo, self.dimo + self.dimag, self.dimo + self.dimag + self.dimg
            view[:, :c0] = np.asarray(o, dtype=np.float32).reshape(n, self.dimo)
            view[:, c0:c1] = np.asarray(ag, dtype=np.float32).reshape(n, self.dimag)
            view[:, c1:c2] = np.asarray(g, dtype=np.float32).reshape(n, self.dimg)
            if self.dimtd > 0:
                view[:, c2:c2 + self.dimtd] = np.asarray(task_descr, dtype=np.float32).reshape(n, self.dimtd)
            din = io['din'][io['k']]
            din.copy_(hin, non_blocking=True)
            o_d, ag_d, g_d = din[:, :c0], din[:, c0:c1], din[:, c1:c2]
            td_d = din[:, c2:c2 + self.dimtd] if self.dimtd > 0 else None
        else:
            o_d, g_d, ag_d = o.reshape(-1, self.dimo), g.reshape(-1, self.dimg), ag.reshape(-1, self.dimag)
            td_d = task_descr.reshape(-1, self.dimtd) if self.dimtd > 0 else None
            n = o_d.shape[0]
        theta = self.theta_target if use_target_net else self.theta
        ws = self._act_ws.get(n)
        if ws is None:
            ws = torch.zeros(ops.workspace_floats(self.net_cfg, n), dtype=torch.float32, device=dev)
            self._act_ws[n] = ws
        u = torch.empty([n, self.dimu], dtype=torch.float32, device=dev)
        Q = torch.empty([n, 1], dtype=torch.float32, device=dev) if compute_Q else None
        ops.policy_forward(self.net_cfg, theta, o_d, g_d, td_d, n, self.clip_obs, ws, u, Q, ag=ag_d,
                           relative_goals=self.relative_goals,
                           o_stats=self.o_stats.state if self.normalize_obs else None,
                           g_stats=self.g_stats.state if self.normalize_obs else None)
        noise_scale = noise_eps * self.max_u
        if self.rng_mode == 'numpy':
            # RNG draws happen even when the eps are 0 (stream consumption matters for seed parity)
            randn = np.random.randn(n, self.dimu)                    # ddpg.py:149
            binom = np.random.binomial(1, random_eps, n).astype(np.float64)   # ddpg.py:152
            unif = self._random_action(n)
            host = np.concatenate([randn.reshape(-1), binom, unif.reshape(-1)])
            d = torch.from_numpy(host).to(dev)
            k = n * self.dimu
            ops.action_noise(u, n, self.dimu, noise_scale, random_eps, self.max_u, d[:k], d[k:k + n], d[k + n:])
        else:
            self._noise_counter += 1
            ops.action_noise(u, n, self.dimu, noise_scale, random_eps, self.max_u,
                             seed=self.seed * 2654435761 + 12345 + self._grank0() * RANK_SEED_STRIDE, counter=self._noise_counter)
        if host_io:
            hout = io['hout'][io['k']]
            hout[:, :self.dimu].copy_(u, non_blocking=True)
            if compute_Q:
                hout[:, self.dimu:].copy_(Q, non_blocking=True)
            io['done'].record()
            io['k'] ^= 1
            io['done'].synchronize()                                 # the only host wait of an acting step
            out = hout.numpy()
            u_h = out[:, :self.dimu].copy()
            if u_h.shape[0] == 1:
                u_h = u_h[0]
            return [u_h, out[:, self.dimu:].copy()] if compute_Q else u_h
        return [u, Q] if compute_Q else u

    def _host_io_blocks(self, n):
        io = getattr(self, '_host_io', {}).get(n)
        if io is None:
            w = self.dimo + self.dimag + self.dimg + max(self.dimtd, 0)
            w = (w + 3) & ~3                                         # rows stay 16-byte aligned (lean layer-0 loads)
            io = dict(hin=[torch.zeros([n, w], dtype=torch.float32).pin_memory() for _ in range(2)],
                      din=[torch.zeros([n, w], dtype=torch.float32, device=self.device) for _ in range(2)],
                      hout=[torch.zeros([n, self.dimu + 1], dtype=torch.float32).pin_memory() for _ in range(2)],
                      done=torch.cuda.Event(), k=0)
            if not hasattr(self, '_host_io'):
                self._host_io = {}
            self._host_io[n] = io
        return io

    def can_act_and_step(self, env, compute_Q):
        """The fused acting step applies to the GPU-resident synthetic env in throughput mode.  compute_Q (the
        evaluator, train.py:308-319): the fused kernels record no Q -- the rollout's Q values are computed afterwards
        from its recorded rows (rollout_q_sum), which needs the whole rollout as one launch (act_rollout)."""
        return (self.rng_mode == 'device' and self.modular
                and self.dimu == 4 and hasattr(env, 'step_all')
                and getattr(env, 'dimo', None) == self.dimo and getattr(env, 'nb_tasks', None) == self.dimtd)

    Q_ROWS = 4096                 # rows per launch of rollout_q_sum (bounds its workspace: ~110 MB)

    def rollout_q_sum(self, env, T, use_target_net=False):
        """sum over the T steps of the batch-mean Q of the rollout that was just enqueued for `env` (a GPU scalar) -- what
        RolloutWorker accumulates step by step from get_actions(compute_Q=True) (rollout.py:187-189,226-232; ddpg.py:140-146:
        Q_pi_tf = Q(o_t, g, pi(o_t, g))), computed AFTER the fused rollout from its recorded rows: record row t of an episode
        holds the observation, goal and task descriptor the policy saw at step t, so one actor + critic forward over the
        [n x (T + 1)] rows of the staging block (a few launches of Q_ROWS rows) yields the same Q values bit for bit;
        their mean is taken over [n_used, T] in one reduction instead of T batch means -- a different order of summation:
        equal within float32 rounding (~1e-6 relative), not bit for bit."""
        n, lay = env.n, env.layout
        rows = env.staging.view(n * (T + 1), lay.row_stride)
        theta = self.theta_target if use_target_net else self.theta
        if getattr(self, '_q_rows', None) is None or self._q_rows[0].numel() != rows.shape[0]:
            chunk = min(self.Q_ROWS, rows.shape[0])
            self._q_rows = (torch.empty(rows.shape[0], dtype=torch.float32, device=self.device),
                            torch.empty([chunk, self.dimu], dtype=torch.float32, device=self.device),
                            torch.zeros(ops.workspace_floats(self.net_cfg, chunk), dtype=torch.float32, device=self.device))
        q, u, ws = self._q_rows
        o_, g_, ag_, td_ = lay.off['o'], lay.off['g'], lay.off['ag'], lay.off['task_descr']
        for r0 in range(0, rows.shape[0], u.shape[0]):
            blk = rows[r0:r0 + u.shape[0]]
            m = blk.shape[0]
            ops.policy_forward(self.net_cfg, theta, blk[:, o_:o_ + self.dimo], blk[:, g_:g_ + self.dimg],
                               blk[:, td_:td_ + self.dimtd] if self.dimtd > 0 else None, m, self.clip_obs, ws, u[:m],
                               q[r0:r0 + m].view(m, 1), ag=blk[:, ag_:ag_ + self.dimag],
                               relative_goals=self.relative_goals,
                               o_stats=self.o_stats.state if self.normalize_obs else None,
                               g_stats=self.g_stats.state if self.normalize_obs else None)
        n_used = getattr(env, 'n_used', n)                           # (idle padding envs and the rows t = T do not count)
        return q.view(n, T + 1)[:n_used, :T].mean() * T

    def act_and_step(self, env, t, noise_eps=0., random_eps=0., use_target_net=False):
        """policy.get_actions(...) + env.step(...) for every env of a BatchedSyntheticArm in one launch
        (curious_policy_act_env_step); same numbers as get_actions followed by env.step_all."""
        n = env.n
        theta = self.theta_target if use_target_net else self.theta
        ws = self._act_ws.get(n)
        if ws is None:
            ws = torch.zeros(ops.workspace_floats(self.net_cfg, n), dtype=torch.float32, device=self.device)
            self._act_ws[n] = ws
        if getattr(self, '_act_u', None) is None or self._act_u.shape[0] != n:
            self._act_u = torch.empty([n, self.dimu], dtype=torch.float32, device=self.device)
        self._noise_counter += 1
        from curious_amd.envs import REWARD_EPS
        ops.policy_act_env_step(self.net_cfg, theta, n, self.clip_obs, ws, noise_eps * self.max_u, random_eps,
                                self.seed * 2654435761 + 12345 + self._grank0() * RANK_SEED_STRIDE, self._noise_counter,
                                self._act_u, env._cfg, env.layout, env.env_id0, env.episode, env.tasks, t, env.o,
                                env.ag, env.g, env.td, env.staging, REWARD_EPS, flags=getattr(env, 'flags', None),
                                o_stats=self.o_stats.state if self.normalize_obs else None,
                                g_stats=self.g_stats.state if self.normalize_obs else None,
                                relative_goals=self.relative_goals)
        return self._act_u

    def act_rollout(self, env, T, noise_eps=0., random_eps=0., use_target_net=False, exploit=None):
        """The T-step acting loop of a batched rollout (rollout.py:226-303 for every env): T x act_and_step, as ONE
        launch (curious_policy_rollout) where the row-local route applies.  With use_graph the launches are captured once
        per (env, noise setting) and replayed; the Philox noise counter is (t + 1) + a device-resident base that advances
        by T per rollout, so replays draw fresh noise and the eager loop draws the same numbers.
        exploit (virtual ranks): one flag per virtual rank -- the envs of a rank that exploits act without exploration
        noise in this rollout (rollout.py:183-189); the envs are V consecutive groups, each drawing its noise from the
        stream of its own global rank."""
        from curious_amd.envs import REWARD_EPS
        n = env.n
        theta = self.theta_target if use_target_net else self.theta
        ws = self._act_ws.get(n)
        if ws is None:
            ws = torch.zeros(ops.workspace_floats(self.net_cfg, n), dtype=torch.float32, device=self.device)
            self._act_ws[n] = ws
        if getattr(self, '_act_u', None) is None or self._act_u.shape[0] != n:
            self._act_u = torch.empty([n, self.dimu], dtype=torch.float32, device=self.device)
        if getattr(self, '_noise_base', None) is None:
            self._noise_base = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._noise_base_val = 0
            self._roll_graphs = {}
        # ONE logical noise counter for every acting path: the host value `_noise_counter` (get_actions / act_and_step
        # pass it as a kernel argument) and its device mirror `_noise_base` (read by the captured launches below).  The
        # mirror is brought up to date here when host-side acting calls ran since the last rollout, so no two acting
        # calls of one agent ever draw from the same (seed, counter) pair.
        if self._noise_base_val != self._noise_counter:
            self._noise_base.fill_(self._noise_counter)
            self._noise_base_val = self._noise_counter
        seed = self.seed * 2654435761 + 12345 + self._grank0() * RANK_SEED_STRIDE     # same stream as get_actions / act_and_step
        u_out = self._act_u
        groups = None
        if self.V > 1:
            group = getattr(env, 'n_used', n) // self.V            # envs per virtual rank (padding envs: groups >= V)
            ng = (n + group - 1) // group
            if getattr(self, '_exploit_dev', None) is None or self._exploit_dev.numel() != ng:
                self._exploit_dev = torch.zeros(ng, dtype=torch.int32, device=self.device)
                self._exploit_pins = [torch.zeros(ng, dtype=torch.int32).pin_memory() for _ in range(4)]
                self._exploit_k = 0
            pin = self._exploit_pins[self._exploit_k]               # (a small ring: the copy is asynchronous)
            self._exploit_k = (self._exploit_k + 1) % len(self._exploit_pins)
            pin.zero_()
            if exploit is not None:
                pin[:self.V] = torch.from_numpy(np.asarray(exploit, dtype=np.int32))
            self._exploit_dev.copy_(pin, non_blocking=True)
            groups = ops.rank_groups(group, RANK_SEED_STRIDE, self._exploit_dev)

        reset_here = bool(getattr(env, '_reset_pending', False))     # the worker only uploaded the draws (reset_all)
        env._reset_pending = False

        def steps():
            # T x policy_act_env_step (noise counters 1 .. T on top of the base): one launch on the row-local route.
            # The env reset that heads the rollout also advances the noise base (one launch less per cycle): the rollout
            # then starts from base + 1 - T
            if reset_here:
                env.launch_reset(counter=self._noise_base, delta=T)
            ops.policy_rollout(self.net_cfg, theta, n, self.clip_obs, ws, noise_eps * self.max_u, random_eps, seed,
                               (1 - T) if reset_here else 1,
                               u_out, env._cfg, env.layout, env.env_id0, env.episode, env.tasks, 0, T, env.o, env.ag,
                               env.g, env.td, env.staging, REWARD_EPS, counter_base=self._noise_base,
                               flags=getattr(env, 'flags', None),
                               o_stats=self.o_stats.state if self.normalize_obs else None,
                               g_stats=self.g_stats.state if self.normalize_obs else None,
                               relative_goals=self.relative_goals, groups=groups)
            if not reset_here:
                ops.counter_add(self._noise_base, T)

        self._noise_counter += T
        self._noise_base_val = self._noise_counter                   # steps() ends with the device-side += T
        if not self.use_graph:
            steps()
            return
        key = (id(env), T, float(noise_eps), float(random_eps), bool(use_target_net), reset_here)
        g = self._roll_graphs.get(key)
        if g is None:
            # capture only records (the side-stream warm-up torch recommends is skipped on purpose: it would step the
            # envs for real); the library's kernels need no lazy initialisation
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                steps()
            self._roll_graphs[key] = g
        g.replay()

    def rewind_rollout(self, env, T):
        """Undo the bookkeeping of the act_rollout that was just enqueued for `env` so that the SAME rollout can be
        generated again (same episode numbers -> same initial states, same noise counters -> same exploration noise):
        used when the weights-resident launch reported itself void (envs.ResidentRolloutVoid) and the rollout is redone
        on the streaming kernel, which computes the same numbers."""
        self._noise_counter -= T
        self._noise_base_val = self._noise_counter
        self._noise_base.fill_(self._noise_counter)
        env.episode.sub_(1)                                          # the reset advanced every env's episode counter

    def drop_rollout_graphs(self):
        """Forget the captured rollout launches (the route of curious_policy_rollout is chosen when it is captured)."""
        if getattr(self, '_roll_graphs', None):
            self._roll_graphs = {}

    def can_eval_rollout(self, env, noise_eps, random_eps):
        """Noise-free rollouts (evaluator, exploit) of the GPU-resident env can be replayed from one hipGraph: with both
        eps at 0 the result does not depend on the noise counter, so nothing host-side changes between replays."""
        return (self.rng_mode == 'device' and self.use_graph and noise_eps == 0 and random_eps == 0
                and hasattr(env, 'step_all'))

    def eval_rollout(self, env, T, use_target_net=False, compute_Q=False):
        """T x [get_actions(noise 0) -> env.step_all (-> mean Q)] (rollout.py:226-263 for every env), the launches of
        the unfused acting path captured once per (env, settings) and replayed.  Returns the sum over steps of the
        batch-mean Q (a GPU scalar) when compute_Q, else None."""
        n = env.n
        theta = self.theta_target if use_target_net else self.theta
        if getattr(self, '_roll_graphs', None) is None:
            self._roll_graphs = {}
        key = ('eval', id(env), T, bool(use_target_net), bool(compute_Q))
        entry = self._roll_graphs.get(key)
        if entry is None:
            ws = self._act_ws.get(n)
            if ws is None:
                ws = torch.zeros(ops.workspace_floats(self.net_cfg, n), dtype=torch.float32, device=self.device)
                self._act_ws[n] = ws
            u = torch.empty([n, self.dimu], dtype=torch.float32, device=self.device)
            Q = torch.empty([n, 1], dtype=torch.float32, device=self.device) if compute_Q else None
            q_acc = torch.zeros((), dtype=torch.float32, device=self.device)
            seed = self.seed * 2654435761 + 12345 + self._grank0() * RANK_SEED_STRIDE

            def steps():
                q_acc.zero_()
                for t in range(T):
                    ops.policy_forward(self.net_cfg, theta, env.o, env.g, env.td if self.dimtd > 0 else None, n,
                                       self.clip_obs, ws, u, Q, ag=env.ag, relative_goals=self.relative_goals,
                                       o_stats=self.o_stats.state if self.normalize_obs else None,
                                       g_stats=self.g_stats.state if self.normalize_obs else None)
                    ops.action_noise(u, n, self.dimu, 0.0, 0.0, self.max_u, seed=seed, counter=0)   # the clip only
                    env.step_all(u, t)
                    if compute_Q:
                        q_acc.add_(Q[:getattr(env, 'n_used', n)].mean())      # (idle padding envs do not count)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                steps()
            entry = (g, q_acc, u, Q)
            self._roll_graphs[key] = entry
        entry[0].replay()                                            # noise-free: the noise counter does not move
        return entry[1] if compute_Q else None

    # ------------------------------------------------------------------ storing
    def store_episode(self, episode_batch, cp, n_ep, update_stats=True):
        """episode_batch: {key: [batch, T or T+1, dim]} NumPy arrays, or the EpisodeViews of a device staging
        block produced by the batched RolloutWorker (ddpg.py:163-223)."""
        self._store_episode(episode_batch, cp, n_ep, update_stats)
        # verdict of an earlier fault-word copy that has arrived (no stall).  Raised AFTER the episodes are stored: a
        # caller that catches HandoffFault and goes on has lost nothing of this call
        self.check_faults(wait=False)

    def _store_episode(self, episode_batch, cp, n_ep, update_stats=True):
        self.settle()
        self.cp = cp
        self.n_episodes = n_ep
        layout = self._layout
        staging = as_records(episode_batch, layout)
        batch_size = staging.shape[0]
        marked, self._async_batch = self._async_batch, None
        if marked is not None and marked[:2] == (staging.data_ptr(), batch_size) and update_stats:
            return self._store_episode_async(staging, batch_size, marked[2], marked[3])
        if self.structure in ('curious', 'task_experts'):
            if 'buffer' in self.task_replay or self.task_replay == 'hand_designed':
                na = batch_size * self.nb_tasks
                if getattr(self, '_route_bufs', None) is None or self._route_bufs[0].numel() < na:
                    # device + pinned host mirrors for the activity flags and the (src, dst) routing pairs
                    self._route_bufs = (torch.empty(na, dtype=torch.int32, device=self.device),
                                        torch.empty(na, dtype=torch.int32).pin_memory(),
                                        torch.empty(na, dtype=torch.int32).pin_memory(),
                                        torch.empty(na, dtype=torch.int64).pin_memory(),
                                        torch.empty(na, dtype=torch.int32, device=self.device),
                                        torch.empty(na, dtype=torch.int64, device=self.device))
                active_dev, active_host = self._route_bufs[0][:na], self._route_bufs[1][:na]
                pre = getattr(self, '_activity_prefetched', None)
                self._activity_prefetched = None
                have = pre == (staging.data_ptr(), batch_size)       # RolloutWorker already fetched it with its flags
                if not have:
                    ops.episode_activity(staging, layout, self.sample_transitions.tasks, batch_size, active_dev)
                    active_host.copy_(active_dev, non_blocking=True)
                    arrived = torch.cuda.Event()
                    arrived.record()
                if update_stats and self.rng_mode == 'device':
                    # the normaliser update needs no host decision and (in this mode) no NumPy draw: enqueue it now, so
                    # that the GPU works while the host routes the episodes
                    self._update_stats(staging, batch_size)
                    update_stats = False
                if not have:
                    arrived.synchronize()                            # one D2H sync per cycle
                active = active_host.numpy().reshape(batch_size, self.nb_tasks)
                per_buffer = {}
                fast_src, fast_dst = [], []
                routed = active.astype(bool)
                if self.nb_tasks >= 5:
                    routed[:, 5:] = False                            # only tasks j < 5 are routed (ddpg.py:183)
                counts = routed.sum(axis=0)
                fits = all(self.buffer[j + 1].current_size + int(counts[j]) <= self.buffer[j + 1].size
                           for j in range(self.nb_tasks) if counts[j])
                distinct = len({id(self.buffer[j + 1]) for j in range(self.nb_tasks) if counts[j]}) == \
                    int((counts > 0).sum())
                if self.V > 1 and not distinct:
                    raise NotImplementedError('virtual ranks: the routed tasks need a buffer each')
                if self.rng_mode == 'device' and distinct:
                    # device RNG mode: the rule of replay_buffer.py:90-109 per episode -- consecutive slots while the
                    # buffer has room, then a random slot -- with the random slots drawn from the Philox stream the
                    # device-routed form uses (curious_route_store_episodes), so both forms store the same thing.
                    # Virtual ranks: rank v's episodes (rows v * per ..) into rank v's buffers, with rank v's key and
                    # ITS episode numbers -- what a process of its own would do
                    call = self._next_store_call()
                    per = batch_size // self.V
                    for v in range(self.V):
                        bufs = self._rank_buffers[v]
                        rows = routed[v * per:(v + 1) * per]
                        for j in range(self.nb_tasks):
                            eps = np.nonzero(rows[:, j])[0]
                            if not eps.size:
                                continue
                            buf = bufs[j + 1]
                            free = max(0, buf.size - buf.current_size)
                            slots = np.arange(buf.current_size, buf.current_size + min(eps.size, free), dtype=np.int64)
                            buf.current_size = min(buf.size, buf.current_size + eps.size)
                            buf.n_transitions_stored += eps.size * self.T
                            if eps.size > free:
                                slots = np.concatenate([slots, ops.store_slots_host(self._store_seed(v), call, j,
                                                                                    buf.size, eps[free:])])
                                # of two episodes on one slot the later one wins (sequential semantics)
                                _, first_rev = np.unique(slots[::-1], return_index=True)
                                keep = np.sort(eps.size - 1 - first_rev)
                                eps, slots = eps[keep], slots[keep]
                            fast_src.append((eps + v * per).astype(np.int32))
                            fast_dst.append(slots.astype(np.int64) + buf.pool_index * buf.pool.capacity)
                elif fits and distinct:
                    # no buffer overflows within this batch -> slots are consecutive and no random number is drawn
                    # (replay_buffer.py:94-95): same result as the per-episode loop below, without the loop
                    for j in range(self.nb_tasks):
                        if counts[j]:
                            buf = self.buffer[j + 1]
                            eps = np.nonzero(routed[:, j])[0]
                            slots = np.arange(buf.current_size, buf.current_size + eps.size)
                            buf.current_size += eps.size
                            buf.n_transitions_stored += eps.size * self.T
                            fast_src.append(eps.astype(np.int32))
                            fast_dst.append(slots.astype(np.int64) + buf.pool_index * buf.pool.capacity)
                else:
                    for b in range(batch_size):                      # ddpg.py:178-195, order of the RNG draws kept
                        for j in range(self.nb_tasks):
                            if routed[b, j]:
                                buf = self.buffer[j + 1]
                                slot = buf._get_storage_idx(1)
                                buf.n_transitions_stored += self.T
                                per_buffer.setdefault(id(buf), (buf, [], []))
                                per_buffer[id(buf)][1].append(b)
                                per_buffer[id(buf)][2].append(slot)
                # sequential semantics of the reference: when two episodes of this batch draw the same (random)
                # slot the later one wins -> keep only the last writer of every destination
                last = {}
                for buf, eps, slots in per_buffer.values():
                    for b, s in zip(eps, slots):
                        last[int(s) + buf.pool_index * buf.pool.capacity] = b
                dst, src = list(last.keys()), list(last.values())
                if fast_src:
                    src, dst = np.concatenate(fast_src), np.concatenate(fast_dst)
                if len(src):
                    k = len(src)
                    assert k <= na
                    src_h, dst_h, src_d, dst_d = (b[:k] for b in self._route_bufs[2:])
                    src_h.numpy()[:] = np.asarray(src, np.int32)
                    dst_h.numpy()[:] = np.asarray(dst, np.int64)
                    src_d.copy_(src_h, non_blocking=True)
                    dst_d.copy_(dst_h, non_blocking=True)
                    ops.store_episodes(self._pool.storage, staging, layout, src_d, dst_d)
                    self._pool.version += 1
            else:
                for b in range(batch_size):
                    slot = self.buffer._get_storage_idx(1)
                    self.buffer.n_transitions_stored += self.T
                    self.buffer.store_records(staging, [b], [slot])
        else:                                                        # flat (ddpg.py:199-204)
            for b in range(batch_size):
                slot = self.buffer._get_storage_idx(1)
                self.buffer.n_transitions_stored += self.T
                self.buffer.store_records(staging, [b], [slot])
        self._tables_dirty = True

        if update_stats:                                             # ddpg.py:207-223
            self._update_stats(staging, batch_size)

    # ------------------------------------------------------------------ storing without waiting for the host (opt-in)
    def can_store_async(self, batch_size):
        """The device can route the episodes of the coming rollout itself (curious_route_store_episodes) and nothing the
        host would compute from the rollout's flags is needed before the updates: every routed buffer is non-empty (the
        replay proportions, ddpg.py:255-286, then depend on the competence progress only).  Single agent on its own
        buffers, device RNG (the random slots of full buffers are Philox draws in that mode).  With several ranks the
        normaliser all-reduce of the store is stream-ordered like everything else: nothing here needs the host either."""
        if not (self.async_store and self.structure == 'curious' and self._multi_buffer() and self.rng_mode == 'device'
                and isinstance(self.buffer, list)):
            return False
        self.settle()
        nr = min(self.nb_tasks, 5)
        for bl in self._rank_buffers:                                # (batch_size: the episodes of ONE rank's rollout)
            bufs = [bl[j + 1] for j in range(nr)]
            if not (len({id(b) for b in bufs}) == nr and all(b.current_size > 0 for b in bufs)):
                return False
        return batch_size <= 2048 and self.dimo + self.dimg <= 256

    def _store_seed(self, v=0):
        return (self.seed * 6700417 + 29 + (self._grank0() + v) * RANK_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF

    def _next_store_call(self):
        self._store_calls = getattr(self, '_store_calls', 0) + 1
        return self._store_calls

    def expect_async_store(self, episode_batch, skip, skip_host=None):
        """Called by the batched RolloutWorker when it returns WITHOUT having waited for the rollout's flags: the next
        store_episode of exactly this batch takes the device-routed form.  skip: the rollout's NaN word (device);
        skip_host: where its value arrives on the host (pinned, the D2H copy already enqueued: the worker's flag copy)."""
        staging = as_records(episode_batch, self._layout)
        self._async_batch = (staging.data_ptr(), staging.shape[0], skip, skip_host)

    def _store_episode_async(self, staging, batch_size, skip, skip_host=None):
        """store_episode (ddpg.py:163-223) with the routing decided on the device: same slots, same table, same stats
        as the host-routed form; the host's mirror of the buffer sizes follows in settle()."""
        layout = self._layout
        na = batch_size * self.nb_tasks
        assert getattr(self, '_activity_prefetched', None) == (staging.data_ptr(), batch_size)
        self._activity_prefetched = None
        self._update_stats(staging, batch_size, skip=skip)           # a NaN rollout feeds no statistics either
        if self._tables_stale() or getattr(self, '_tables', None) is None:
            self._refresh_device_tables()                            # from the host's (settled) sizes
        nb1 = self.nb_tasks + 1
        n0 = nb1 + 1
        if getattr(self, '_route_count', None) is None:
            self._route_count = torch.zeros(self.V, dtype=torch.int32, device=self.device)
            self._nan_pin = torch.zeros(1, dtype=torch.float32).pin_memory()
        src_d, dst_d = self._route_bufs[4][:na], self._route_bufs[5][:na]
        ops.route_store_episodes(self._pool.storage, staging, layout, self._route_bufs[0][:na], self.nb_tasks,
                                 min(self.nb_tasks, 5), batch_size // self.V, self._tables[n0 + 2 * nb1:],
                                 self._tables[n0:], self._pool.capacity, self._store_seed(),
                                 self._next_store_call(), skip, src_d, dst_d, self._route_count, n_ranks=self.V,
                                 tab_stride=4 * nb1 + 1, seed_stride=RANK_SEED_STRIDE,
                                 tasks=self.sample_transitions.tasks if getattr(self, '_activity_in_route', False) else None)
        if getattr(self, '_activity_in_route', False):               # the flags the routing launch evaluated: to the host
            self._route_bufs[1][:na].copy_(self._route_bufs[0][:na], non_blocking=True)
            self._activity_in_route = False
        if skip_host is None:
            self._nan_pin.copy_(skip, non_blocking=True)
            skip_host = self._nan_pin
        arrived = torch.cuda.Event()
        arrived.record()                                             # behind the D2H copies of the activity and rollout flags
        self._pool.version += 1
        self._tables_dirty = False
        self._tables_sizes = self._sizes_key()                       # the device table is ahead of the host's sizes
        self._batch_stale = True                                     # until settle(); the next batch is drawn from it
        self._store_pending = (self._route_bufs[1][:na], batch_size, arrived, skip_host)

    def settle(self):
        """Bring the host's mirror of the buffer sizes up to date with a device-routed store (async_store)."""
        p, self._store_pending = self._store_pending, None
        if p is None:
            return
        active_host, batch_size, arrived, skip_host = p
        arrived.synchronize()
        if float(skip_host[0]) == 0.0:                               # (a NaN rollout was dropped on the device as well)
            routed = active_host.numpy().reshape(batch_size, self.nb_tasks).astype(bool)
            if self.nb_tasks >= 5:
                routed[:, 5:] = False                                # only tasks j < 5 are routed (ddpg.py:183)
            per = batch_size // self.V
            for v, bufs in enumerate(self._rank_buffers):
                counts = routed[v * per:(v + 1) * per].sum(axis=0)
                for j in range(self.nb_tasks):
                    if counts[j]:
                        buf = bufs[j + 1]
                        buf.current_size = min(buf.size, buf.current_size + int(counts[j]))
                        buf.n_transitions_stored += int(counts[j]) * self.T
        self._tables_sizes = self._sizes_key()

    def prefetch_activity(self, episode_batch, in_route=False):
        """Called by the batched RolloutWorker right after it enqueued a rollout: the task-activity test of the coming
        store_episode (ddpg.py:179-184) and its D2H copy are enqueued now, so that they arrive with the rollout flags
        the worker waits for anyway -- one host sync per cycle instead of two.  in_route: the store will be routed on the
        device (expect_async_store follows): its routing launch evaluates the flags itself, only the buffers are set up."""
        if not (self.structure in ('curious', 'task_experts') and self._multi_buffer()):
            return
        layout = self._layout
        staging = as_records(episode_batch, layout)
        batch_size = staging.shape[0]
        na = batch_size * self.nb_tasks
        if getattr(self, '_route_bufs', None) is None or self._route_bufs[0].numel() < na:
            self._route_bufs = (torch.empty(na, dtype=torch.int32, device=self.device),
                                torch.empty(na, dtype=torch.int32).pin_memory(),
                                torch.empty(na, dtype=torch.int32).pin_memory(),
                                torch.empty(na, dtype=torch.int64).pin_memory(),
                                torch.empty(na, dtype=torch.int32, device=self.device),
                                torch.empty(na, dtype=torch.int64, device=self.device))
        self._activity_in_route = bool(in_route)
        if not in_route:
            ops.episode_activity(staging, layout, self.sample_transitions.tasks, batch_size, self._route_bufs[0][:na])
            self._route_bufs[1][:na].copy_(self._route_bufs[0][:na], non_blocking=True)
        self._activity_prefetched = (staging.data_ptr(), batch_size)

    def _update_stats(self, staging, batch_size, skip=None):
        """HER-sample batch_size * T transitions from the fresh episodes and feed both normalisers (ddpg.py:207-223).
        skip: the NaN word of the rollout (device) on the device-routed path -- non-zero = nothing is accumulated."""
        layout = self._layout
        n = batch_size * self.T
        if self.rng_mode == 'numpy':
            ep, t, u_her, u_off, given = self.sample_transitions.draw(batch_size, self.T, n)
            plan = upload_plan(n, ep, t, u_her, u_off, ttr=given)
            rng = None
        else:
            plan, rng = None, self._stats_rng(batch_size, n)
        if getattr(self, '_stats_batch', None) is None or self._stats_batch.shape[0] != n:
            self._stats_batch = torch.empty([n, layout.batch_stride], dtype=torch.float32, device=self.device)
        batch = self._stats_batch
        P = self.sample_transitions.params(self.clip_obs, self.relative_goals)
        # (virtual ranks: every rank draws its batch_size / V * T transitions from ITS episodes -- "buffer" v of the staging
        #  block = the records of rank v)
        ops.her_sample(staging, (batch_size // self.V) * layout.rec_floats if self.V > 1 else 0, layout,
                       self.sample_transitions.tasks, P, n, batch, plan=plan, rng=rng)
        cols = layout.batch_cols
        if self.dimo + self.dimg > 256:
            # wider than the paired kernel's one workgroup: the two normalisers one after the other (ddpg.py:216-223)
            assert skip is None
            self.o_stats.update(batch[:, cols['o'][0]:cols['o'][0] + self.dimo])
            self.g_stats.update(batch[:, cols['g'][0]:cols['g'][0] + self.dimg])
            recompute_many([self.o_stats, self.g_stats], packed=self._stats_acc)
            return
        # both normalisers from the one batch in two launches; on a single rank the second one also recomputes the
        # statistics, with several ranks the (packed) accumulators are all-reduced first (normalizer.py:84-94)
        need = ops.norm_pair_scratch_doubles(n, self.dimo, self.dimg)
        if getattr(self, '_stats_scratch', None) is None or self._stats_scratch.numel() < need:
            self._stats_scratch = torch.empty(need, dtype=torch.float64, device=self.device)
        # one rank: the finishing launch also recomputes the statistics.  Several (real or virtual) ranks: the accumulators
        # hold the SUM over this process's virtual ranks; all-reduced over the processes, then divided by the number of
        # ranks (normalizer.py:84-94: the MEAN over ranks of every rank's local sums)
        single = not dist.is_distributed() and self.V == 1
        ops.norm_update_pair(batch, n, batch.stride(0), cols['o'][0], self.dimo, cols['g'][0], self.dimg,
                             self.o_stats.acc, self.g_stats.acc, self.o_stats.state if single else None,
                             self.g_stats.state if single else None, self.o_stats.eps, self.g_stats.eps,
                             self._stats_scratch, skip=skip)
        if not single:
            recompute_many([self.o_stats, self.g_stats], packed=self._stats_acc, ranks_per_process=self.V)

    def _stats_rng(self, n_episodes, n):
        """Sampler description of the normaliser batch: n transitions from the n_episodes fresh episodes; with virtual
        ranks n / V from each rank's n_episodes / V (one table row [prefix 0, prefix 1, size, alias, task] per rank)."""
        r = _lib.SampleRng()
        V = self.V
        if getattr(self, '_stats_tables_key', None) != (n_episodes, n):
            rows = [[0, n // V, n_episodes // V, v, -1] for v in range(V)]
            self._stats_tables = torch.tensor(rows, dtype=torch.int32, device=self.device).reshape(-1)
            self._stats_tables_key = (n_episodes, n)
        r.seed = (self.seed * 7919 + 17 + self._grank0() * RANK_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF
        r.step_ctr = None
        self._stats_calls = getattr(self, '_stats_calls', 0) + 1
        r.step_host = self._stats_calls
        t = self._stats_tables
        r.prop_prefix, r.cur_size, r.buf_alias, r.buf_task = (t[0:].data_ptr(), t[2:].data_ptr(), t[3:].data_ptr(),
                                                              t[4:].data_ptr())
        r.nbuf = 1
        if V > 1:
            r.rank_rows, r.rank_tab_stride, r.rank_seed_stride = n // V, 5, RANK_SEED_STRIDE
        return r

    def get_current_buffer_size(self):
        self.settle()                                                # (virtual ranks: rank 0's, like everything a rank logs)
        return sum([self.buffer[i].get_current_size() for i in range(self.nb_tasks)])

    # ------------------------------------------------------------------ optimiser plumbing
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

    # ------------------------------------------------------------------ sampling
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

    # ------------------------------------------------------------------ training
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

    # ------------------------------------------------------------------ several ranks, fused IPC all-reduce + Adam (opt-in)
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
        self._ipc_verdict()                                          # of the previous run (its copy arrived long ago)
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

    def _ipc_verdict(self, wait=True):
        ipc = getattr(self, '_ipc', None)
        if ipc is None or not ipc['err_pending']:
            return
        if wait:
            ipc['err_ev'].synchronize()
        elif not ipc['err_ev'].query():
            return
        ipc['err_pending'] = False
        if int(ipc['err_pin'][0]):
            raise _lib.CuriousHipError('curious_allreduce_adam_ipc: a wait for a peer rank gave up (rank %d): that epoch '
                                       "was skipped on this rank, the replicas may differ" % dist.rank())

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

    def _init_target_net(self):
        ops.polyak_update(self.theta_target, self.theta, 0.0)        # ddpg.py:459-460

    def update_target_net(self):
        # the fault word travels to the host every FAULT_CHECK_EVERY-th cycle (train.py:154: once per cycle we are here):
        # a faulted update freezes the parameters until the word is cleared, so a late report loses nothing
        self._fault_tick = getattr(self, '_fault_tick', 0) + 1
        ops.polyak_update(self.theta_target, self.theta, self.polyak)   # ddpg.py:461-462
        if self._fault_tick % FAULT_CHECK_EVERY == 1:
            self._enqueue_fault_check()
        # verdict of an earlier copy, raised when the work of this call is done.  One rank: as soon as the copy has
        # arrived (no stall).  Several ranks: a fault on ANY rank froze ALL of them (the flag element of the gradient
        # all-reduce), and all of them read their verdict at the SAME cycle count, FAULT_CHECK_EVERY - 1 cycles after
        # the copy was enqueued (it arrived long ago) -- every rank raises, clears and resumes in the same cycle, the
        # replicas stay identical
        if not dist.is_distributed():
            self.check_faults(wait=False)
        elif self._fault_tick % FAULT_CHECK_EVERY == 0:
            self._fault_verdict(wait=True)

    # ------------------------------------------------------------------ guard of the in-kernel Q' hand-off
    def _enqueue_fault_check(self):
        """Asynchronous D2H copy of the workspace's fault word, stream-ordered behind everything enqueued so far."""
        if getattr(self, '_fault', None) is None:
            self._fault = ops.fault_word(self.net_cfg, self._Bt, self._workspace)
            self._fault_pin = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._fault_ev = torch.cuda.Event()
        self._fault_pin.copy_(self._fault, non_blocking=True)
        self._fault_ev.record()
        self._fault_pending = True

    def check_faults(self, wait=True):
        """Raises HandoffFault when a consumer of Q' gave up in an update since the last check.  wait=False looks only
        at a copy that has already arrived (the training loop: the verdict of cycle c is read during cycle c + 1);
        wait=True enqueues a fresh copy and waits for it.  The word is cleared before raising, so a caller that catches
        the exception can go on training from the last good parameters."""
        if wait:
            self._enqueue_fault_check()
            self._ipc_verdict(wait=True)                             # (fused IPC all-reduce: a wait for a peer gave up)
        elif dist.is_distributed():
            return                                                   # read at a fixed cycle count: update_target_net
        self._fault_verdict(wait)

    def _fault_verdict(self, wait):
        if not getattr(self, '_fault_pending', False):
            return
        if wait:
            self._fault_ev.synchronize()
        elif not self._fault_ev.query():
            return
        self._fault_pending = False
        n = int(self._fault_pin[0])
        if n:
            ops.fault_word(self.net_cfg, self._Bt, self._workspace, 64).zero_()
            raise HandoffFault("%d consumer wave(s) of the row-local update never received Q' from their target group "
                               '(agent %s, rank %d): the optimiser was skipped from that update on' %
                               (n, self.scope, dist.rank()))

    def clear_buffer(self):
        self.settle()
        for bl in self._rank_buffers:
            for i in range(self.nb_tasks):
                bl[i].clear_buffer()
        self._tables_dirty = True

    # ------------------------------------------------------------------ logging / persistence
    def logs(self, prefix=''):
        logs = []
        logs += [('stats_o/mean', float(self.o_stats.mean.mean()))]
        logs += [('stats_o/std', float(self.o_stats.std.mean()))]
        logs += [('stats_g/mean', float(self.g_stats.mean.mean()))]
        logs += [('stats_g/std', float(self.g_stats.std.mean()))]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + key, val) for key, val in logs]
        return logs

    def _net_arrays(self, vec, critic):
        off = 0 if critic else self.off_pi
        flat = vec[off:off + (self.P_Q if critic else self.P_pi)].cpu().numpy()
        out, o = [], 0
        for s in self._shapes(critic):
            n = int(np.prod(s))
            out.append(flat[o:o + n].reshape(s).copy())
            o += n
        return out

    def _load_net_arrays(self, vec, critic, arrays):
        off = 0 if critic else self.off_pi
        flat = np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrays])
        assert flat.size == (self.P_Q if critic else self.P_pi)
        vec[off:off + flat.size].copy_(torch.from_numpy(flat))

    def _stats_arrays(self, nz):
        d, s = nz.size, nz.state.cpu().numpy()
        # TF global-variable creation order of Normalizer (normalizer.py:31-45): sum, sumsq, count, mean, std
        return [s[:d].copy(), s[d:2 * d].copy(), s[2 * d:2 * d + 1].copy(), s[2 * d + 1:3 * d + 1].copy(),
                s[3 * d + 1:].copy()]

    def save_weights(self, path):
        """Pickled list of lists in the reference's order: main/Q, main/pi, target/Q, target/pi, o_stats, g_stats
        (ddpg.py:481-497)."""
        with open(path + '_weights.pkl', 'wb') as f:
            pickle.dump(self._weights_lists(), f)

    def _weights_lists(self):
        return [self._net_arrays(self.theta, True), self._net_arrays(self.theta, False),
                self._net_arrays(self.theta_target, True), self._net_arrays(self.theta_target, False),
                self._stats_arrays(self.o_stats), self._stats_arrays(self.g_stats)]

    def _set_weights_lists(self, weights):
        assert len(weights) == 6, 'expected main/Q, main/pi, target/Q, target/pi, o_stats, g_stats (ddpg.py:483-484)'
        self._load_net_arrays(self.theta, True, weights[0])
        self._load_net_arrays(self.theta, False, weights[1])
        self._load_net_arrays(self.theta_target, True, weights[2])
        self._load_net_arrays(self.theta_target, False, weights[3])
        for nz, arrs in ((self.o_stats, weights[4]), (self.g_stats, weights[5])):
            nz.state.copy_(torch.from_numpy(np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in arrs])))

    def load_weights(self, path):
        with open(path + '_weights.pkl', 'rb') as f:
            weights = pickle.load(f)                                 # ddpg.py:499-509
        self._set_weights_lists(weights)

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
