"""Acting half of the agent (DDPG.get_actions baselines/her/ddpg.py:114-161 and what the batched RolloutWorker drives instead
of it): the policy forward for host environments through pinned blocks, the fused act + env-step launches of the
GPU-resident env, the whole T-step rollout as one launch (captured once, replayed), the evaluator's Q values from the
rollout's recorded rows.  Mixed into curious_amd.ddpg.DDPG."""
import numpy as np
import torch

from curious_amd import dist, ops
from curious_amd.dist import RANK_SEED_STRIDE
from curious_amd.update_schedules import CAPTURE_MODE


class ActingMixin:
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

    def _action_block(self, n):
        """The [n, dimu] block the fused acting launches leave their actions in: one per batch size, for the agent's
        lifetime -- captured rollout launches hold its address (a block re-allocated when another env's size came along
        was written to by the graphs of the first)."""
        blocks = self.__dict__.setdefault('_act_blocks', {})
        if n not in blocks:
            blocks[n] = torch.empty([n, self.dimu], dtype=torch.float32, device=self.device)
        return blocks[n]

    def can_act_and_step(self, env, compute_Q):
        """The fused acting step applies to the GPU-resident synthetic env in throughput mode.  compute_Q (the
        evaluator, train.py:308-319): the fused kernels record no Q -- the rollout's Q values are computed afterwards
        from its recorded rows (rollout_q_sum), which needs the whole rollout as one launch (act_rollout)."""
        return (self.rng_mode == 'device' and self.modular
                and self.dimu == 4 and hasattr(env, 'step_all')
                and getattr(env, 'dimo', None) == self.dimo and getattr(env, 'nb_tasks', None) == self.dimtd)

    Q_ROWS = 13056                # rows per launch of rollout_q_sum: one rollout of 256 envs x (T + 1 = 51) rows -- 816
                                  # workgroups of 16 rows, three to a CU (policy_fwd16_kernel); bounds the workspace

    def rollout_q_sum(self, env, T, use_target_net=False, rollouts=None):
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
        # (a batch of slots whose last rollouts are idle: only the rows of the first `live` rollouts are evaluated)
        n_rows = rows.shape[0] if rollouts is None or len(rollouts) < 4 else rollouts[3] * rollouts[1] * (T + 1)
        # (option fwd16: these forwards may take the 16-row form -- the values to 1e-6, not the bits of get_actions)
        with ops.option('fwd16', 1):
            for r0 in range(0, n_rows, u.shape[0]):
                blk = rows[r0:min(r0 + u.shape[0], n_rows)]
                m = blk.shape[0]
                ops.policy_forward(self.net_cfg, theta, blk[:, o_:o_ + self.dimo], blk[:, g_:g_ + self.dimg],
                                   blk[:, td_:td_ + self.dimtd] if self.dimtd > 0 else None, m, self.clip_obs, ws, u[:m],
                                   q[r0:r0 + m].view(m, 1), ag=blk[:, ag_:ag_ + self.dimag],
                                   relative_goals=self.relative_goals,
                                   o_stats=self.o_stats.state if self.normalize_obs else None,
                                   g_stats=self.g_stats.state if self.normalize_obs else None)
        n_used = getattr(env, 'n_used', n)                           # (idle padding envs and the rows t = T do not count)
        if rollouts is not None:
            # a batch of slots (envs.BatchedSyntheticArm wrap): R rollouts of nB envs side by side, the first `used` of every
            # rollout count -- one value per rollout (a fourth entry: only that many rollouts are live, the values of the
            # rest are stale)
            R, nB, used = rollouts[:3]
            return q.view(R, nB, T + 1)[:, :used, :T].mean(dim=(1, 2)) * T
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
        self._act_u = self._action_block(n)
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

    def act_rollout(self, env, T, noise_eps=0., random_eps=0., use_target_net=False, exploit=None, evaluation=False):
        """The T-step acting loop of a batched rollout (rollout.py:226-303 for every env): T x act_and_step, as ONE
        launch (curious_policy_rollout) where the row-local route applies.  With use_graph the launches are captured once
        per (env, noise setting) and replayed; the Philox noise counter is (t + 1) + a device-resident base that advances
        by T per rollout, so replays draw fresh noise and the eager loop draws the same numbers.
        exploit (virtual ranks): one flag per virtual rank -- the envs of a rank that exploits act without exploration
        noise in this rollout (rollout.py:183-189); the envs are V consecutive groups, each drawing its noise from the
        stream of its own global rank.
        evaluation (the evaluator's noise-free rollouts, train.py:156-158): the rollout does not consume the agent's noise
        counter -- the training rollouts draw the noise they would draw without an evaluator -- but the launch still needs
        a counter of its own: the weights-resident kernel tags the words its workgroups exchange with it, and a tag must
        not repeat on an exchange buffer.  Evaluation launches therefore count on `_eval_base` and exchange through a
        workspace of their own."""
        from curious_amd.envs import REWARD_EPS
        n = env.n
        theta = self.theta_target if use_target_net else self.theta
        evaluation = bool(evaluation) and noise_eps == 0 and random_eps == 0
        ws_key = ('eval', n) if evaluation else n
        ws = self._act_ws.get(ws_key)
        if ws is None:
            ws = torch.zeros(ops.workspace_floats(self.net_cfg, n), dtype=torch.float32, device=self.device)
            self._act_ws[ws_key] = ws
        self._act_u = self._action_block(n)
        if getattr(self, '_noise_base', None) is None:
            self._noise_base = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._noise_base_val = 0
            self._roll_graphs = {}
        # ONE logical noise counter for every acting path: the host value `_noise_counter` (get_actions / act_and_step
        # pass it as a kernel argument) and its device mirror `_noise_base` (read by the captured launches below).  The
        # mirror is brought up to date here when host-side acting calls ran since the last rollout, so no two acting
        # calls of one agent ever draw from the same (seed, counter) pair.
        if getattr(self, '_eval_base', None) is None:
            self._eval_base = torch.zeros(1, dtype=torch.int64, device=self.device)
        if not evaluation and self._noise_base_val != self._noise_counter:
            self._noise_base.fill_(self._noise_counter)
            self._noise_base_val = self._noise_counter
        base = self._eval_base if evaluation else self._noise_base
        seed = self.seed * 2654435761 + 12345 + self._grank0() * RANK_SEED_STRIDE     # same stream as get_actions / act_and_step
        u_out = self._act_u
        groups = None
        if self.V > 1:
            group = getattr(env, 'n_used', n) // self.V            # envs per virtual rank (padding envs: groups >= V)
            ng = (n + group - 1) // group
            # one flag vector PER ENV OBJECT, for its lifetime: the captured rollout launches of an env hold its address
            # (the evaluator's slot env has another number of groups than the training worker's: a vector shared between
            #  them was re-allocated at every switch, under the graphs that had captured it)
            bufs = self.__dict__.setdefault('_exploit_bufs', {})
            eb = bufs.get(id(env))
            if eb is None or eb['dev'].numel() != ng:
                eb = dict(dev=torch.zeros(ng, dtype=torch.int32, device=self.device),
                          pins=[torch.zeros(ng, dtype=torch.int32).pin_memory() for _ in range(4)], k=0, env=env)
                bufs[id(env)] = eb
                self._roll_graphs = {k: g for k, g in self._roll_graphs.items() if k[0] != id(env)}
            pin = eb['pins'][eb['k']]                               # (a small ring: the copy is asynchronous)
            eb['k'] = (eb['k'] + 1) % len(eb['pins'])
            pin.zero_()
            if exploit is not None:
                pin[:self.V] = torch.from_numpy(np.asarray(exploit, dtype=np.int32))
            eb['dev'].copy_(pin, non_blocking=True)
            groups = ops.rank_groups(group, RANK_SEED_STRIDE, eb['dev'])

        reset_here = bool(getattr(env, '_reset_pending', False))     # the worker only uploaded the draws (reset_all)
        env._reset_pending = False

        def steps():
            # T x policy_act_env_step (noise counters 1 .. T on top of the base): one launch on the row-local route.
            # The env reset that heads the rollout also advances the noise base (one launch less per cycle): the rollout
            # then starts from base + 1 - T
            if reset_here:
                env.launch_reset(counter=base, delta=T)
            ops.policy_rollout(self.net_cfg, theta, n, self.clip_obs, ws, noise_eps * self.max_u, random_eps, seed,
                               (1 - T) if reset_here else 1,
                               u_out, env._cfg, env.layout, env.env_id0, env.episode, env.tasks, 0, T, env.o, env.ag,
                               env.g, env.td, env.staging, REWARD_EPS, counter_base=base,
                               flags=getattr(env, 'flags', None),
                               o_stats=self.o_stats.state if self.normalize_obs else None,
                               g_stats=self.g_stats.state if self.normalize_obs else None,
                               relative_goals=self.relative_goals, groups=groups)
            if not reset_here:
                ops.counter_add(base, T)

        if not evaluation:
            self._noise_counter += T
            self._noise_base_val = self._noise_counter               # steps() ends with the device-side += T
        if not self.use_graph:
            steps()
            return
        key = (id(env), T, float(noise_eps), float(random_eps), bool(use_target_net), reset_here, evaluation)
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

    def rewind_rollout(self, env, T, evaluation=False):
        """Undo the bookkeeping of the act_rollout that was just enqueued for `env` so that the SAME rollout can be
        generated again (same episode numbers -> same initial states, same noise counters -> same exploration noise):
        used when the weights-resident launch reported itself void (envs.ResidentRolloutVoid) and the rollout is redone
        on the streaming kernel, which computes the same numbers.  (evaluation: the launch counted on its own counter,
        which only has to go on)"""
        if not evaluation:
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
