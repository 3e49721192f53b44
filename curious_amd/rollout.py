"""Rollout worker.  Mirrors RolloutWorker baselines/her/rollout.py:13-501 (constructor kwargs, methods, attributes).

Two execution paths behind the same interface:
  * batched (make_env has .make_batched): all `rollout_batch_size` environments live on the GPU; one T-step rollout
    is T x {actor forward, noise epilogue, env step} kernel launches with no host round trip, and the episode record
    is written in place in the staging block that DDPG.store_episode copies from (rollout.py:209-303 without the
    Python lists, np.array(...).swapaxes of util.py:174-184, or per-env loops);
  * generic (a Python list of gym-style envs): the reference's loop, kept for real environments.
MPI call sites replaced: C7/C8 each rank draws its own tasks/goals from the shared p; C9 one
all-gather of (task, success, valid) per rollout after which every rank updates identical competence queues, which
makes the C10 broadcasts of p / CP unnecessary.
"""
import pickle
from collections import deque
from itertools import islice

import numpy as np
import torch

from curious_amd import dist
from curious_amd.queues import CompetenceQueue, task_probabilities
from curious_amd.rollout_eval import _NOTHING, EvalRolloutsMixin
from curious_amd.util import convert_episode_to_batch_major, store_args



class RolloutWorker(EvalRolloutsMixin):
    @store_args
    def __init__(self, make_env, policy, dims, logger, T, rollout_batch_size=1, exploit=False, use_target_net=False,
                 compute_Q=False, noise_eps=0, random_eps=0, history_len=100, render=False, structure='curious',
                 task_selection='random', goal_selection='random', queue_length=500, eval=False, unique_task=None,
                 temperature=None, **kwargs):
        """Same arguments as the reference (rollout.py:16-40)."""
        assert self.T > 0
        if goal_selection not in ('random', 'active'):
            raise ValueError("goal_selection must be 'random' or 'active' (rollout.py:37)")
        # the GPU-resident batched env is multi-task; the flat structure (rollout.py:93-95) runs the generic host loop
        self.batched = hasattr(make_env, 'make_batched') and structure != 'flat'
        self.rank = dist.rank()
        # virtual ranks (DDPG virtual_ranks = V): this worker runs the rollouts of V of the reference's ranks -- V x
        # rollout_batch_size envs in one batched env, env ids, exploit decisions and task / goal draws per rank
        # (a list of policies -- the task_experts evaluator, rollout.py:212-224 -- stands for the ranks its experts stand for)
        pol0 = policy[0] if isinstance(policy, (list, tuple)) else policy
        self.V = int(getattr(pol0, 'virtual_ranks', 1) or 1)
        # global rank of this process's first virtual rank, ranks of the job (uneven layouts: dist.virtual_layout)
        self.rank_base = int(getattr(pol0, 'rank_base', self.rank * self.V))
        self.nb_cpu = int(getattr(pol0, 'total_ranks', dist.world_size() * self.V))
        self._uneven = self.nb_cpu != dist.world_size() * self.V     # (the ranks' record blocks then differ in length)
        self._slot0 = self.rank_base * rollout_batch_size            # this process's first rollout among the job's
        self._nloc = rollout_batch_size * self.V                     # envs of this process
        if self.V > 1 and not (hasattr(make_env, 'make_batched') and structure in ('curious', 'task_experts')):
            raise ValueError("virtual ranks need the GPU-resident batched env and structure 'curious' or 'task_experts'")
        self._vrng = None                                            # per-virtual-rank host streams (seed_ranks)
        if self.batched:
            self.benv = (make_env.make_batched(self._nloc, env_id0=self.rank_base * rollout_batch_size, pad_to=4) if self.V > 1
                         else make_env.make_batched(self._nloc, env_id0=self.rank_base * rollout_batch_size))
            self.envs = [self.benv]          # attribute kept; the batch is ONE object
            spec = self.benv
        else:
            self.envs = [make_env() for _ in range(rollout_batch_size)]
            spec = self.envs[0].unwrapped
        self.info_keys = [key.replace('info_', '') for key in dims.keys() if key.startswith('info_')]
        self.success_history = deque(maxlen=history_len)
        self.reward_history = deque(maxlen=history_len)
        self.Q_history = deque(maxlen=history_len)
        self.n_episodes = 0
        self.g = np.empty((rollout_batch_size, dims['g']), np.float32)
        self.initial_o = np.empty((rollout_batch_size, dims['o']), np.float32)
        self.initial_ag = np.empty((rollout_batch_size, dims['ag']), np.float32)
        self.nb_goals_per_rollout = self.nb_cpu * rollout_batch_size
        self.nb_tasks = spec.nb_tasks
        self.C = np.zeros([self.nb_tasks])
        self.CP = np.zeros([self.nb_tasks])
        if structure in ('curious', 'task_experts'):
            self.tasks_ag_id = spec.tasks_ag_id
            self.tasks_g_id = spec.tasks_g_id
            self.task_descr = np.empty((rollout_batch_size, self.nb_tasks), np.float32)
            self.p = 1 / self.nb_tasks * np.ones([self.nb_tasks])
            if structure == 'task_experts' and not self.eval:
                self.p = np.zeros([self.nb_tasks])
                self.p[unique_task] = 1
            self.competence_computers = [CompetenceQueue(window=queue_length) for _ in range(self.nb_tasks)]
            self.task_history = deque()
            self.goal_history = deque()
            self.split_histories = [deque() for _ in range(self.nb_tasks)]
            if goal_selection == 'active':                       # SAGG-RIAC per task (rollout.py:81-87)
                from curious_amd.active_goal_sampling import SAGG_RIAC
                ids = self.tasks_g_id
                lo = [spec._compute_goal(-np.ones([len(ids[i])]), i, eval=False)[0][ids[i]] for i in range(self.nb_tasks)]
                hi = [spec._compute_goal(np.ones([len(ids[i])]), i, eval=False)[0][ids[i]] for i in range(self.nb_tasks)]
                self.goal_selectors = [SAGG_RIAC(lo[i], hi[i]) for i in range(self.nb_tasks)]
        elif structure == 'flat':
            for i in range(rollout_batch_size):
                self.envs[i].unwrapped.set_flat_env()
        self.stochastic_reset = False
        self.count = -1
        if not self.batched:
            self.reset_all_rollouts()
        self.clear_history()

    # ================================================================== generic path (list of host envs)
    def reset_rollout(self, i):
        """rollout.py:102-162; each rank samples for itself."""
        obs = self.envs[i].reset()
        if self.structure in ('curious', 'task_experts'):
            task = int(np.random.choice(range(self.nb_tasks), p=self.p, size=1)[0])          # rollout.py:120
            active_goal = self.goal_selection == 'active' and not self.eval                   # rollout.py:121-128
            if active_goal:
                goal = self.goal_selectors[task].sample_goal()
            else:
                goal = np.random.uniform(-1, 1, len(self.tasks_g_id[task]))                   # rollout.py:129
            self.tasks[self._slot0 + i] = task
            self.goals[self._slot0 + i] = \
                self.envs[i].unwrapped._compute_goal(goal, task, eval=self.eval)[0][self.tasks_g_id[task]]
            self.count += 1
            obs = self.envs[i].unwrapped.reset_task_goal(goal=goal, task=task, directly=active_goal, eval=self.eval)
        else:
            goal = np.random.uniform(-1, 1, self.dims['g'])
            obs = self.envs[i].unwrapped.reset_task_goal(goal=goal)
        self.initial_o[i] = obs['observation']
        self.initial_ag[i] = obs['achieved_goal']
        self.g[i] = obs['desired_goal']
        if self.structure in ('curious', 'task_experts'):
            self.task_descr[i] = obs['mask']

    def reset_all_rollouts(self):
        self.goals = [[] for _ in range(self.nb_goals_per_rollout)]
        if self.structure in ('curious', 'task_experts'):
            self.tasks = [[] for _ in range(self.nb_goals_per_rollout)]
        for i in range(self.rollout_batch_size):
            self.reset_rollout(i)

    def _decide_exploit_ranks(self):
        """rollout.py:183-189 for every virtual rank, each from its own stream (train.py:242-243): _exploit_v[v].  The
        worker-wide `exploit` = any of them (the cycle then has the competence exchange)."""
        if self.eval:
            self._exploit_v = np.ones(self.V, bool)
        else:
            self._exploit_v = np.array([self._rng(v).random_sample() < 0.1 for v in range(self.V)])
        self.exploit = bool(self._exploit_v.any())

    def _rng(self, v):
        if self._vrng is None:
            self.seed_ranks([12345 + 1000000 * (self.rank_base + k) for k in range(self.V)])
        return self._vrng[v]

    def seed_ranks(self, seeds):
        """Host streams of the virtual ranks: seeds[v] = what train.py:242-243 gives global rank rank * V + v."""
        assert len(seeds) == self.V
        self._vrng = [np.random.RandomState(int(s) % (2 ** 32)) for s in seeds]

    def _decide_exploit(self):
        """rollout.py:183-189."""
        if self.V > 1:
            return self._decide_exploit_ranks()
        if self.structure in ('curious', 'task_experts') and not self.eval:
            self.exploit = True if np.random.random() < 0.1 else False
            if self.exploit and self.structure == 'curious':
                self.p = 1 / self.nb_tasks * np.ones([self.nb_tasks])
        elif self.eval:
            self.exploit = True
            self.p = 1 / self.nb_tasks * np.ones([self.nb_tasks])

    def generate_rollouts(self):
        """Returns (episode batch, CP, n_episodes) (rollout.py:177-406)."""
        if self.batched:
            return self._generate_rollouts_batched()
        self._decide_exploit()
        self.reset_all_rollouts()
        B = self.rollout_batch_size
        o = np.empty((B, self.dims['o']), np.float32)
        ag = np.empty((B, self.dims['ag']), np.float32)
        o[:] = self.initial_o
        ag[:] = self.initial_ag
        obs, achieved_goals, acts, goals, successes = [], [], [], [], []
        info_values = [np.empty((self.T, B, self.dims['info_' + key]), np.float32) for key in self.info_keys]
        Qs, task_descrs, changes = [], [], []
        multi = self.structure in ('curious', 'task_experts')
        for t in range(self.T):
            if self.structure == 'task_experts' and self.eval:       # rollout.py:212-224
                act_output = np.zeros([B, self.dims['u']])
                q_output = np.zeros([B, 1])
                for i in range(B):
                    tsk = int(np.argmax(self.task_descr[i]))
                    out = self.policy[tsk].get_actions(
                        o[i].reshape([1, -1]), ag[i].reshape([1, -1]), self.g[i].reshape([1, -1]),
                        task_descr=self.task_descr[i].reshape([1, -1]), compute_Q=self.compute_Q,
                        noise_eps=self.noise_eps if not self.exploit else 0.,
                        random_eps=self.random_eps if not self.exploit else 0., use_target_net=self.use_target_net)
                    if self.compute_Q:
                        act_output[i, :], q_output[i, 0] = out[0], np.asarray(out[1]).reshape(-1)[0]
                    else:
                        act_output[i, :] = out
                policy_output = [act_output, q_output] if self.compute_Q else act_output
            else:
                policy_output = self.policy.get_actions(
                    o, ag, self.g, task_descr=self.task_descr if self.structure == 'curious' else None,
                    compute_Q=self.compute_Q, noise_eps=self.noise_eps if not self.exploit else 0.,
                    random_eps=self.random_eps if not self.exploit else 0., use_target_net=self.use_target_net)
            if self.compute_Q:
                u, Q = policy_output
                Qs.append(Q)
            else:
                u = policy_output
            if u.ndim == 1:
                u = u.reshape(1, -1)
            o_new = np.empty((B, self.dims['o']))
            ag_new = np.empty((B, self.dims['ag']))
            success = np.zeros(B)
            r_competence = np.zeros(B)
            for i in range(B):                                        # rollout.py:250-263
                if self.render:
                    self.envs[i].render()
                curr_o_new, r_competence[i], _, info = self.envs[i].step(u[i])
                if 'is_success' in info:
                    success[i] = info['is_success']
                o_new[i] = curr_o_new['observation']
                ag_new[i] = curr_o_new['achieved_goal']
                self.g[i] = curr_o_new['desired_goal']
                for idx, key in enumerate(self.info_keys):
                    info_values[idx][t, i] = info[key]
            if np.isnan(o_new).any():                                 # rollout.py:268-271
                self.logger.warning('NaN caught during rollout generation. Trying again...')
                self.reset_all_rollouts()
                return self.generate_rollouts()
            obs.append(o.copy())
            achieved_goals.append(ag.copy())
            successes.append(success.copy())
            acts.append(u.copy())
            goals.append(self.g.copy())
            o[...] = o_new
            ag[...] = ag_new
            if multi:
                task_descrs.append(self.task_descr.copy())
                changes.append(np.abs(achieved_goals[0] - ag) > 1e-3)   # rollout.py:284
        obs.append(o.copy())
        achieved_goals.append(ag.copy())
        episode = dict(o=obs, u=acts, g=goals, ag=achieved_goals)
        if multi:
            episode['task_descr'] = task_descrs
            episode['change'] = changes
        self.initial_o[:] = o
        for key, value in zip(self.info_keys, info_values):
            episode['info_{}'.format(key)] = value
        successful = np.array(successes)[-1, :]
        assert successful.shape == (B,)
        mean_Q = np.mean(Qs) if self.compute_Q else None
        tasks_now = [self.envs[i].unwrapped.task for i in range(B)] if multi else None
        goals_now = None
        if multi and self.goal_selection == 'active' and not self.eval and self.exploit:           # rollout.py:321
            goals_now = np.stack([np.asarray(self.envs[i].unwrapped.goal)[self.tasks_g_id[tasks_now[i]]]
                                  for i in range(B)])
        self._finish_rollout(successful, r_competence, mean_Q, tasks_now, goals_now)
        return convert_episode_to_batch_major(episode), self.CP, self.n_episodes

    # ================================================================== batched path (GPU-resident envs)
    # ------------------------------------------------------------------ rollouts whose flags the host reads late
    def _async_ok(self, env, fused):
        """Nothing the host would compute from this rollout's flags is needed before its updates are enqueued: no exploit
        rollout ON ANY RANK (competence queues, CP and task probabilities stay as they are, rollout.py:318-330), no Q
        statistics, no SAGG-RIAC update, and a policy that can route the episodes on the device (DDPG.can_store_async).
        A rank-local decision: a cycle without exploit rollouts has no host-side collective (see _finish_rollout)."""
        return (fused and not self.eval and not self._any_exploit and not self.compute_Q and self.structure == 'curious'
                and self.goal_selection != 'active' and hasattr(self.policy, 'can_store_async')
                and self.policy.can_store_async(self.rollout_batch_size))

    def settle(self):
        """Process the flags of a rollout that was returned without waiting for them (async_store)."""
        p, self._pending = getattr(self, '_pending', None), None
        if p is None:
            return
        from curious_amd.envs import ResidentRolloutVoid
        try:
            successful, o_has_nan = self.benv.wait_flags()
            lost = bool(np.isnan(successful).any() or o_has_nan)
        except ResidentRolloutVoid as err:
            self._resident_off(err)
            lost = True
        if lost:
            # the sync path generates such a rollout again before anybody sees it (rollout.py:268-271).  Here its updates
            # are already enqueued and the device dropped its episodes instead of storing them: the replacement is
            # generated and stored NOW, one cycle late, so the buffers miss nothing
            self.logger.warning('NaN caught during rollout generation. Trying again...')
            self.n_episodes -= self.rollout_batch_size * self.nb_cpu  # the lost rollout does not count
            self._replace_lost_rollout()
            return
        exploit, self.exploit = self.exploit, False
        any_exploit, self._any_exploit = getattr(self, '_any_exploit', False), False
        self.tasks, self.goals = p['tasks'], p['goals']
        self.n_episodes -= self.rollout_batch_size * self.nb_cpu     # counted when the rollout was returned
        self._finish_rollout(successful, successful - 1.0, None, p['task_list'], None)
        self.exploit, self._any_exploit = exploit, any_exploit

    MAX_NAN_RETRIES = 3                   # the reference recurses without a bound (rollout.py:268-271)

    def _resident_off(self, err):
        """A void weights-resident rollout: switch the process to the streaming rollout kernel (same numbers), once."""
        from curious_amd import ops
        if ops.get_option('resident'):
            self.logger.warning('%s -- switching this process to the streaming rollout kernel' % err)
            ops.set_option('resident', 0)
        for pol in (self.policy if isinstance(self.policy, (list, tuple)) else [self.policy]):
            if hasattr(pol, 'drop_rollout_graphs'):
                pol.drop_rollout_graphs()                         # they captured the resident launch

    def _replace_lost_rollout(self):
        """async_store: the rollout just settled was lost (NaN / void launch) and dropped on the device.  Generate its
        replacement, waiting for its flags, and store it through the policy.  With several ranks the replacement feeds no
        normaliser statistics: their update is a collective the other ranks are not part of at this point."""
        exploit, self.exploit = self.exploit, False               # (only exploration rollouts take the async form)
        any_exploit, self._any_exploit = getattr(self, '_any_exploit', False), False
        try:
            episode, cp, n_ep = self._generate_rollouts_batched(retry=True, force_sync=True)
            self.policy.store_episode(episode, cp, n_ep, update_stats=not dist.is_distributed())
        finally:
            self.exploit, self._any_exploit = exploit, any_exploit

    def _generate_rollouts_batched(self, retry=False, force_sync=False, n_retry=0, redo=None, defer=None):
        """redo = (tasks, goals): generate exactly the rollout that was just enqueued once more (DDPG.rewind_rollout put
        the episode and noise counters back) instead of drawing a new one.
        defer = k (evaluators, generate_eval_rollouts): enqueue only; the flags and the Q sum travel to pinned slot k and
        the returned closure -- called once everything is enqueued -- waits and books the rollout."""
        self.settle()
        if hasattr(self.policy, 'settle'):
            self.policy.settle()
        if not retry:
            self._decide_exploit()
            # Does ANY rank exploit in this cycle?  Exploit rollouts are the only ones that feed the (replicated)
            # competence queues, through an all-gather every rank has to take part in (C9, rollout.py:332-336) -- so the
            # ranks agree here, on the host (gloo side group: Python blocks, the GPU keeps working on what is enqueued),
            # whether this cycle has that exchange at all.  (A rollout regenerated after a NaN keeps the decision: the
            # ranks' host-side exchanges have to stay paired.)
            self._any_exploit = True if self.eval else dist.host_any(self.exploit)
        B, env = self._nloc, self.benv
        experts = isinstance(self.policy, (list, tuple))          # task_experts evaluator (rollout.py:212-224)
        tasks, goals = redo if redo is not None else self._draw_tasks_goals(experts)
        fused = not experts and hasattr(self.policy, 'can_act_and_step') and \
            self.policy.can_act_and_step(env, self.compute_Q)
        # (a rollout that is ONE launch: its reset launch heads the captured rollout, where it also advances the noise base)
        one_launch = fused and hasattr(self.policy, 'act_rollout')
        env.reset_all(tasks, goals, launch=not one_launch)
        env._reset_pending = one_launch
        self.count += B
        if self.__dict__.get('_ep_host') is not None:
            self._ep_host += 1                                    # (mirror of env.episode: the reset advances it)
        q_sum = torch.zeros((), device=env.device) if self.compute_Q else None
        if experts:
            q_sum = self._expert_steps(env, tasks, q_sum)
        if fused and hasattr(self.policy, 'act_rollout') and self.V > 1:
            # the noise is switched off rank by rank inside the launch (a rank that exploits: rollout.py:183-189)
            self.policy.act_rollout(env, self.T, noise_eps=0. if self.eval else self.noise_eps,
                                    random_eps=0. if self.eval else self.random_eps,
                                    use_target_net=self.use_target_net, exploit=self._exploit_v, evaluation=self.eval)
        elif fused and hasattr(self.policy, 'act_rollout'):
            self.policy.act_rollout(env, self.T, noise_eps=self.noise_eps if not self.exploit else 0.,
                                    random_eps=self.random_eps if not self.exploit else 0.,
                                    use_target_net=self.use_target_net, evaluation=self.eval)
        elif self.V > 1 and not self.eval:
            raise NotImplementedError('virtual ranks: training rollouts need the fused rollout (DDPG.act_rollout)')
        if fused and hasattr(self.policy, 'act_rollout') and self.compute_Q:
            # the fused rollout records no Q: one actor + critic forward over its recorded rows (DDPG.rollout_q_sum)
            q_sum = self.policy.rollout_q_sum(env, self.T, use_target_net=self.use_target_net)
        noise_eps = self.noise_eps if not self.exploit else 0.
        random_eps = self.random_eps if not self.exploit else 0.
        graphed = not fused and not experts and hasattr(self.policy, 'can_eval_rollout') and \
            self.policy.can_eval_rollout(env, noise_eps, random_eps)
        if graphed:                                              # evaluator / exploit rollouts: one graph replay
            q_acc = self.policy.eval_rollout(env, self.T, use_target_net=self.use_target_net,
                                             compute_Q=self.compute_Q)
            if self.compute_Q:
                q_sum = q_acc
        for t in range(self.T if not (experts or graphed or (fused and hasattr(self.policy, 'act_rollout'))) else 0):
            if fused:
                self.policy.act_and_step(env, t, noise_eps=self.noise_eps if not self.exploit else 0.,
                                         random_eps=self.random_eps if not self.exploit else 0.,
                                         use_target_net=self.use_target_net)
                continue
            out = self.policy.get_actions(env.o, env.ag, env.g, task_descr=env.td, compute_Q=self.compute_Q,
                                          noise_eps=self.noise_eps if not self.exploit else 0.,
                                          random_eps=self.random_eps if not self.exploit else 0.,
                                          use_target_net=self.use_target_net)
            if self.compute_Q:
                u, Q = out
                q_sum += Q.mean()
            else:
                u = out
            env.step_all(u, t)
        # success flags and the NaN check of rollout.py:268-271 in ONE D2H sync per rollout
        async_ok = not force_sync and self._async_ok(env, fused and hasattr(self.policy, 'act_rollout'))
        if not self.eval and hasattr(self.policy, 'prefetch_activity'):
            # arrives with the flags: one host sync per cycle.  (A device-routed store evaluates the flags in its routing
            # launch: nothing to enqueue here)
            self.policy.prefetch_activity(env.episode_views(), in_route=async_ok)
        if async_ok:
            # return without waiting: the policy routes the episodes on the device, the flags are read in settle()
            env.request_flags()
            task_list = tasks.tolist()
            tk = [_NOTHING] * self.nb_goals_per_rollout
            tk[self._slot0:self._slot0 + B] = task_list
            self._pending = dict(tasks=tk, goals=[_NOTHING] * self.nb_goals_per_rollout, task_list=task_list)
            self.n_episodes += self.rollout_batch_size * self.nb_cpu
            views = env.episode_views()
            self.policy.expect_async_store(views, env.flags[env.n:env.n + 1], env._flags_pin[env.n:env.n + 1])
            return views, self.CP, self.n_episodes
        if getattr(self.policy, '_async_batch', None) is not None:
            self.policy._async_batch = None                       # a marked rollout that was never stored: forget it
        if defer is not None:
            env.request_flags(slot=defer)
            q_pin = None
            if self.compute_Q:
                pins = self.__dict__.setdefault('_q_slots', [])
                while len(pins) <= defer:
                    pins.append(torch.zeros(1, dtype=torch.float32).pin_memory())
                q_pin = pins[defer]
                q_pin.copy_(q_sum.reshape(1), non_blocking=True)
                env._flags_ready.record()                         # (behind the Q copy as well)
            exploit, any_exploit, task_list = self.exploit, self._any_exploit, tasks.tolist()

            def finish():
                from curious_amd.envs import ResidentRolloutVoid
                try:
                    successful, o_has_nan = env.wait_flags(slot=defer)
                except ResidentRolloutVoid as err:
                    self._resident_off(err)
                    successful, o_has_nan = np.zeros(B), True
                if np.isnan(successful).any() or o_has_nan:
                    # (the sync path regenerates; a deferred rollout's successors are enqueued already: book this one anew)
                    self.logger.warning('NaN caught during rollout generation. Trying again...')
                    self._generate_rollouts_batched(retry=True, force_sync=True)
                    return
                saved = self.exploit, self._any_exploit
                self.exploit, self._any_exploit = exploit, any_exploit
                self.tasks = [_NOTHING] * self.nb_goals_per_rollout
                self.tasks[self._slot0:self._slot0 + B] = task_list
                self.goals = [_NOTHING] * self.nb_goals_per_rollout
                self._finish_rollout(successful, successful - 1.0,
                                     float(q_pin[0]) / self.T if self.compute_Q else None, task_list, None)
                self.exploit, self._any_exploit = saved
            return finish
        from curious_amd.envs import ResidentRolloutVoid
        try:
            successful, o_has_nan = env.fetch_flags()             # written by the last env step of the rollout
        except ResidentRolloutVoid as err:
            # the launch was not fully resident: nothing of it may be used.  Switch the process to the streaming kernel
            # and generate the SAME rollout again -- same tasks and goals, same episode numbers, same noise counters: the
            # job goes on with exactly the numbers it would have had
            if n_retry >= self.MAX_NAN_RETRIES:
                raise
            self._resident_off(err)
            self.policy.rewind_rollout(env, self.T, evaluation=self.eval)
            self.count -= B
            if self.__dict__.get('_ep_host') is not None:
                self._ep_host -= 1
            return self._generate_rollouts_batched(retry=True, force_sync=force_sync, n_retry=n_retry + 1,
                                                   redo=(tasks, goals))
        if np.isnan(successful).any() or o_has_nan:
            if n_retry >= self.MAX_NAN_RETRIES:
                raise RuntimeError('%d rollouts in a row produced NaN observations (rollout.py:268-271 would try '
                                   'again for ever): the policy has diverged' % (n_retry + 1))
            self.logger.warning('NaN caught during rollout generation. Trying again...')
            return self._generate_rollouts_batched(retry=True, force_sync=force_sync, n_retry=n_retry + 1)
        mean_Q = float(q_sum) / self.T if self.compute_Q else None
        task_list = tasks.tolist()
        self.tasks = [_NOTHING] * self.nb_goals_per_rollout
        self.tasks[self._slot0:self._slot0 + B] = task_list
        self.goals = [_NOTHING] * self.nb_goals_per_rollout
        goals_now = None
        if self.goal_selection == 'active' and not self.eval and self.exploit:
            goals_now = 0.5 * goals                               # the envs' goals on their task slots (goal space)
        self._finish_rollout(successful, successful - 1.0, mean_Q, task_list, goals_now)
        return env.episode_views(), self.CP, self.n_episodes

    # ================================================================== statistics, competence, task probabilities
    def _finish_rollout(self, successful, r_competence, mean_Q, tasks_now, goals_now=None):
        """rollout.py:305-404.  goals_now: [B, len(task slots)] goals of the exploit rollouts for SAGG-RIAC."""
        B = self._nloc                                               # (virtual ranks: the envs of all of them)
        self.success_history.append(np.mean(successful))
        self.reward_history.append(r_competence)
        if self.compute_Q:
            self.Q_history.append(mean_Q)
        self.n_episodes += self.rollout_batch_size * self.nb_cpu
        if self.structure not in ('curious', 'task_experts'):
            return
        if not getattr(self, '_any_exploit', self.exploit or dist.is_distributed()) and self.goal_selection != 'active':
            # Only exploit rollouts count (rollout.py:318-330): when no rank had one, nothing valid would arrive, the
            # competence queues would see empty lists and C, CP and the task probabilities come out as they went in.
            # Skipped on every rank alike: this sits between the end of a rollout and the launch of the updates that
            # wait for it.  (_any_exploit: agreed between the ranks at the start of the rollout.)
            self.task_history.extend(list(self.tasks))
            self.goal_history.extend(list(self.goals))
            return
        # C9: gather (task, success, valid) of every rank; only exploit rollouts count (rollout.py:318-330)
        rec = np.zeros([B, 3], np.float64)
        rec[:, 0] = tasks_now
        rec[:, 1] = successful
        rec[:, 2] = 1.0 if self.exploit else 0.0
        if self.V > 1:                                               # only the rollouts of ranks that exploited count
            rec[:, 2] = np.repeat(self._exploit_v.astype(np.float64), self.rollout_batch_size)
        allrec = dist.allgather_numpy(rec, uneven=self._uneven)
        valid = allrec[:, 2] != 0
        task_ids = allrec[:, 0].astype(np.int64)
        task_succ_list = [allrec[valid & (task_ids == task), 1].tolist() for task in range(self.nb_tasks)]
        for task in range(self.nb_tasks):
            self.competence_computers[task].update(task_succ_list[task])   # rollout.py:355-356
        if self.goal_selection == 'active' and not self.eval:             # rollout.py:357-365
            gdim = len(self.tasks_g_id[0])
            grec = np.zeros([B, gdim], np.float64) if goals_now is None else np.asarray(goals_now, np.float64)
            allgoals = dist.allgather_numpy(grec, uneven=self._uneven)     # same order as allrec: every rank agrees
            for task in range(self.nb_tasks):
                sel = valid & (task_ids == task)
                new_split, _ = self.goal_selectors[task].update([g.astype(np.float32) for g in allgoals[sel]],
                                                                allrec[sel, 1].tolist())
                self.split_histories[task].append(
                    [self.goal_selectors[task].get_regions, self.goal_selectors[task].probas] if new_split else None)
        self.C = np.array([self.get_C()]).squeeze()
        self.task_history.extend(list(self.tasks))
        self.goal_history.extend(list(self.goals))
        if not self.eval:
            if self.task_selection == 'active_competence_progress' and self.structure != 'task_experts':
                self.CP = np.array([self.get_CP()]).squeeze()
                self.p = task_probabilities(self.CP, self.nb_tasks, 0.4)    # epsilon hard-coded (rollout.py:383)
            elif self.structure == 'task_experts':
                self.p = np.zeros([self.nb_tasks])
                self.p[self.unique_task] = 1

    def clear_history(self):
        self.settle()
        self.success_history.clear()
        self.reward_history.clear()
        self.Q_history.clear()

    def clear_competence_queue(self):
        for i in range(self.nb_tasks):
            self.competence_computers[i].clear_queue()

    def current_success_rate(self):
        self.settle()
        return np.mean(self.success_history)

    def current_mean_Q(self):
        return np.mean(self.Q_history)

    def save_policy(self, path, snapshot=None):
        """rollout.py:425-433.  The reference swallows every exception of save_weights because a task_experts policy
        is a list without that method; here the list case is handled (one weights file per expert) and real errors
        surface."""
        snap = snapshot if snapshot is not None else self.snapshot_policy()
        writer = getattr(self, 'writer', None)
        if writer is None:
            self._write_policy(path, snap)
        else:
            writer.submit(lambda: self._write_policy(path, snap))     # (experiment.train: curious_amd.util.BackgroundWriter)

    def snapshot_policy(self):
        """Host copies of everything save_policy writes (one D2H round per network vector), valid for the point in time of
        the call (logs() saves the same policy up to three times: one snapshot serves all of them)."""
        from curious_amd.util import PolicySnapshot
        if isinstance(self.policy, (list, tuple)):
            return [PolicySnapshot(p) for p in self.policy]
        return PolicySnapshot(self.policy)

    @staticmethod
    def _write_policy(path, snap):
        with open(path, 'wb') as f:
            pickle.dump(snap, f)
        if isinstance(snap, (list, tuple)):
            for i, s in enumerate(snap):
                with open(path + str(i) + '_weights.pkl', 'wb') as f:
                    pickle.dump(s.state['weights'], f)             # DDPG.save_weights (ddpg.py:481-497)
        else:
            with open(path + '_weights.pkl', 'wb') as f:
                pickle.dump(snap.state['weights'], f)

    def save_goal_task_history(self, path):
        pass                                                          # rollout.py:437-449 (commented out upstream)

    def logs(self, prefix='worker'):
        self.settle()
        logs = []
        logs += [('success_rate', np.mean(self.success_history))]
        logs += [('avg_reward', np.mean(self.reward_history))]
        if self.compute_Q:
            logs += [('mean_Q', np.mean(self.Q_history))]
        logs += [('episode', self.n_episodes)]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + key, val) for key, val in logs]
        return logs

    def additional_logs(self, prefix='worker'):
        self.settle()
        logs = []
        if self.structure in ('curious', 'task_experts'):
            for i in range(self.nb_tasks):
                Cs = self.get_C()
                logs += [('C_task' + str(i), "%.3g" % Cs[i])]
                if not self.eval:
                    CPs = self.get_CP()
                    logs += [('CP_task' + str(i), "%.3g" % CPs[i])]
                    # the last 100 entries (rollout.py:475), without copying a history of millions every epoch
                    hist = list(islice(reversed(self.task_history), 100))
                    per = np.mean(np.array([h == i for h in hist])) if hist else 0.0
                    logs += [('%_task' + str(i), "%.3g" % per)]
                    logs += [('p_task' + str(i), "%.3g" % self.p[i])]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + key, val) for key, val in logs]
        return logs

    def get_CP(self):
        return [cq.CP for cq in self.competence_computers]

    def get_C(self):
        return [cq.C for cq in self.competence_computers]

    def seed(self, seed):
        if self.batched:
            self.benv.seed(seed)
            self._ep_host = 0
            self.__dict__.pop('_eval_env', None)
        else:
            for idx, env in enumerate(self.envs):
                env.seed(seed + 1000 * idx)
