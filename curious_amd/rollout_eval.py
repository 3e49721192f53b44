"""Evaluation rollouts and the task / goal draws of the batched RolloutWorker (baselines/her/rollout.py:107-143 for every env
at once; train.py:156-158: `for _ in range(n_test_rollouts): evaluator.generate_rollouts()`), mixed into
curious_amd.rollout.RolloutWorker: the draws in the reference's order (per virtual rank from its own stream), the T acting
steps with one expert per task (rollout.py:212-224), and the n evaluation rollouts of an epoch enqueued back to back -- several
of them to a launch where the whole-rollout kernels apply."""
import numpy as np
import torch

# What a slot of the per-rollout task / goal lists holds when this rank has nothing for it (rollout.py:149-150: []).  The
# batched worker fills hundreds of slots per cycle and the histories keep them all (rollout.py:370-371): ONE shared empty
# list instead of a new one per slot per cycle (nothing ever appends to a slot; slots are assigned).
_NOTHING = []


class EvalRolloutsMixin:
    def generate_eval_rollouts(self, n):
        """`for _ in range(n): evaluator.generate_rollouts()` (train.py:156-158): same draws, same statistics, same
        order.  On the batched path of an evaluator the n rollouts are ENQUEUED back to back -- each keeps its flags and its
        Q sum in a pinned slot of its own -- and waited for once: round 3 waited for every rollout's flags before the next
        one was even enqueued (10 round trips of host latency per epoch).  Nothing an evaluation rollout draws
        depends on the previous one's outcome (uniform task probabilities, rollout.py:187-189)."""
        if not (self.batched and self.eval and n > 1):
            for _ in range(n):
                self.generate_rollouts()
            return
        if self._eval_slots_ok():
            return self._generate_eval_rollouts_slots(n)
        finish = [self._generate_rollouts_batched(defer=k) for k in range(n)]
        for fin in finish:
            fin()

    EVAL_SLOTS = 1024        # env slots per launch of the slot form below (256 envs: 4 rollouts side by side)

    def _eval_slots_ok(self):
        """The n evaluation rollouts as a few launches of up to EVAL_SLOTS env slots apply: one policy acting through the
        whole-rollout launch (DDPG.act_rollout), at least two rollouts to a launch."""
        pol = self.policy
        return (not isinstance(pol, (list, tuple)) and hasattr(pol, 'act_rollout') and hasattr(pol, 'can_act_and_step')
                and pol.can_act_and_step(self.benv, self.compute_Q) and 2 * self.benv.n <= self.EVAL_SLOTS
                and hasattr(self.make_env, 'make_batched'))

    def _generate_eval_rollouts_slots(self, n):
        """`for _ in range(n): evaluator.generate_rollouts()` (train.py:156-158) with SEVERAL rollouts to a launch: rollout k
        of env e is slot k' x n_envs + e of a second batched env (envs.BatchedSyntheticArm wrap = n_envs: the same env ids,
        every slot at the episode that env would be at in its k-th next rollout), up to EVAL_SLOTS slots per launch -- ten
        rollouts of 256 envs are three launches (4 + 4 + 2 rollouts) instead of ten: the weights-resident rollout kernel keeps
        64 of the 256 CUs busy for 0.33 ms per rollout, the streaming kernel all of them for 0.45 ms per FOUR.  Same draws in
        the same order, same episodes, same bookkeeping rollout by rollout; test/mean_Q is a mean taken in another order
        (1e-6)."""
        self.settle()
        if hasattr(self.policy, 'settle'):
            self.policy.settle()
        env, T = self.benv, self.T
        nB, used = env.n, self._nloc
        per = max(2, min(n, self.EVAL_SLOTS // nB))               # rollouts per launch
        big = self.__dict__.get('_eval_env')
        if big is None or big.n != per * nB:
            big = self.make_env.make_batched(per * nB, env_id0=env.env_id0, wrap=nB)
            big.seed(env._seed)
            self._eval_env = big
            self._eval_slot_k = torch.arange(per, dtype=torch.int32, device=big.device).repeat_interleave(nB)
            self._eval_q_pins = []
        base = self._episodes_started()
        launches = []
        for j, k0 in enumerate(range(0, n, per)):
            R = min(per, n - k0)
            self._decide_exploit()
            tasks = np.zeros(per * nB, np.int64)
            goals = np.zeros([per * nB, 3], np.float32)
            lists = []
            for kk in range(R):                                   # the draws of rollout k0 + kk, in the order of the loop
                tk, gl = self._draw_tasks_goals()
                tasks[kk * nB:kk * nB + used] = tk
                goals[kk * nB:kk * nB + used] = gl
                lists.append(tk.tolist())
            big.episode.copy_(self._eval_slot_k + (base + k0))    # slot (kk, e): env e at its (k0 + kk)-th next episode
            big.reset_all(tasks, goals, launch=False)
            big._reset_pending = True
            if self.V > 1:
                self.policy.act_rollout(big, T, noise_eps=0., random_eps=0., use_target_net=self.use_target_net,
                                        exploit=self._exploit_v, evaluation=True)
            else:
                self.policy.act_rollout(big, T, noise_eps=0., random_eps=0., use_target_net=self.use_target_net,
                                        evaluation=True)
            q_pin = None
            if self.compute_Q:
                q = self.policy.rollout_q_sum(big, T, use_target_net=self.use_target_net, rollouts=(per, nB, used, R))
                while len(self._eval_q_pins) <= j:
                    self._eval_q_pins.append(torch.zeros(per, dtype=torch.float32).pin_memory())
                q_pin = self._eval_q_pins[j]
                q_pin.copy_(q, non_blocking=True)
            big.request_flags(slot=j)
            launches.append((j, R, lists, q_pin))
        # the envs themselves went through n episodes
        env.episode.add_(n)
        self._ep_host = base + n
        self.count += used * n
        for (j, R, lists, q_pin) in launches:
            big._flags_ready.synchronize()
            host = big._flags_slots[j].numpy()
            if host[big.n] != 0:
                # (a NaN observation in one of the slots: the sync path would generate such a rollout again, rollout.py:268-271)
                raise RuntimeError('evaluation rollouts produced NaN observations: the policy has diverged')
            for kk in range(R):
                successful = host[kk * nB:kk * nB + used].astype(np.float64)
                self.tasks = [_NOTHING] * self.nb_goals_per_rollout
                self.tasks[self._slot0:self._slot0 + used] = lists[kk]
                self.goals = [_NOTHING] * self.nb_goals_per_rollout
                self._finish_rollout(successful, successful - 1.0,
                                     float(q_pin[kk]) / T if self.compute_Q else None, lists[kk], None)

    def _episodes_started(self):
        """Episodes every env of the batched env has started (= the rollouts generated so far): a host mirror of
        benv.episode -- reading the device counter would wait for everything enqueued."""
        if self.__dict__.get('_ep_host') is None:
            self._ep_host = int(self.benv.episode[0])            # (once; after a checkpoint was loaded)
        return self._ep_host

    def _draw_tasks_goals(self, experts=False):
        """Task / goal draws for all envs of this process at once (vectorised form of rollout.py:120,129)."""
        B = self._nloc
        if self.V > 1:
            # every virtual rank draws for its own envs from its own stream; a rank that exploits draws its tasks from
            # the uniform distribution (rollout.py:184-186)
            per = self.rollout_batch_size
            uni = 1 / self.nb_tasks * np.ones([self.nb_tasks])
            tasks = np.concatenate([self._rng(v).choice(range(self.nb_tasks), size=per,
                                                        p=uni if (self._exploit_v[v] or self.eval) else self.p)
                                    for v in range(self.V)])
            goals = np.concatenate([self._rng(v).uniform(-1, 1, (per, 3)) for v in range(self.V)]).astype(np.float32)
        else:
            tasks = np.random.choice(range(self.nb_tasks), p=self.p, size=B)
        if experts:
            # the draws are i.i.d., so any order of the envs is the same distribution: sorted by task, every expert's
            # envs are one contiguous row range of the batched env
            tasks = np.sort(tasks)
        if self.V > 1:
            pass
        elif self.goal_selection == 'active' and not self.eval:
            # SAGG-RIAC goals live in goal space; reset_task_goal(directly=True) (rollout.py:143) = raw draw x 2 here
            goals = np.stack([2.0 * self.goal_selectors[int(ta)].sample_goal() for ta in tasks]).astype(np.float32)
        else:
            goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
        return tasks, goals

    def _expert_steps(self, env, tasks, q_sum):
        """The T acting steps with one expert per task (rollout.py:212-224: policy[task_of_env].get_actions per env):
        expert j acts on the contiguous rows of the envs that drew task j."""
        B = len(tasks)                                               # (virtual ranks: the envs of all of them)
        bounds = np.searchsorted(tasks, np.arange(self.nb_tasks + 1))
        u_all = torch.zeros([env.n, self.dims['u']], dtype=torch.float32, device=env.device)   # (idle padding envs: no action)
        noise_eps = self.noise_eps if not self.exploit else 0.
        random_eps = self.random_eps if not self.exploit else 0.
        for t in range(self.T):
            for j in range(self.nb_tasks):
                a, b = int(bounds[j]), int(bounds[j + 1])
                if a == b:
                    continue
                out = self.policy[j].get_actions(env.o[a:b], env.ag[a:b], env.g[a:b], task_descr=env.td[a:b],
                                                 compute_Q=self.compute_Q, noise_eps=noise_eps, random_eps=random_eps,
                                                 use_target_net=self.use_target_net)
                if self.compute_Q:
                    u, Q = out
                    q_sum += Q.sum() / B
                else:
                    u = out
                u_all[a:b].copy_(u)
            env.step_all(u_all, t)
        return q_sum
