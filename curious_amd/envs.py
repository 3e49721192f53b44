"""Synthetic MultiTaskFetchArm stand-ins resident on the GPU.

The reference trains on MuJoCo environments from the un-vendored `gym_flowers` package (config.py:2-3,112-122);
neither MuJoCo nor gym_flowers exists here, so the rollout half of the hot path is exercised on a small
deterministic point-mass arm with the same interface, dimensions and task/goal tables (specification:
oracle/env.py; kernels: curious_amd/csrc/env.hip).  Two front-ends share those kernels:

  BatchedSyntheticArm   n environments stepped by ONE kernel launch, episode records written in place in the
                        staging block the replay store reads (used by RolloutWorker's batched path);
  SyntheticArmEnv       a single environment answering the gym-style protocol the reference consumes
                        (SURVEY 8b "Env protocol consumed"), for code that insists on a Python list of envs.
"""
import numpy as np
import torch

from curious_amd import _lib, ops
from curious_amd.layout import RecordLayout
from curious_amd.replay_buffer import EpisodeViews

ENV_CONFIGS = {
    # name: (nb_tasks, dimo, T)  -- dimo / T are assumptions, the real values come from gym_flowers
    'MultiTaskFetchArm4-v5': (4, 40, 50),
    'MultiTaskFetchArm8-v5': (8, 52, 50),
}
REWARD_EPS = 0.05


class _Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


class _DictSpace:
    def __init__(self, spaces):
        self.spaces = spaces


class ArmSpec:
    """Static description shared by both front-ends."""

    def __init__(self, name):
        if name not in ENV_CONFIGS:
            raise KeyError('unknown synthetic env %r (have %s)' % (name, sorted(ENV_CONFIGS)))
        self.name = name
        self.nb_tasks, self.dimo, self.T = ENV_CONFIGS[name]
        self.dimg = self.dimag = 3 * self.nb_tasks
        self.dimu = 4
        self.tasks_g_id = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(self.nb_tasks)]
        self.tasks_ag_id = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(self.nb_tasks)]
        self.info = {'is_success': 0.0}
        self._max_episode_steps = self.T
        self.observation_space = _DictSpace(dict(observation=_Space([self.dimo]), achieved_goal=_Space([self.dimag]),
                                                 desired_goal=_Space([self.dimg])))
        self.action_space = _Space([self.dimu])
        self.reward_spec = dict(kind='sparse_l2', eps=REWARD_EPS)
        self.flat = False

    def set_flat_env(self):
        """Single-task view (config.py:116-117, rollout.py:93-95): goals span every slot, the reward is the sparse L2
        threshold over the whole goal vector."""
        self.flat = True

    def compute_reward(self, achieved_goal, goal, task_descr=None, info=None):
        """The env's own reward (what config.py:158-159 wraps): -1 while the L2 distance between the achieved and the
        desired goal on the task's slots (all slots for the flat view) exceeds reward_spec['eps'], else 0; float64,
        shape [n, 1].  Same rule as the HER kernel evaluates on the GPU."""
        ag = np.asarray(achieved_goal, dtype=np.float64).reshape(-1, self.dimag)
        g = np.asarray(goal, dtype=np.float64).reshape(-1, self.dimg)
        n = ag.shape[0]
        if self.flat or task_descr is None:
            ids_g = [i for ids in self.tasks_g_id for i in ids]
            ids_ag = [i for ids in self.tasks_ag_id for i in ids]
            sel_g, sel_ag = g[:, ids_g], ag[:, ids_ag]
        else:
            task = np.argmax(np.asarray(task_descr).reshape(n, self.nb_tasks), axis=1)
            sel_g = np.stack([g[i, self.tasks_g_id[task[i]]] for i in range(n)])
            sel_ag = np.stack([ag[i, self.tasks_ag_id[task[i]][:len(self.tasks_g_id[task[i]])]] for i in range(n)])
        d2 = np.zeros(n)
        for k in range(sel_g.shape[1]):                       # sequential float64 sum, like the kernel
            d = sel_ag[:, k] - sel_g[:, k]
            d2 = d2 + d * d
        return -(np.sqrt(d2) > self.reward_spec['eps']).astype(np.float64).reshape(n, 1)

    def _compute_goal(self, g, task, eval=False):
        """Goal-space image of a raw draw g in [-1, 1]^3 for `task` (the protocol of rollout.py:85-86,135)."""
        goal = np.zeros(self.dimg, np.float32)
        goal[self.tasks_g_id[task]] = np.float32(0.5) * np.asarray(g, dtype=np.float32)
        mask = np.zeros(self.nb_tasks, np.float32)
        mask[task] = 1
        return goal, mask

    def dims(self):
        return dict(o=self.dimo, u=self.dimu, g=self.dimg, ag=self.dimag, task_descr=self.nb_tasks,
                    info_is_success=1)

    def buffer_shapes(self):
        T = self.T                                                   # config.py:198-206
        return dict(o=(T + 1, self.dimo), u=(T, self.dimu), g=(T, self.dimg), ag=(T + 1, self.dimag),
                    info_is_success=(T, 1), task_descr=(T, self.nb_tasks), change=(T, self.dimag))


def sparse_reward_fun(spec):
    """reward_fun(ag_2, g, task_descr, info) placeholder carrying the kernel-side reward description.  The HER
    kernel evaluates the reward on the GPU; this callable exists so that the reference's plumbing
    (config.py:158-167) has something to pass around, and refuses to be evaluated on the host."""
    def reward_fun(ag_2=None, g=None, task_descr=None, info=None):
        raise NotImplementedError('the sparse reward is evaluated inside curious_her_sample on the GPU')
    reward_fun.spec = dict(spec)
    return reward_fun


class ResidentRolloutVoid(_lib.CuriousHipError):
    """The weights-resident rollout launch (policy_resident_kernel) reported itself void: a workgroup never got an answer
    from a peer of its group, i.e. the launch was not fully resident (shared device, CU mask, partition).  Nothing of the
    rollout may be used; the streaming kernel (option 'resident' = 0) computes the same numbers."""


class BatchedSyntheticArm(ArmSpec):
    def __init__(self, name, n, seed=0, env_id0=0, T=None, pad_to=1, wrap=0):
        """pad_to: the batch is filled up to a multiple of it with idle envs (they step like any other env, nobody
        reads their episodes): the one-launch rollout kernels take whole groups of 4 envs, and 19 virtual ranks x 2
        rollouts (the reference's regime, readme.md:16) are 38.  n_used = the envs that count, n = the envs launched.
        wrap > 0: a batch of SLOTS (curious_env_cfg_t.wrap) -- slot i is env env_id0 + i % wrap at the episode episode[i]:
        several rollouts of the same envs side by side (RolloutWorker.generate_eval_rollouts)."""
        super().__init__(name)
        if T is not None:
            self.T = self._max_episode_steps = int(T)
        self.n_used = int(n)
        n = (int(n) + pad_to - 1) // pad_to * pad_to
        self.n, self.env_id0 = int(n), int(env_id0)
        self._seed = int(seed)
        self._wrap = int(wrap)
        dev = torch.device('cuda', torch.cuda.current_device())
        self.device = dev
        self.layout = RecordLayout(self.buffer_shapes(), self.T)
        self.o = torch.zeros([n, self.dimo], device=dev)
        self.ag = torch.zeros([n, self.dimag], device=dev)
        self.g = torch.zeros([n, self.dimg], device=dev)
        self.td = torch.zeros([n, self.nb_tasks], device=dev)
        self.staging = torch.zeros([n, self.T + 1, self.layout.row_stride], device=dev)
        self.episode = torch.zeros(n, dtype=torch.int32, device=dev)     # episodes started so far, per env
        # tasks (int32) and raw goals of a rollout live in one device block: ONE H2D copy per reset
        self._tg_dev = torch.zeros(n * 4, dtype=torch.float32, device=dev)
        self.tasks = self._tg_dev[:n].view(torch.int32)
        self.tasks_host = np.zeros(n, np.int32)
        self.goals_host = np.zeros([n, 3], np.float32)
        # pinned staging for the per-rollout task / goal upload: the copy is enqueued without blocking the host,
        # so the next rollout can be queued behind the previous cycle's updates
        # (a small ring: evaluation rollouts are enqueued back to back without a host wait in between, so the block of
        #  reset k must not be refilled before its copy has run; a block is reused only behind its own event)
        self._pins = [torch.empty(n * 4, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._pin_events = [None] * len(self._pins)
        self._pin_k = 0
        self._goals_dev = self._tg_dev[n:].view(n, 3)
        self._cfg = ops.make_env_cfg(self.nb_tasks, self.dimo, self.T, self._seed, self._wrap)
        # rollout flags written by the last env step: is_success per env + one "an observation is NaN" word
        self.flags = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self._flags_pin = torch.zeros(n + 1, dtype=torch.float32).pin_memory()
        self._flags_ready = torch.cuda.Event()

    @property
    def unwrapped(self):
        return self

    def seed(self, seed):
        self._seed = int(seed)
        self._cfg = ops.make_env_cfg(self.nb_tasks, self.dimo, self.T, self._seed, self._wrap)
        self.episode.zero_()

    def reset_all(self, tasks, goals_raw, launch=True):
        """tasks[n] int, goals_raw[n,3] in [-1,1] (rollout.py:120-143 for every env at once).  launch=False: only the
        upload of the draws; the reset launch itself heads the captured rollout (launch_reset, DDPG.act_rollout)."""
        self.tasks_host[:self.n_used] = tasks                        # (idle padding envs: task 0, goal 0)
        self.goals_host[:self.n_used] = goals_raw
        n = self.n
        k = self._pin_k
        self._pin_k = (k + 1) % len(self._pins)
        if self._pin_events[k] is not None:
            self._pin_events[k].synchronize()                        # (its previous copy ran long ago, as a rule)
        pin = self._pins[k]
        pin[:n].view(torch.int32).copy_(torch.from_numpy(self.tasks_host))
        pin[n:].view(n, 3).copy_(torch.from_numpy(self.goals_host))
        self._tg_dev.copy_(pin, non_blocking=True)
        if self._pin_events[k] is None:
            self._pin_events[k] = torch.cuda.Event()
        self._pin_events[k].record()
        if launch:
            self.launch_reset()

    def launch_reset(self, counter=None, delta=0):
        ops.env_reset(self._cfg, self.layout, self.env_id0, self.episode, self.tasks, self._goals_dev, self.n,
                      self.o, self.ag, self.g, self.td, self.staging,      # also advances self.episode on the device
                      flags=self.flags, counter=counter, delta=delta)      # and clears the NaN word of the coming rollout

    def step_all(self, u, t):
        ops.env_step(self._cfg, self.layout, self.env_id0, self.episode, self.tasks, u, t, self.n, self.o, self.ag,
                     self.g, self.td, self.staging, REWARD_EPS, flags=self.flags)

    def request_flags(self, slot=None):
        """Enqueue the D2H copy of the rollout flags (no wait).  slot: one of several pinned blocks -- rollouts that are
        enqueued back to back and waited for together (RolloutWorker.generate_eval_rollouts) each keep their own."""
        if slot is None:
            pin = self._flags_pin
        else:
            pins = self.__dict__.setdefault('_flags_slots', [])
            while len(pins) <= slot:
                pins.append(torch.zeros(self.n + 1, dtype=torch.float32).pin_memory())
            pin = pins[slot]
        pin.copy_(self.flags, non_blocking=True)
        self._flags_ready.record()

    def wait_flags(self, slot=None):
        self._flags_ready.synchronize()
        host = (self._flags_pin if slot is None else self._flags_slots[slot]).numpy()
        if host[self.n] == 2.0:
            # include/curious_hip.h, curious_policy_rollout: a workgroup of the resident-weights rollout never got an
            # answer from a peer of its group (the launch was not fully resident) -- the rollout is void
            raise ResidentRolloutVoid('curious_policy_rollout: a member of a workgroup group gave up waiting for its '
                                      "peers (the launch was not fully resident); option 'resident' = 0 selects the "
                                      'streaming kernel')
        return host[:self.n_used].astype(np.float64), bool(host[self.n] != 0)

    def fetch_flags(self):
        """(is_success of the final step per env [n], any observation NaN) -- the one D2H sync of a rollout."""
        self.request_flags()
        return self.wait_flags()

    def episode_views(self):
        # the staging block is allocated once: so are the views cut from it
        if getattr(self, '_views', None) is None:
            self._views = EpisodeViews(self.staging[:self.n_used], self.layout, with_next=False)
        return self._views

    def last_success(self):
        """is_success of the final step, [n] float32 on the GPU."""
        return self.staging[:self.n_used, self.T - 1, self.layout.off['info_is_success']]


class SyntheticArmEnv(ArmSpec):
    """Single environment with the protocol of the reference's gym_flowers envs (one-env batched arm inside)."""

    def __init__(self, name, seed=0, env_id=0):
        super().__init__(name)
        self._b = BatchedSyntheticArm(name, 1, seed=seed, env_id0=env_id)
        self.task = 0
        self.goal = np.zeros(self.dimg, np.float32)
        self._t = 0

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        self._b.seed(seed)

    def _obs(self):
        return dict(observation=self._b.o[0].cpu().numpy(), achieved_goal=self._b.ag[0].cpu().numpy(),
                    desired_goal=self._b.g[0].cpu().numpy(), mask=self._b.td[0].cpu().numpy())

    def reset(self):
        self._b.reset_all(np.array([self.task]), np.zeros([1, 3], np.float32))
        self._t = 0
        return self._obs()

    def reset_task_goal(self, goal, task=0, directly=False, eval=False):
        """Sets task and goal of the CURRENT episode (the state drawn by reset() is kept)."""
        self.task = int(task)
        raw = np.asarray(goal, dtype=np.float32) * (np.float32(2.0) if directly else np.float32(1.0))
        b = self._b
        b.tasks_host[0] = self.task
        b.tasks.copy_(torch.from_numpy(b.tasks_host))
        if self.flat:                                                # the goal spans every slot (rollout.py:93-95)
            full, mask = np.float32(0.5) * raw.reshape(self.dimg), np.zeros(self.nb_tasks, np.float32)
        else:
            full, mask = self._compute_goal(raw, self.task)
        self.goal = full
        b.g[0].copy_(torch.from_numpy(full))
        b.td[0].copy_(torch.from_numpy(mask))
        return self._obs()

    def step(self, u):
        u_d = torch.as_tensor(np.asarray(u, dtype=np.float32).reshape(1, -1)).to(self._b.device)
        t = min(self._t, self.T - 1)
        self._b.step_all(u_d, t)
        self._t += 1
        obs = self._obs()
        if self.flat:
            succ = float(self.compute_reward(obs['achieved_goal'], obs['desired_goal'])[0, 0] + 1.0)
        else:
            succ = float(self._b.staging[0, t, self._b.layout.off['info_is_success']])
        return obs, succ - 1.0, False, {'is_success': succ}

    def render(self):
        pass


class EnvFactory:
    """`make_env` of config.prepare_params (config.py:112-115): callable -> single env; .make_batched(n) -> batched."""

    def __init__(self, name):
        self.name = name
        self._count = 0

    def __call__(self):
        e = SyntheticArmEnv(self.name, env_id=self._count)
        self._count += 1
        return e

    def make_batched(self, n, env_id0=0, pad_to=1, wrap=0):
        return BatchedSyntheticArm(self.name, n, env_id0=env_id0, pad_to=pad_to, wrap=wrap)
