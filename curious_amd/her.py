"""HER transition samplers on the GPU.  Mirrors baselines/her/her.py (factories and call signature).

    make_sample_her_transitions(goal_replay, her_replay_k, reward_fun, task_replay='', tasks_ag_id=, tasks_g_id=)
    make_sample_multi_task_her_transitions(goal_replay, her_replay_k, task_replay, reward_fun, tasks_ag_id=, tasks_g_id=)
        -> fn(episode_batch, batch_size_in_transitions, task_to_replay=None, cp_proba=None) -> {key: [B, dim]}

The random draws consume the NumPy legacy global stream in exactly the reference's order (her.py:108-116, and the
per-sample np.random.choice of her.py:139,142), so a seeded run selects the same episodes, time steps, HER mask and
future offsets as the reference; everything after the draws (index math, gather, goal/task relabel, reward) runs in
curious_her_sample.  Rewards: a `reward_fun` carrying `.spec = {'kind': 'sparse_l2', 'eps': e}` (curious_amd.envs) is
evaluated inside the kernel.  Any other callable -- the closure over env.compute_reward of config.py:158-159 for a real
gym_flowers environment -- is evaluated ON THE HOST for every sampled batch (her.py:166-176: download ag_2 / g /
task_descr / info_*, call it, upload r): the parity-audit mode for real environments, one PCIe round trip per batch, and
not available to the device-resident update loop.
"""
from collections import OrderedDict

import numpy as np
import torch

from curious_amd import _lib, ops
from curious_amd.replay_buffer import EpisodeViews, as_records


class TransitionBatch(OrderedDict):
    """{key: GPU view [B, dim]} cut from one packed batch tensor (`.packed`, layout `.layout`)."""
    packed = None
    layout = None


def _reward_spec(reward_fun):
    """(kernel threshold, host callable or None)."""
    spec = getattr(reward_fun, 'spec', None)
    if spec:
        if spec.get('kind') != 'sparse_l2':
            raise NotImplementedError("reward spec %r: the HER kernel implements {'kind': 'sparse_l2', 'eps': ...}; "
                                      "pass a plain callable to have the reward evaluated on the host" % (spec,))
        return float(spec['eps']), None
    if not callable(reward_fun):
        raise TypeError('reward_fun must be callable (config.py:158-159) or carry a .spec')
    return 0.0, reward_fun


def upload_plan(n, ep, t, u_her, u_off, buf=None, ttr=None, out_row=None):
    """One host->device copy for the whole sample plan; returns a _lib.SamplePlan.
    Host block: [u_her f64 n | u_off f64 n | ep i32 n | t | buf | task_to_replay | out_row]."""
    host = np.empty(n * 36, np.uint8)
    f64 = host[:16 * n].view(np.float64)
    i32 = host[16 * n:].view(np.int32)
    f64[:n] = u_her
    f64[n:] = u_off
    i32[0 * n:1 * n] = ep
    i32[1 * n:2 * n] = t
    i32[2 * n:3 * n] = 0 if buf is None else buf
    i32[3 * n:4 * n] = -1 if ttr is None else ttr
    i32[4 * n:5 * n] = np.arange(n) if out_row is None else out_row
    dev = torch.from_numpy(host).cuda(non_blocking=True)
    d64 = dev[:16 * n].view(torch.float64)
    d32 = dev[16 * n:].view(torch.int32)
    return ops.make_plan(d32[0:n], d32[n:2 * n], d64[:n], d64[n:], buf=d32[2 * n:3 * n],
                         task_to_replay=d32[3 * n:4 * n], out_row=d32[4 * n:5 * n])


class HerSampler:
    """Callable with the reference's sampler signature; also exposes the pieces DDPG fuses across buffers."""

    def __init__(self, goal_replay, her_replay_k, task_replay, reward_fun, tasks_ag_id, tasks_g_id, flat):
        self.future_p = 1 - (1. / (1 + her_replay_k)) if goal_replay == 'her' else 0      # her.py:86-89
        self.task_replay = task_replay
        self.flat = flat
        self.nb_tasks = len(tasks_ag_id)
        self.tasks_ag_id, self.tasks_g_id = tasks_ag_id, tasks_g_id
        self.tasks = _lib.make_tasks(tasks_ag_id, tasks_g_id)
        self.reward_eps, self.host_reward = _reward_spec(reward_fun)
        self.reward_fun = reward_fun
        self.multiple_buffers = ('buffer' in task_replay) or task_replay == 'hand_designed'   # her.py:94-97
        if flat:
            self.mode = _lib.RELABEL_FLAT
        elif task_replay == 'replay_current_task_transition':
            self.mode = _lib.RELABEL_CURRENT_TASK
        elif self.multiple_buffers:
            self.mode = _lib.RELABEL_BUFFER_TASK
        else:
            self.mode = _lib.RELABEL_GIVEN_TASK

    # ---- the reference's RNG consumption, in order
    def draw(self, n_episodes, T, batch_size, cp_proba=None):
        ep = np.random.randint(0, n_episodes, batch_size)            # her.py:108
        t = np.random.randint(T, size=batch_size)                    # her.py:109
        u_her = np.random.uniform(size=batch_size)                   # her.py:115
        u_off = np.random.uniform(size=batch_size)                   # her.py:116
        given = None
        if self.mode == _lib.RELABEL_GIVEN_TASK:
            given = np.full(batch_size, -1, np.int32)
            for i in np.where(u_her < self.future_p)[0]:             # loop of her.py:129-142
                if self.task_replay == 'replay_random_task_transition':
                    given[i] = np.random.choice(range(self.nb_tasks))
                elif self.task_replay == 'replay_cp_task_transition':
                    given[i] = np.random.choice(range(self.nb_tasks), p=cp_proba)
                else:
                    raise NotImplementedError('task_replay = %r' % self.task_replay)
        return ep, t, u_her, u_off, given

    def params(self, clip_obs=np.inf, relative_goals=False):
        key = (float(clip_obs), bool(relative_goals), float(self.future_p), self.reward_eps, self.mode, bool(self.flat))
        if getattr(self, '_params_cache', (None, None))[0] == key:
            return self._params_cache[1]
        P = _lib.SampleParams()
        P.future_p = float(self.future_p)
        P.reward_eps = self.reward_eps
        P.clip_obs = float(clip_obs)
        P.relative_goals = int(bool(relative_goals))
        P.relabel_mode = self.mode
        P.flat_reward = int(self.flat)
        self._params_cache = (key, P)
        return P

    def apply_host_reward(self, batch, layout):
        """her.py:166-176 on the host: r = reward_fun(ag_2, g, task_descr, info) for a staged batch (float64 like the
        reference's buffers), written into the batch's r column.  `batch` must hold the un-clipped relabelled goals."""
        if self.host_reward is None:
            return
        info_keys = [k for k in layout.batch_cols if k.startswith('info_')]
        names = ['ag_2', 'g'] + (['task_descr'] if 'task_descr' in layout.batch_cols and not self.flat else [])
        views = layout.batch_views(batch, names + info_keys)
        host = {k: v.cpu().numpy().astype(np.float64) for k, v in views.items()}
        kw = {k: host[k] for k in names}
        kw['info'] = {k.replace('info_', ''): host[k] for k in info_keys}
        r = np.asarray(self.host_reward(**kw), dtype=np.float32).reshape(batch.shape[0], 1)
        off = layout.batch_cols['r'][0]
        batch[:, off:off + 1].copy_(torch.from_numpy(r))

    def __call__(self, episode_batch, batch_size_in_transitions, task_to_replay=None, cp_proba=None):
        layout = episode_batch.layout if isinstance(episode_batch, EpisodeViews) else None
        if layout is None:
            from curious_amd.layout import RecordLayout
            T = episode_batch['u'].shape[1]
            shapes = {k: (v.shape[1], v.shape[2]) for k, v in episode_batch.items() if k not in ('o_2', 'ag_2')}
            layout = RecordLayout(shapes, T)
        records = as_records(episode_batch, layout)
        E, T, B = records.shape[0], layout.T, int(batch_size_in_transitions)
        ep, t, u_her, u_off, given = self.draw(E, T, B, cp_proba)
        if given is not None:
            ttr = given
        else:
            ttr = np.full(B, -1 if task_to_replay is None else int(task_to_replay), np.int32)
        plan = upload_plan(B, ep, t, u_her, u_off, ttr=ttr)
        batch = torch.empty([B, layout.batch_stride], dtype=torch.float32, device=records.device)
        ops.her_sample(records, 0, layout, self.tasks, self.params(), B, batch, plan=plan)
        self.apply_host_reward(batch, layout)
        out = TransitionBatch()
        keys = [k for k in episode_batch.keys()] + ['r']
        if 'o_2' not in episode_batch:
            keys += ['o_2', 'ag_2']
        for k, v in layout.batch_views(batch, keys).items():
            out[k] = v
        out.packed, out.layout = batch, layout
        assert out['u'].shape[0] == batch_size_in_transitions        # her.py:180
        return out


def make_sample_multi_task_her_transitions(goal_replay, her_replay_k, task_replay, reward_fun, tasks_ag_id=None,
                                           tasks_g_id=None):
    """Sampler for structure 'curious' / 'task_experts' (her.py:72-185)."""
    return HerSampler(goal_replay, her_replay_k, task_replay, reward_fun, tasks_ag_id, tasks_g_id, flat=False)


def make_sample_her_transitions(goal_replay, her_replay_k, reward_fun, task_replay='', tasks_ag_id=None,
                                tasks_g_id=None):
    """Sampler for structure 'flat' (her.py:5-68)."""
    return HerSampler(goal_replay, her_replay_k, task_replay, reward_fun, tasks_ag_id, tasks_g_id, flat=True)
