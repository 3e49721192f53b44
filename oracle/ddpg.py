"""DDPG agent glue (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates the NumPy-level logic of baselines/her/ddpg.py:
  * _preprocess_og            ddpg.py:118-127
  * get_actions post-process  ddpg.py:147-156   (noise, clip, eps-greedy; RNG order kept)
  * store_episode routing     ddpg.py:174-197   (per-task activity test, buffer[task+1])
  * normaliser update         ddpg.py:207-223
  * sample_batch              ddpg.py:251-360   (CP-proportional multi-buffer mix + shuffle)
  * train / update_target_net ddpg.py:368-379, 459-462
on top of oracle.networks (TF graph restated from source), oracle.optim, oracle.normalizer.
``allreduce_sum`` stands in for the MPI Allreduce(SUM) calls (mpi_adam.py:26, normalizer.py:86).
"""
from collections import OrderedDict
import numpy as np

from oracle.networks import DDPGMath
from oracle.normalizer import Normalizer
from oracle.optim import adam_update, polyak_update


def preprocess_og(o, ag, g, clip_obs, relative_goals=False):
    if relative_goals:                                             # ddpg.py:119-124
        g = g - ag
    o = np.clip(o, -clip_obs, clip_obs)                            # ddpg.py:125
    g = np.clip(g, -clip_obs, clip_obs)                            # ddpg.py:126
    return o, g


def action_postprocess(u, rng, noise_eps, random_eps, max_u):
    """ddpg.py:149-155 with the RNG calls in the reference's order (randn, binomial, uniform)."""
    dimu = u.shape[1]
    noise = noise_eps * max_u * rng.randn(*u.shape)                # ddpg.py:149
    u = u + noise
    u = np.clip(u, -max_u, max_u)                                  # ddpg.py:151
    b = rng.binomial(1, random_eps, u.shape[0]).reshape(-1, 1)     # ddpg.py:152
    rand_u = rng.uniform(low=-max_u, high=max_u, size=(u.shape[0], dimu))   # ddpg.py:115
    u = u + b * (rand_u - u)
    return u


def buffer_proportions(buffers_sizes, T, batch_size, task_replay, cp, eps_task):
    """ddpg.py:256-286 ('curious', multi-buffer).  Returns int proportions[nb_tasks+1]."""
    buffers_sizes = np.asarray(buffers_sizes)
    nb1 = buffers_sizes.shape[0]
    proportions = np.zeros([nb1])
    if buffers_sizes[1:].sum() < T:                                # ddpg.py:260-263
        ind_valid = np.array([0])
        n_valid = 1
        proportions = buffers_sizes / buffers_sizes.sum() * batch_size
    else:
        ind_valid = np.argwhere(buffers_sizes[1:] > 0).reshape(-1)  # ddpg.py:265-267
        n_valid = len(ind_valid)
        if task_replay == 'replay_task_random_buffer':
            proba = 1 / ind_valid.size * np.ones([n_valid])         # ddpg.py:271
        elif task_replay == 'replay_task_cp_buffer':
            CP = np.asarray(cp)[ind_valid]                          # ddpg.py:273
            if CP.sum() == 0:
                proba = (1 / n_valid) * np.ones([n_valid])
            else:
                proba = eps_task * (1 / n_valid) * np.ones([n_valid]) + (1 - eps_task) * CP / CP.sum()
            proba[-1] = 1 - proba[:-1].sum()                        # ddpg.py:279
        else:
            raise NotImplementedError(task_replay)
        proportions[ind_valid + 1] = proba * batch_size             # ddpg.py:280
    proportions = proportions.astype(int)                           # ddpg.py:282
    remain = batch_size - proportions.sum()
    for i in range(remain):                                         # ddpg.py:284-285
        proportions[ind_valid[i % n_valid] + 1] += 1
    return proportions


def expert_proportions(buffers_sizes, batch_size, t_id):
    """ddpg.py:303-318 (task_experts, replay_current_task_buffer)."""
    buffers_sizes = np.asarray(buffers_sizes)
    ind_valid = np.argwhere(buffers_sizes > 0).reshape(-1)
    n_valid = len(ind_valid)
    proportions = np.zeros([buffers_sizes.shape[0]])
    if buffers_sizes[t_id + 1] > 0:
        proportions[t_id + 1] = 1
    else:
        proportions[ind_valid] = 1 / len(ind_valid)
    proportions *= batch_size
    proportions = proportions.astype(int)
    remain = batch_size - proportions.sum()
    for i in range(remain):
        proportions[ind_valid[i % n_valid]] += 1
    return proportions


def active_tasks_of(change_last, tasks_ag_id, tasks_g_id):
    """ddpg.py:179-184: tasks whose outcome moved in the episode (only j<5 routed when nb_tasks>=5)."""
    nb = len(tasks_g_id)
    act = []
    for j in range(nb):
        if any(change_last[tasks_ag_id[j][:len(tasks_g_id[j])]]):
            if nb < 5 or j < 5:
                act.append(j)
    return act


STAGE_KEYS = ['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r']   # ddpg.py:75-83 (sorted + _2 + r)


class OracleDDPG:
    """CPU DDPG agent for structure='curious' with per-task buffers (the BASELINE configs)."""

    def __init__(self, dims, T, buffers, sample_transitions, tasks_ag_id, tasks_g_id, *,
                 hidden=256, layers=3, polyak=0.95, batch_size=256, Q_lr=1e-3, pi_lr=1e-3,
                 norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1., clip_obs=200.,
                 clip_return=None, gamma=None, task_replay='replay_task_cp_buffer', eps_task=0.4,
                 relative_goals=False, structure='curious', t_id=None, rng=None, weight_rng=None,
                 allreduce_sum=None, comm_size=1, dtype=np.float32):
        self.dims, self.T = dims, T
        self.dimo, self.dimg, self.dimag, self.dimu = dims['o'], dims['g'], dims['ag'], dims['u']
        self.dimtd = dims['task_descr']
        self.buffer = buffers
        if isinstance(self.buffer, list) and len(self.buffer) > 5:   # ddpg.py:106-110
            for i in range(6, len(self.buffer)):
                self.buffer[i] = self.buffer[5]
        self.sample_transitions = sample_transitions
        self.tasks_ag_id, self.tasks_g_id = tasks_ag_id, tasks_g_id
        self.nb_tasks = len(tasks_g_id)
        self.gamma = (1. - 1. / T) if gamma is None else gamma        # config.py:125
        self.clip_return = (1. / (1. - self.gamma)) if clip_return is None else clip_return
        self.polyak, self.batch_size = polyak, batch_size
        self.Q_lr, self.pi_lr, self.max_u = Q_lr, pi_lr, max_u
        self.clip_obs, self.relative_goals = clip_obs, relative_goals
        self.task_replay, self.eps_task, self.structure, self.t_id = task_replay, eps_task, structure, t_id
        self.rng = np.random if rng is None else rng
        self.math = DDPGMath(self.dimo, self.dimg, self.dimu, self.dimtd, hidden, layers, max_u,
                             self.gamma, self.clip_return, True, action_l2, True, dtype)
        wr = np.random.RandomState(0) if weight_rng is None else weight_rng
        self.theta = self.math.init(wr)
        self.theta_target = self.theta.copy()                        # ddpg.py:459-460
        P = self.theta.shape[0]
        self.m = np.zeros(P, np.float32)
        self.v = np.zeros(P, np.float32)
        self.t_Q = 0
        self.t_pi = 0
        self._allreduce_sum = allreduce_sum
        ar = None if allreduce_sum is None else (lambda x: allreduce_sum(x.copy()))
        self.o_stats = Normalizer(self.dimo, norm_eps, norm_clip, ar, comm_size)
        self.g_stats = Normalizer(self.dimg, norm_eps, norm_clip, ar, comm_size)
        self.cp = np.zeros(self.nb_tasks)
        self.proportions = None

    # ------------------------------------------------------------------ acting
    def get_actions(self, o, ag, g, task_descr=None, noise_eps=0., random_eps=0.,
                    use_target_net=False, compute_Q=False):
        o, g = preprocess_og(o, ag, g, self.clip_obs, self.relative_goals)
        theta = self.theta_target if use_target_net else self.theta
        Qp, pip = self.math.split(theta)
        dt = self.math.dtype
        o2 = o.reshape(-1, self.dimo).astype(dt)
        g2 = g.reshape(-1, self.dimg).astype(dt)
        td = task_descr.reshape(-1, self.dimtd).astype(dt)
        pi, _, _ = self.math.actor(pip, o2, td, g2)
        ret = [pi]
        if compute_Q:
            Q, _ = self.math.critic(Qp, o2, td, g2, pi / dt(self.max_u))
            ret.append(Q)
        u = action_postprocess(ret[0], self.rng, noise_eps, random_eps, self.max_u)
        if u.shape[0] == 1:
            u = u[0]
        ret[0] = u.copy()
        return ret[0] if len(ret) == 1 else ret

    # ------------------------------------------------------------------ storing
    def store_episode(self, episode_batch, cp, n_ep, update_stats=True):
        batch_size = episode_batch['ag'].shape[0]
        self.cp = cp
        self.n_episodes = n_ep
        multi = ('buffer' in self.task_replay) or self.task_replay == 'hand_designed'
        for b in range(batch_size):
            act = active_tasks_of(episode_batch['change'][b, -1], self.tasks_ag_id, self.tasks_g_id)
            ep = {k: v[b].reshape([1, v.shape[1], v.shape[2]]) for k, v in episode_batch.items()}
            if multi:
                for task in act:                                     # ddpg.py:194-195
                    self.buffer[task + 1].store_episode(ep)
            else:
                self.buffer.store_episode(ep)                        # ddpg.py:196-197
        if update_stats:
            eb = dict(episode_batch)
            eb['o_2'] = eb['o'][:, 1:, :]
            eb['ag_2'] = eb['ag'][:, 1:, :]
            n = eb['u'].shape[0] * eb['u'].shape[1]                  # util.py:187-191
            tr = self.sample_transitions(eb, n, task_to_replay=None)  # ddpg.py:213
            o, g = preprocess_og(tr['o'], tr['ag'], tr['g'], self.clip_obs, self.relative_goals)
            self.o_stats.update(o)
            self.g_stats.update(g)
            self.o_stats.recompute_stats()
            self.g_stats.recompute_stats()

    # ------------------------------------------------------------------ sampling
    def _finish_batch(self, transitions):
        o, o_2, g = transitions['o'], transitions['o_2'], transitions['g']
        ag, ag_2 = transitions['ag'], transitions['ag_2']
        transitions['o'], transitions['g'] = preprocess_og(o, ag, g, self.clip_obs, self.relative_goals)
        transitions['o_2'], transitions['g_2'] = preprocess_og(o_2, ag_2, g, self.clip_obs, self.relative_goals)
        return [transitions[k] for k in STAGE_KEYS]                  # ddpg.py:350-358

    def sample_batch(self):
        if not (('buffer' in self.task_replay) or self.task_replay == 'hand_designed'):
            # single buffer (ddpg.py:288-299): the sampler picks the replay task itself
            proba = None
            if self.task_replay == 'replay_cp_task_transition':
                CP = np.asarray(self.cp, dtype=np.float64).copy()
                if CP.sum() == 0:
                    proba = (1 / self.nb_tasks) * np.ones([self.nb_tasks])
                else:
                    proba = self.eps_task * (1 / self.nb_tasks) * np.ones([self.nb_tasks]) + \
                        (1 - self.eps_task) * CP / CP.sum()
                proba[-1] = 1 - proba[:-1].sum()
            return self._finish_batch(self.buffer.sample(self.batch_size, task_to_replay=None, cp_proba=proba))
        sizes = np.array([self.buffer[i].current_size * self.T for i in range(self.nb_tasks + 1)])
        if self.structure == 'curious':
            self.proportions = buffer_proportions(sizes, self.T, self.batch_size, self.task_replay,
                                                  self.cp, self.eps_task)
        else:
            self.proportions = expert_proportions(sizes, self.batch_size, self.t_id)
        assert self.proportions.sum() == self.batch_size           # ddpg.py:323
        trans = []
        for i in range(self.nb_tasks + 1):                           # ddpg.py:327-336
            if self.proportions[i] > 0:
                if self.structure == 'curious':
                    ttr = i - 1 if i > 0 else None
                else:
                    ttr = self.t_id
                trans.append(self.buffer[i].sample(self.proportions[i], task_to_replay=ttr))
        shuffle_inds = np.arange(self.batch_size)
        self.rng.shuffle(shuffle_inds)                               # ddpg.py:338-339
        transitions = {}
        for key in trans[0].keys():
            tmp = np.concatenate([ts[key] for ts in trans])
            transitions[key] = tmp[shuffle_inds, :]                  # ddpg.py:345
        o, o_2, g = transitions['o'], transitions['o_2'], transitions['g']
        ag, ag_2 = transitions['ag'], transitions['ag_2']
        transitions['o'], transitions['g'] = preprocess_og(o, ag, g, self.clip_obs, self.relative_goals)
        transitions['o_2'], transitions['g_2'] = preprocess_og(o_2, ag_2, g, self.clip_obs,
                                                               self.relative_goals)
        return [transitions[k] for k in STAGE_KEYS]                  # ddpg.py:358

    # ------------------------------------------------------------------ training
    def grads(self, batch_list):
        batch = OrderedDict(zip(STAGE_KEYS, batch_list))
        return self.math.losses_and_grads(self.theta, self.theta_target, batch)

    def train(self, batch_list=None):
        if batch_list is None:
            batch_list = self.sample_batch()
        out = self.grads(batch_list)
        Qg, pig = out['Q_grad'], out['pi_grad']
        if self._allreduce_sum is not None:                          # mpi_adam.py:26 (SUM, not mean)
            Qg = self._allreduce_sum(Qg.copy())
            pig = self._allreduce_sum(pig.copy())
        PQ = self.math.P_Q
        th, m, v, self.t_Q = adam_update(self.theta[:PQ], self.m[:PQ], self.v[:PQ], self.t_Q, Qg, self.Q_lr)
        self.theta[:PQ], self.m[:PQ], self.v[:PQ] = th, m, v
        th, m, v, self.t_pi = adam_update(self.theta[PQ:], self.m[PQ:], self.v[PQ:], self.t_pi, pig, self.pi_lr)
        self.theta[PQ:], self.m[PQ:], self.v[PQ:] = th, m, v
        return out['Q_loss'], out['Q_pi']                            # ddpg.py:237-243 ("actor_loss" is Q_pi)

    def update_target_net(self):
        self.theta_target = polyak_update(self.theta_target, self.theta, self.polyak)

    def logs(self, prefix=''):
        logs = [('stats_o/mean', np.mean(self.o_stats.mean)), ('stats_o/std', np.mean(self.o_stats.std)),
                ('stats_g/mean', np.mean(self.g_stats.mean)), ('stats_g/std', np.mean(self.g_stats.std))]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + k, v) for k, v in logs]
        return logs
