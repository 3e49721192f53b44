#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference (read-only) in this container.

Run here only:   python tools/gen_golden.py            (needs /root/reference)
The reference's Python never travels to the GPU box; only the vectors written by this
script do.  Fixtures hold data (seeded inputs, the random draws, expected outputs), never
reference source text.

Stub modules placed in sys.modules so that the pure-NumPy parts of the reference import:
  mpi4py (single-rank COMM_WORLD), tensorflow (permissive object; nothing TF is executed),
  gym / gym.spaces.Box, mujoco_py.MujocoException, pandas.ewma attribute.
The duck-typed environment handed to the reference RolloutWorker is oracle.env's
synthetic arm (the reference's own environment, gym_flowers, is not vendored).
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('CURIOUS_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


# ----------------------------------------------------------------------------- stubs
class _Comm:
    def Get_rank(self): return 0
    def Get_size(self): return 1
    def Allreduce(self, src, dst, op=None): dst[...] = src
    def Bcast(self, buf, root=0): pass
    def bcast(self, obj, root=0): return obj
    def scatter(self, objs, root=0): return objs[0]
    def gather(self, obj, root=0): return [obj]
    def Abort(self): raise SystemExit(1)


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        m = _Permissive(self.__name__ + '.' + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Permissive('call')


def install_stubs():
    mpi4py = types.ModuleType('mpi4py')
    MPI = types.ModuleType('mpi4py.MPI')
    MPI.COMM_WORLD = _Comm()
    MPI.SUM = 'SUM'
    mpi4py.MPI = MPI
    sys.modules['mpi4py'] = mpi4py
    sys.modules['mpi4py.MPI'] = MPI
    tf = _Permissive('tensorflow')
    sys.modules['tensorflow'] = tf
    for sub in ['contrib', 'contrib.staging', 'python', 'python.client']:
        sys.modules['tensorflow.' + sub] = getattr(tf, sub.split('.')[0]) if '.' not in sub else _Permissive(sub)
    gym = types.ModuleType('gym')
    spaces = types.ModuleType('gym.spaces')

    class Box:
        # the members SAGG-RIAC uses, with gym's semantics (private copies of the bounds, closed intervals); sample()
        # draws from the NumPy GLOBAL stream (gym keeps a private generator) so that the fixture is reproducible
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high = np.array(low, dtype=dtype), np.array(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

        def sample(self):
            return np.random.uniform(low=self.low, high=self.high).astype(self.dtype)
    spaces.Box = Box
    gym.spaces = spaces
    gym.make = lambda name: None
    sys.modules['gym'] = gym
    sys.modules['gym.spaces'] = spaces
    mj = types.ModuleType('mujoco_py')

    class MujocoException(Exception):
        pass
    mj.MujocoException = MujocoException
    sys.modules['mujoco_py'] = mj
    import pandas
    if not hasattr(pandas, 'ewma'):
        pandas.ewma = lambda *a, **k: None


# ----------------------------------------------------------------------------- synthetic data
def synth_episodes(rng, E, T, dimo, nb_tasks, dimu=4):
    """float32-valued synthetic episodes in the reference's buffer layout (SURVEY 8d cfg 2)."""
    G = AG = 3 * nb_tasks
    o = np.empty([E, T + 1, dimo], np.float32)
    o[:, 0] = rng.randn(E, dimo).astype(np.float32)
    steps = (0.01 * rng.randn(E, T, dimo)).astype(np.float32)
    # some coordinates stay frozen so that `change` is not all-True
    frozen = rng.rand(E, 1, dimo) < 0.4
    steps = np.where(frozen, np.float32(0), steps)
    for t in range(T):
        o[:, t + 1] = o[:, t] + steps[:, t]
    ag = o[:, :, :AG].copy()
    task = rng.randint(nb_tasks, size=E)
    td = np.zeros([E, T, nb_tasks], np.float32)
    td[np.arange(E), :, task] = 1
    g = np.zeros([E, T, G], np.float32)
    for e in range(E):
        sl = slice(3 * task[e], 3 * task[e] + 3)
        # goals close to the trajectory so that both rewards 0 and -1 occur
        g[e, :, sl] = ag[e, rng.randint(T + 1), sl] + (0.03 * rng.randn(3)).astype(np.float32)
    u = rng.uniform(-1, 1, [E, T, dimu]).astype(np.float32)
    change = (np.abs(ag[:, :1] - ag[:, 1:]) > 1e-3)
    succ = rng.randint(2, size=[E, T, 1]).astype(np.float32)
    return dict(o=o, u=u, g=g, ag=ag, task_descr=td, change=change, info_is_success=succ)


def to_f64_buffers(ep):
    """What ReplayBuffer holds: float64 arrays (replay_buffer.py:23) with float32 values."""
    return {k: v.astype(np.float64) for k, v in ep.items()}


def tables(nb_tasks):
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb_tasks)]
    return ids, [list(x) for x in ids]


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def flat_dict(prefix, d):
    return {prefix + k: np.asarray(v) for k, v in d.items()}


# ----------------------------------------------------------------------------- generators
def gen_her():
    from baselines.her import her as ref_her
    from oracle.reward import make_reward_fun
    cases = []
    rng = np.random.RandomState(1234)
    cfgs = [
        # name, nb_tasks, dimo, E, T, B, task_replay, goal_replay, task_to_replay, flat
        ('arm4_buffer_none', 4, 40, 23, 50, 256, 'replay_task_cp_buffer', 'her', None, False),
        ('arm4_buffer_t2', 4, 40, 23, 50, 256, 'replay_task_cp_buffer', 'her', 2, False),
        ('arm4_buffer_t0_E1', 4, 40, 1, 50, 64, 'replay_task_random_buffer', 'her', 0, False),
        ('arm8_buffer_t5', 8, 52, 17, 50, 256, 'replay_task_cp_buffer', 'her', 5, False),
        ('arm4_current', 4, 40, 11, 50, 128, 'replay_current_task_transition', 'her', None, False),
        ('arm4_random', 4, 40, 11, 50, 128, 'replay_random_task_transition', 'her', None, False),
        ('arm4_cp', 4, 40, 11, 50, 128, 'replay_cp_task_transition', 'her', None, False),
        ('arm4_noher', 4, 40, 9, 50, 96, 'replay_task_cp_buffer', 'none', 1, False),
        ('arm4_T7_ragged', 4, 19, 5, 7, 33, 'replay_task_cp_buffer', 'her', 3, False),
        ('arm4_flat', 4, 40, 13, 50, 128, '', 'her', None, True),
    ]
    out = {}
    names = []
    for i, (name, nb, dimo, E, T, B, task_replay, goal_replay, ttr, flat) in enumerate(cfgs):
        ag_ids, g_ids = tables(nb)
        ep = synth_episodes(rng, E, T, dimo, nb)
        buf = to_f64_buffers(ep)
        buf['o_2'] = buf['o'][:, 1:, :]
        buf['ag_2'] = buf['ag'][:, 1:, :]
        reward = make_reward_fun(ag_ids, g_ids)
        seed = 100 + i
        cp_proba = None
        if task_replay == 'replay_cp_task_transition':
            cp_proba = np.array([0.5, 0.2, 0.2, 0.1])
        if flat:
            fn = ref_her.make_sample_her_transitions(goal_replay, 4, reward, tasks_ag_id=ag_ids, tasks_g_id=g_ids)
            def reward_flat(ag_2, g, task_descr, info, _r=reward):
                return _r(ag_2, g, None, info)
            fn = ref_her.make_sample_her_transitions(goal_replay, 4, reward_flat, tasks_ag_id=ag_ids,
                                                     tasks_g_id=g_ids)
        else:
            fn = ref_her.make_sample_multi_task_her_transitions(goal_replay, 4, task_replay, reward,
                                                                tasks_ag_id=ag_ids, tasks_g_id=g_ids)
        np.random.seed(seed)
        tr = fn(buf, B, task_to_replay=ttr, cp_proba=cp_proba)
        # the raw draws (same legacy stream, re-drawn) for kernel-level tests
        rs = np.random.RandomState(seed)
        d_ep = rs.randint(0, E, B)
        d_t = rs.randint(T, size=B)
        d_uher = rs.uniform(size=B)
        d_uoff = rs.uniform(size=B)
        names.append(name)
        out.update(flat_dict(name + '/in/', ep))
        out.update(flat_dict(name + '/out/', tr))
        out[name + '/draw/ep'] = d_ep
        out[name + '/draw/t'] = d_t
        out[name + '/draw/u_her'] = d_uher
        out[name + '/draw/u_off'] = d_uoff
        out[name + '/cfg'] = np.array([nb, dimo, E, T, B, seed, -1 if ttr is None else ttr, int(flat)])
        out[name + '/task_replay'] = np.array(task_replay)
        out[name + '/goal_replay'] = np.array(goal_replay)
        if cp_proba is not None:
            out[name + '/cp_proba'] = cp_proba
    out['names'] = np.array(names)
    save('her', **out)


def gen_replay_buffer():
    from baselines.her import her as ref_her
    from baselines.her.replay_buffer import ReplayBuffer
    from oracle.reward import make_reward_fun
    nb, dimo, T = 4, 40, 10
    ag_ids, g_ids = tables(nb)
    reward = make_reward_fun(ag_ids, g_ids)
    fn = ref_her.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', reward,
                                                        tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, 12), ag=(T + 1, 12), task_descr=(T, nb),
                  change=(T, 12), info_is_success=(T, 1))
    rb = ReplayBuffer(shapes, T * 7, T, fn)           # capacity 7 episodes
    rng = np.random.RandomState(77)
    incs = [1, 1, 3, 1, 4, 2, 1, 3]                   # crosses the "append -> random" boundary
    np.random.seed(4242)
    out = {'incs': np.array(incs)}
    for step, inc in enumerate(incs):
        ep = synth_episodes(rng, inc, T, dimo, nb)
        ep64 = {k: v.astype(np.float64) for k, v in ep.items()}
        rb.store_episode(ep64)
        out.update(flat_dict('step%d/in/' % step, ep))
        out['step%d/current_size' % step] = np.array(rb.get_current_episode_size())
        out['step%d/n_stored' % step] = np.array(rb.get_transitions_stored())
        # buffer contents are fully defined only below current_size
        out['step%d/o_rows' % step] = rb.buffers['o'][:rb.current_size, 0, :3].copy()
    tr = rb.sample(64, task_to_replay=1)
    out.update(flat_dict('sample/', tr))
    for k in rb.buffers:
        out['final/' + k] = rb.buffers[k][:rb.current_size].copy()
    out['cfg'] = np.array([nb, dimo, T, 7, 4242])
    save('replay_buffer', **out)


def gen_queues():
    from baselines.her.queues import CompetenceQueue
    rng = np.random.RandomState(5)
    q = CompetenceQueue(window=10)
    chunks, Cs, CPs, sizes = [], [], [], []
    for i in range(40):
        n = rng.randint(0, 4)
        p = 0.2 + 0.6 * (i / 40.0)
        ch = (rng.rand(n) < p).astype(np.float64)
        q.update(ch.tolist())
        chunks.append(np.concatenate([ch, -np.ones(4 - n)]))
        Cs.append(q.C)
        CPs.append(q.CP)
        sizes.append(q.size)
    save('queues', chunks=np.array(chunks), C=np.array(Cs, dtype=np.float64), CP=np.array(CPs, dtype=np.float64),
         size=np.array(sizes), window=np.array(10))


class FakePolicy:
    """Deterministic linear policy + the reference's action post-processing (ddpg.py:147-156),
    written with np.random so it consumes the global stream like DDPG.get_actions."""

    def __init__(self, A, max_u=1.0):
        self.A = A
        self.max_u = max_u

    def get_actions(self, o, ag, g, task_descr=None, noise_eps=0., random_eps=0., use_target_net=False,
                    compute_Q=False):
        x = np.concatenate([o.reshape(len(o), -1), g.reshape(len(g), -1), task_descr.reshape(len(o), -1)], axis=1)
        u = 0.1 * np.tanh(x.astype(np.float64) @ self.A)
        # goal-directed on the gripper slots so that the Reach task succeeds now and then
        gg = g.reshape(len(g), -1)
        u[:, :3] += 4.0 * (gg[:, :3] - o.reshape(len(o), -1)[:, :3]) * (gg[:, :1] > 0)   # fails for half the goals
        Q = u.sum(axis=1, keepdims=True)
        noise = noise_eps * self.max_u * np.random.randn(*u.shape)
        u = u + noise
        u = np.clip(u, -self.max_u, self.max_u)
        u += np.random.binomial(1, random_eps, u.shape[0]).reshape(-1, 1) * (
            np.random.uniform(low=-self.max_u, high=self.max_u, size=u.shape) - u)
        if u.shape[0] == 1:
            u = u[0]
        return [u.copy(), Q] if compute_Q else u.copy()


def gen_rollout():
    from baselines.her.rollout import RolloutWorker
    from baselines import logger
    from oracle.env import SyntheticMultiTaskArm
    nb, dimo, T, B = 4, 40, 25, 3
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    rngA = np.random.RandomState(9)
    A = (0.5 * rngA.randn(dimo + 12 + nb, 4))
    out = {'A': A, 'cfg': np.array([nb, dimo, T, B])}
    for mode, (eval_, task_sel) in {'train': (False, 'active_competence_progress'),
                                    'eval': (True, 'active_competence_progress')}.items():
        counter = [0]

        def make_env():
            e = SyntheticMultiTaskArm(nb, dimo, T, seed=0, env_id=counter[0])
            counter[0] += 1
            return e
        np.random.seed(31337)
        w = RolloutWorker(make_env, FakePolicy(A), dims, logger, T=T, rollout_batch_size=B,
                          exploit=eval_, use_target_net=False, compute_Q=eval_, noise_eps=0.2,
                          random_eps=0.3, structure='curious', task_selection=task_sel,
                          goal_selection='random', queue_length=4, eval=eval_)
        w.seed(555)
        n_cycles = 80
        for c in range(n_cycles):
            ep, CP, n_ep = w.generate_rollouts()
            if c in (0, 1, n_cycles - 1):
                out.update(flat_dict('%s/cycle%d/' % (mode, c), ep))
            out['%s/CP%d' % (mode, c)] = np.asarray(CP, dtype=np.float64)
            out['%s/p%d' % (mode, c)] = np.asarray(w.p, dtype=np.float64)
            out['%s/C%d' % (mode, c)] = np.asarray(w.C, dtype=np.float64)
            out['%s/exploit%d' % (mode, c)] = np.array(bool(w.exploit))
            out['%s/n_ep%d' % (mode, c)] = np.array(n_ep)
        out['%s/success_rate' % mode] = np.array(w.current_success_rate())
        if eval_:
            out['%s/mean_Q' % mode] = np.array(w.current_mean_Q())
        out['%s/n_cycles' % mode] = np.array(n_cycles)
    save('rollout', **out)


def gen_adam():
    from baselines.common.mpi_adam import MpiAdam
    rng = np.random.RandomState(3)
    P = 1000
    theta0 = rng.randn(P).astype(np.float32)
    ad = object.__new__(MpiAdam)
    ad.beta1, ad.beta2, ad.epsilon, ad.scale_grad_by_procs = 0.9, 0.999, 1e-08, False
    ad.m = np.zeros(P, 'float32')
    ad.v = np.zeros(P, 'float32')
    ad.t = 0
    store = {'theta': theta0.copy()}
    ad.getflat = lambda: store['theta']
    ad.setfromflat = lambda x: store.__setitem__('theta', np.asarray(x).astype(np.float32))   # TF var is float32
    from mpi4py import MPI
    ad.comm = MPI.COMM_WORLD
    grads, thetas, ms, vs = [], [], [], []
    for k in range(12):
        g = (rng.randn(P) * (10.0 ** rng.randint(-6, 2))).astype(np.float32)
        ad.update(g, 1e-3)
        grads.append(g)
        thetas.append(store['theta'].copy())
        ms.append(ad.m.copy())
        vs.append(ad.v.copy())
    save('adam', theta0=theta0, grads=np.array(grads), thetas=np.array(thetas), ms=np.array(ms), vs=np.array(vs),
         numpy_version=np.array(np.__version__))


def gen_normalizer():
    from baselines.her.normalizer import Normalizer
    rng = np.random.RandomState(8)
    size = 12
    nz = object.__new__(Normalizer)
    nz.size = size
    nz.local_sum = np.zeros(size, np.float32)
    nz.local_sumsq = np.zeros(size, np.float32)
    nz.local_count = np.zeros(1, np.float32)
    import threading
    nz.lock = threading.Lock()
    vs, sums, sumsqs, counts = [], [], [], []
    for k in range(5):
        v = (rng.randn(37, size) * 3).astype(np.float32).astype(np.float64)
        nz.update(v)
        vs.append(v)
        s, ss, c = nz.synchronize(nz.local_sum.copy(), nz.local_sumsq.copy(), nz.local_count.copy())
        sums.append(s)
        sumsqs.append(ss)
        counts.append(c)
    save('normalizer', vs=np.array(vs), sums=np.array(sums), sumsqs=np.array(sumsqs), counts=np.array(counts))


def gen_mpi_moments():
    from baselines.common.mpi_moments import mpi_moments
    rng = np.random.RandomState(2)
    xs, means, stds = [], [], []
    for k in range(4):
        x = rng.randn(7)
        mean, std, count = mpi_moments(x)
        xs.append(x)
        means.append(mean)
        stds.append(std)
    save('mpi_moments', xs=np.array(xs), means=np.array(means), stds=np.array(stds))


def gen_sagg_riac():
    """baselines/her/active_goal_sampling.py driven for 60 rounds on a 3-D goal space whose competence improves over
    time in the half-space x > 0.1 only (so that regions split): inputs, the split decisions, the region tree and the
    goals sampled in between, all on one seeded NumPy stream."""
    import contextlib
    import io
    from baselines.her.active_goal_sampling import SAGG_RIAC
    np.random.seed(77)
    lo, hi = -0.5 * np.ones(3, np.float32), 0.5 * np.ones(3, np.float32)
    sel = SAGG_RIAC(lo, hi)
    data_rng = np.random.RandomState(5)
    rounds_goals, rounds_comp, splits, orders, samples, nregs = [], [], [], [], [], []
    for rnd in range(60):
        n = int(data_rng.randint(0, 40))
        goals = [data_rng.uniform(lo, hi).astype(np.float32) for _ in range(n)]
        p_succ = min(1.0, rnd / 25.0)
        comp = [float(g[0] > 0.1 and data_rng.rand() < p_succ) for g in goals]
        with contextlib.redirect_stdout(io.StringIO()):          # the reference prints while it searches
            new_split, order = sel.update(goals, comp)
        rounds_goals.append(np.array(goals, np.float32).reshape(n, 3))
        rounds_comp.append(np.array(comp, np.float64))
        splits.append(bool(new_split))
        orders.append(np.array(order if order is not None else [], np.int64))
        nregs.append(sel.nb_regions)
        samples.append(np.array([sel.sample_goal() for _ in range(3)], np.float32))
    assert sel.nb_regions >= 3, sel.nb_regions
    arrs = dict(lo=lo, hi=hi, n_rounds=np.array(60), splits=np.array(splits), nregs=np.array(nregs),
                final_low=np.array([b.low for b in sel.region_bounds]),
                final_high=np.array([b.high for b in sel.region_bounds]),
                final_probas=np.array(sel.probas), final_interest=np.array(sel.interest, np.float64),
                final_sizes=np.array([len(r[0]) for r in sel.regions]), max_difference=np.array(sel.max_difference),
                end_draw=np.array(np.random.uniform()))
    for i in range(60):
        arrs['goals_%d' % i] = rounds_goals[i]
        arrs['comp_%d' % i] = rounds_comp[i]
        arrs['order_%d' % i] = orders[i]
        arrs['samples_%d' % i] = samples[i]
    save('sagg_riac', **arrs)


def gen_ddpg_host():
    """The NumPy halves of the reference's DDPG -- store_episode (ddpg.py:163-223: task-activity test, routing into the
    per-task buffers, the HER-sampled batch fed to the normalisers) and sample_batch (ddpg.py:251-360: buffer
    proportions, per-buffer sampling, concat + shuffle, clip) -- executed from the reference's own class.  The module
    imports against the permissive tensorflow stub; an instance is made with object.__new__ and only the attributes
    these two methods read are set (no graph, no session).  The normalisers are recorders."""
    from collections import OrderedDict
    if not hasattr(np, 'int'):
        np.int = int                                    # the reference uses the alias NumPy removed in 1.24
    from baselines.her import her as ref_her
    from baselines.her.ddpg import DDPG
    from baselines.her.replay_buffer import ReplayBuffer
    from oracle.reward import make_reward_fun

    class Rec:                                          # stands in for Normalizer: keeps what update() was given
        def __init__(self): self.seen = []
        def update(self, v): self.seen.append(np.array(v, dtype=np.float64, copy=True))
        def recompute_stats(self): pass

    def build(nb, dimo, T, structure, task_replay, t_id=None, batch_size=64, cap_eps=40):
        ag_ids, g_ids = tables(nb)
        G = 3 * nb
        fn = ref_her.make_sample_multi_task_her_transitions('her', 4, task_replay, make_reward_fun(ag_ids, g_ids),
                                                            tasks_ag_id=ag_ids, tasks_g_id=g_ids)
        shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), task_descr=(T, nb), change=(T, G),
                      info_is_success=(T, 1))
        d = object.__new__(DDPG)
        d.structure, d.task_replay, d.nb_tasks, d.t_id = structure, task_replay, nb, t_id
        d.tasks_ag_id, d.tasks_g_id, d.T, d.batch_size, d.eps_task = ag_ids, g_ids, T, batch_size, 0.4
        d.relative_goals, d.clip_obs, d.dimg, d.dimag = False, 200., G, G
        d.buffer = [ReplayBuffer(shapes, T * cap_eps, T, fn) for _ in range(nb + 1)]
        if len(d.buffer) > 5:                           # ddpg.py:106-110
            for i in range(6, len(d.buffer)):
                d.buffer[i] = d.buffer[5]
        d.sample_transitions = fn
        d.o_stats, d.g_stats = Rec(), Rec()
        d.stage_shapes = OrderedDict((k, None) for k in ['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r'])
        return d

    out = {}
    cases = [('arm4', 4, 40, 'curious', 'replay_task_cp_buffer', None),
             ('arm8', 8, 52, 'curious', 'replay_task_cp_buffer', None),
             ('arm4rand', 4, 40, 'curious', 'replay_task_random_buffer', None),
             ('expert2', 4, 40, 'task_experts', 'replay_current_task_buffer', 2)]
    T = 10
    for ci, (name, nb, dimo, structure, tr, t_id) in enumerate(cases):
        d = build(nb, dimo, T, structure, tr, t_id)
        rng = np.random.RandomState(900 + ci)
        cps = [np.zeros(nb), np.linspace(0.5, 0.0, nb), np.array([0.3] + [0.0] * (nb - 1))]
        for rnd in range(2):                            # two stores: the second one meets non-empty buffers
            ep = synth_episodes(rng, 12, T, dimo, nb)
            out.update(flat_dict('%s/store%d/in/' % (name, rnd), ep))
            np.random.seed(100 * ci + rnd)
            d.store_episode({k: v.astype(np.float64) for k, v in ep.items()}, cps[rnd], 12 * (rnd + 1))
            out['%s/store%d/sizes' % (name, rnd)] = np.array([b.current_size for b in d.buffer])
            out['%s/store%d/stats_o' % (name, rnd)] = d.o_stats.seen[-1]
            out['%s/store%d/stats_g' % (name, rnd)] = d.g_stats.seen[-1]
        for i in range(min(nb + 1, 6)):
            out['%s/buffer%d/o' % (name, i)] = d.buffer[i].buffers['o'][:d.buffer[i].current_size].copy()
            out['%s/buffer%d/g' % (name, i)] = d.buffer[i].buffers['g'][:d.buffer[i].current_size].copy()
        for k, cp in enumerate(cps):
            d.cp = cp
            np.random.seed(7000 + 10 * ci + k)
            batch = d.sample_batch()
            out['%s/sample%d/cp' % (name, k)] = cp
            out['%s/sample%d/proportions' % (name, k)] = np.asarray(d.proportions)
            for key, arr in zip(d.stage_shapes.keys(), batch):
                out['%s/sample%d/%s' % (name, k, key)] = np.asarray(arr)
    out['cases'] = np.array([c[0] for c in cases])
    out['cfg'] = np.array([T, 64, 40])
    save('ddpg_host', **out)


def gen_get_actions():
    """The post-processing half of the reference's DDPG.get_actions (ddpg.py:147-160: in-place float32 noise add, clip,
    eps-greedy replacement, 1-D result for a single row) executed from the reference's own method; the TensorFlow half
    (sess.run of the policy) is replaced by a session object that returns given float32 policy outputs."""
    from baselines.her.ddpg import DDPG

    class Sess:
        def __init__(self): self.out = None
        def run(self, vals, feed_dict=None): return [a.copy() for a in self.out[:len(vals)]]

    class Net:
        pi_tf = 'pi'; Q_pi_tf = 'Qpi'; o_tf = 'o'; g_tf = 'g'; u_tf = 'u'; td_tf = 'td'

    d = object.__new__(DDPG)
    d.structure, d.relative_goals, d.clip_obs, d.max_u = 'curious', False, 200., 1.
    d.dimo, d.dimg, d.dimag, d.dimu, d.dimtd = 40, 12, 12, 4, 4
    d.main = d.target = Net()
    d.sess = Sess()
    out = {}
    rng = np.random.RandomState(31)
    ns = [1, 2, 17, 256]
    for n in ns:
        pi = rng.uniform(-1.2, 1.2, [n, 4]).astype(np.float32)
        Q = rng.randn(n, 1).astype(np.float32)
        o = rng.randn(n, 40).astype(np.float32); g = rng.randn(n, 12).astype(np.float32)
        td = np.eye(4, dtype=np.float32)[rng.randint(4, size=n)]
        for tag, ne, re in (('noisy', 0.2, 0.3), ('greedy', 0.0, 0.0)):
            d.sess.out = [pi, Q]
            np.random.seed(1000 + n)
            u, q = d.get_actions(o, o[:, :12], g, task_descr=td, noise_eps=ne, random_eps=re, compute_Q=True)
            out['n%d/%s/u' % (n, tag)] = np.asarray(u)
            out['n%d/%s/Q' % (n, tag)] = np.asarray(q)
            out['n%d/%s/next_uniform' % (n, tag)] = np.array(np.random.uniform())
        out['n%d/pi' % n] = pi
        out['n%d/Qin' % n] = Q
    out['ns'] = np.array(ns)
    save('get_actions', **out)


if __name__ == '__main__':
    install_stubs()
    gen_sagg_riac()
    gen_her()
    gen_replay_buffer()
    gen_queues()
    gen_rollout()
    gen_adam()
    gen_normalizer()
    gen_mpi_moments()
    gen_ddpg_host()
    gen_get_actions()
