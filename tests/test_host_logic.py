"""Host-side logic of curious_amd that needs no GPU: RolloutWorker (generic env path), competence queues, task
probabilities, buffer proportions, helpers -- against the reference's golden vectors and the oracle."""
import os
import types

import numpy as np
import pytest

from conftest import load_golden, sub


class FakePolicy:
    """The linear policy the golden rollout fixture was generated with (tools/gen_golden.py)."""

    def __init__(self, A, max_u=1.0):
        self.A, self.max_u = A, max_u

    def get_actions(self, o, ag, g, task_descr=None, noise_eps=0., random_eps=0., use_target_net=False,
                    compute_Q=False):
        x = np.concatenate([o.reshape(len(o), -1), g.reshape(len(g), -1), task_descr.reshape(len(o), -1)], axis=1)
        u = 0.1 * np.tanh(x.astype(np.float64) @ self.A)
        gg = g.reshape(len(g), -1)
        u[:, :3] += 4.0 * (gg[:, :3] - o.reshape(len(o), -1)[:, :3]) * (gg[:, :1] > 0)
        Q = u.sum(axis=1, keepdims=True)
        noise = noise_eps * self.max_u * np.random.randn(*u.shape)
        u = u + noise
        u = np.clip(u, -self.max_u, self.max_u)
        u += np.random.binomial(1, random_eps, u.shape[0]).reshape(-1, 1) * (
            np.random.uniform(low=-self.max_u, high=self.max_u, size=u.shape) - u)
        if u.shape[0] == 1:
            u = u[0]
        return [u.copy(), Q] if compute_Q else u.copy()


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_rollout_worker_generic_path_matches_reference(mode):
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    from oracle.env import SyntheticMultiTaskArm
    G = load_golden('rollout')
    nb, dimo, T, B = [int(x) for x in G['cfg']]
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    eval_ = mode == 'eval'
    counter = [0]

    def make_env():
        e = SyntheticMultiTaskArm(nb, dimo, T, seed=0, env_id=counter[0])
        counter[0] += 1
        return e
    np.random.seed(31337)
    w = RolloutWorker(make_env, FakePolicy(G['A']), dims, logger, T=T, rollout_batch_size=B, exploit=eval_,
                      use_target_net=False, compute_Q=eval_, noise_eps=0.2, random_eps=0.3, structure='curious',
                      task_selection='active_competence_progress', goal_selection='random', queue_length=4,
                      eval=eval_)
    w.seed(555)
    n_cycles = int(G['%s/n_cycles' % mode])
    for c in range(n_cycles):
        ep, CP, n_ep = w.generate_rollouts()
        if ('%s/cycle%d/o' % (mode, c)) in G.files:
            want = sub(G, '%s/cycle%d/' % (mode, c))
            assert set(ep.keys()) == set(want.keys())
            for k in want:
                assert ep[k].shape == want[k].shape and ep[k].dtype == want[k].dtype, k
                np.testing.assert_array_equal(ep[k], want[k], err_msg=k)
        np.testing.assert_array_equal(np.asarray(CP, dtype=np.float64), G['%s/CP%d' % (mode, c)])
        np.testing.assert_array_equal(np.asarray(w.p, dtype=np.float64), G['%s/p%d' % (mode, c)])
        np.testing.assert_array_equal(np.asarray(w.C, dtype=np.float64), G['%s/C%d' % (mode, c)])
        assert bool(w.exploit) == bool(G['%s/exploit%d' % (mode, c)])
        assert n_ep == int(G['%s/n_ep%d' % (mode, c)])
    assert w.current_success_rate() == float(G['%s/success_rate' % mode])
    if eval_:
        assert w.current_mean_Q() == float(G['%s/mean_Q' % mode])
    keys = [k for k, _ in w.logs('x')]
    assert keys[:2] == ['x/success_rate', 'x/avg_reward'] and keys[-1] == 'x/episode'


def test_competence_queue_and_probabilities_match_reference():
    from curious_amd.queues import CompetenceQueue, task_probabilities
    G = load_golden('queues')
    q = CompetenceQueue(window=int(G['window']))
    for i, row in enumerate(G['chunks']):
        q.update([x for x in row if x >= 0])
        assert (q.size, float(q.C), float(q.CP)) == (int(G['size'][i]), float(G['C'][i]), float(G['CP'][i]))
    R = load_golden('rollout')
    for c in range(int(R['train/n_cycles'])):
        np.testing.assert_array_equal(task_probabilities(R['train/CP%d' % c], 4, 0.4), R['train/p%d' % c])
    # fix-up branches (rollout.py:390-393)
    p = task_probabilities([0.3, 0.3, 0.1], 3, 0.4)
    assert abs(p.sum() - 1) < 1e-15


@pytest.mark.parametrize('case', range(6))
def test_buffer_proportions_match_oracle(case):
    from curious_amd.ddpg import DDPG
    from oracle.ddpg import buffer_proportions, expert_proportions
    rng = np.random.RandomState(case)
    nb = [4, 8, 4, 4, 8, 4][case]
    T, B = 50, 256
    sizes = rng.randint(0, 30, nb + 1)
    sizes[0] = 0
    if case == 2:
        sizes[1:] = 0
        sizes[2] = 3
    cp = rng.rand(nb) * (rng.rand(nb) > 0.4)
    if case == 3:
        cp[:] = 0
    fake = types.SimpleNamespace(nb_tasks=nb, T=T, batch_size=B, cp=cp, eps_task=0.4,
                                 buffer=[types.SimpleNamespace(current_size=int(s)) for s in sizes],
                                 structure='curious',
                                 task_replay='replay_task_cp_buffer' if case != 5 else 'replay_task_random_buffer',
                                 t_id=None)
    got = DDPG._proportions(fake)
    want = buffer_proportions(sizes * T, T, B, fake.task_replay, cp, 0.4)
    np.testing.assert_array_equal(got, want)
    assert got.sum() == B
    fake.structure, fake.t_id = 'task_experts', 1
    np.testing.assert_array_equal(DDPG._proportions(fake), expert_proportions(sizes * T, B, 1))


def test_store_args_and_plugin_strings():
    from curious_amd.util import store_args, import_function, convert_episode_to_batch_major

    class A:
        @store_args
        def __init__(self, a, b=2, *, c=3, **kw):
            pass
    x = A(1, c=5, z=9)
    assert (x.a, x.b, x.c, x.z) == (1, 2, 5, 9)
    assert import_function('baselines.her.actor_critic:MultiTaskActorCritic').modular is True
    assert import_function('curious_amd.actor_critic:ActorCritic').modular is False
    ep = convert_episode_to_batch_major(dict(o=[np.zeros((3, 2))] * 5))
    assert ep['o'].shape == (3, 5, 2)


def test_logger_csv_protocol(tmp_path):
    from curious_amd import logger
    logger.configure(dir=str(tmp_path))
    logger.record_tabular('epoch', 0)
    logger.record_tabular('test/success_rate', '0.5')
    logger.dump_tabular()
    logger.record_tabular('epoch', 1)
    logger.record_tabular('test/success_rate', '0.6')
    logger.record_tabular('train/episode', 10)
    logger.dump_tabular()
    import csv
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), 'progress.csv'))))
    assert [r['epoch'] for r in rows] == ['0', '1'] and rows[1]['train/episode'] == '10'


def test_record_layout_views_roundtrip():
    import torch
    from curious_amd.layout import RecordLayout, pack_episodes
    T = 5
    shapes = dict(o=(T + 1, 7), u=(T, 2), g=(T, 3), ag=(T + 1, 3), task_descr=(T, 2), change=(T, 3),
                  info_is_success=(T, 1))
    L = RecordLayout(shapes, T)
    assert L.off['o'] == 0 and L.off['ag'] == 7 and L.row_stride % 4 == 0
    rng = np.random.RandomState(0)
    ep = {k: rng.randn(4, s[0], s[1]).astype(np.float32) for k, s in shapes.items()}
    rec = torch.from_numpy(pack_episodes(L, ep))
    views = L.record_views(rec)
    for k in shapes:
        np.testing.assert_array_equal(views[k].numpy(), ep[k])
    assert set(L.batch_cols) == {'o', 'task_descr', 'u', 'g', 'o_2', 'g_2', 'r', 'ag', 'ag_2', 'change',
                                 'info_is_success'}


def test_perturbation_switch_sets_bias_on_the_first_two_envs_or_refuses():
    """train.py:142-146: at epoch 250 of a perturbation study env.unwrapped.bias = True on envs 0 and 1 of both workers."""
    from curious_amd.experiment.train import perturb_envs

    class Env:
        def __init__(self, with_bias=True):
            if with_bias:
                self.bias = False
        unwrapped = property(lambda self: self)

    class Worker:
        def __init__(self, envs):
            self.envs = envs

    a, b = Worker([Env() for _ in range(3)]), Worker([Env() for _ in range(2)])
    perturb_envs(a, b)
    assert [e.bias for e in a.envs] == [True, True, False] and [e.bias for e in b.envs] == [True, True]
    with pytest.raises(NotImplementedError):                         # the GPU-resident batch is ONE env object
        perturb_envs(Worker([Env()]), b)
    with pytest.raises(NotImplementedError):                         # an env without the switch: refused, not ignored
        perturb_envs(Worker([Env(False), Env(False)]), b)


def test_adam_step_size_table_equals_the_scalar_formula():
    """The ring of Adam step sizes the device reads (DDPG._fill_alpha_table) is filled by a vectorised form of
    mpi_adam.py:30; it has to give the float32 values of the scalar form element for element, for any start step."""
    from curious_amd import ops
    for lr, b1, b2 in ((1e-3, 0.9, 0.999), (5e-4, 0.8, 0.99)):
        for t0 in (1, 4090, 123456, 10 ** 7):
            ts = np.arange(t0, t0 + 600)
            want = np.array([ops.adam_alpha(lr, int(t), b1, b2) for t in ts])
            got = ops.adam_alpha_table(lr, ts, b1, b2)
            assert got.dtype == np.float32 and np.array_equal(got, want)


def test_checkpointed_worker_histories_are_length_and_tail():
    """curious_amd.checkpoint.worker_state keeps, of the ever-growing task / goal histories (rollout.py:392-393), their
    length and their last HISTORY_TAIL entries -- all the '%_task' columns ever read (rollout.py:475); load_worker_state
    restores a history of that length with that tail, and a second checkpoint of the restored worker equals the first."""
    from collections import deque
    from curious_amd import checkpoint as ck
    from curious_amd.queues import CompetenceQueue

    def worker(n):
        w = types.SimpleNamespace(n_episodes=7, C=np.zeros(2), CP=np.ones(2), success_history=deque([1.0, 0.0]),
                                  reward_history=deque([-1.0]), Q_history=deque(), count=3, exploit=False, envs=[],
                                  _vrng=[np.random.RandomState(5)], p=np.array([0.5, 0.5]),
                                  competence_computers=[CompetenceQueue(window=4), CompetenceQueue(window=4)],
                                  task_history=deque(int(i % 2) for i in range(n)),
                                  goal_history=deque([float(i), 0.0, 1.0] if i % 3 else [] for i in range(n)))
        w.settle = lambda: None
        return w
    src = worker(250)
    st = ck.worker_state(src)
    assert st['task_history']['n'] == 250 and st['task_history']['tail'] == [int(i % 2) for i in range(150, 250)]
    assert st['goal_history']['n'] == 250 and len(st['goal_history']['tail']) == ck.HISTORY_TAIL
    dst = worker(3)
    ck.load_worker_state(dst, st)
    assert len(dst.task_history) == 250 and list(dst.task_history)[-100:] == list(src.task_history)[-100:]
    assert len(dst.goal_history) == 250 and list(dst.goal_history)[-1] == list(src.goal_history)[-1]
    again = ck.worker_state(dst)
    assert again['task_history'] == st['task_history'] and again['goal_history'] == st['goal_history']
    short = ck.worker_state(worker(30))                                # shorter than the tail: everything is kept
    assert short['task_history']['n'] == 30 and len(short['task_history']['tail']) == 30
    old = dict(st, task_history=[0, 1, 1], goal_history=[[1.0], [2.0]])   # a file written before the histories were cut
    ck.load_worker_state(dst, old)
    assert list(dst.task_history) == [0, 1, 1] and len(dst.goal_history) == 2
