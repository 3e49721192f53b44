"""The oracle against golden vectors captured from the imported reference (tools/gen_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, sub
from oracle import her as oher
from oracle.reward import make_reward_fun
from oracle.replay_buffer import ReplayBuffer
from oracle.queues import CompetenceQueue, task_probabilities
from oracle.optim import adam_update
from oracle.normalizer import Normalizer


def tables(nb):
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb)]
    return ids, [list(x) for x in ids]


HER = load_golden('her')


@pytest.mark.parametrize('name', [str(n) for n in HER['names']])
def test_her_sampler_matches_reference(name):
    nb, dimo, E, T, B, seed, ttr, flat = [int(x) for x in HER[name + '/cfg']]
    ttr = None if ttr < 0 else ttr
    task_replay = str(HER[name + '/task_replay'])
    goal_replay = str(HER[name + '/goal_replay'])
    ep = {k: v.astype(np.float64) for k, v in sub(HER, name + '/in/').items()}
    ep['o_2'] = ep['o'][:, 1:, :]
    ep['ag_2'] = ep['ag'][:, 1:, :]
    ag_ids, g_ids = tables(nb)
    reward = make_reward_fun(ag_ids, g_ids)
    rng = np.random.RandomState(seed)
    cp = HER[name + '/cp_proba'] if (name + '/cp_proba') in HER.files else None
    if flat:
        fn = oher.make_sample_her_transitions(goal_replay, 4, reward, tasks_ag_id=ag_ids, tasks_g_id=g_ids, rng=rng)
    else:
        fn = oher.make_sample_multi_task_her_transitions(goal_replay, 4, task_replay, reward, tasks_ag_id=ag_ids,
                                                         tasks_g_id=g_ids, rng=rng)
    tr = fn(ep, B, task_to_replay=ttr, cp_proba=cp)
    want = sub(HER, name + '/out/')
    assert set(tr.keys()) == set(want.keys())
    for k in want:
        assert tr[k].shape == want[k].shape, k
        np.testing.assert_array_equal(np.asarray(tr[k], dtype=np.float64), want[k].astype(np.float64), err_msg=k)
    # the stored raw draws are the same legacy stream
    d = oher.draw_her(np.random.RandomState(seed), E, T, B)
    for a, key in zip(d, ['ep', 't', 'u_her', 'u_off']):
        np.testing.assert_array_equal(a, HER[name + '/draw/' + key])
    # both reward values occur somewhere in the fixture set (checked globally below)


def test_her_fixtures_cover_both_rewards_and_relabels():
    seen = set()
    for name in [str(n) for n in HER['names']]:
        seen |= set(np.unique(HER[name + '/out/r']).tolist())
    assert seen == {0.0, -1.0}


def test_replay_buffer_matches_reference():
    G = load_golden('replay_buffer')
    nb, dimo, T, cap, seed = [int(x) for x in G['cfg']]
    ag_ids, g_ids = tables(nb)
    rng = np.random.RandomState(seed)
    fn = oher.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', make_reward_fun(ag_ids, g_ids),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids, rng=rng)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, 12), ag=(T + 1, 12), task_descr=(T, nb), change=(T, 12),
                  info_is_success=(T, 1))
    rb = ReplayBuffer(shapes, T * cap, T, fn, rng=rng)
    for step, inc in enumerate(G['incs']):
        ep = {k: v.astype(np.float64) for k, v in sub(G, 'step%d/in/' % step).items()}
        rb.store_episode(ep)
        assert rb.get_current_episode_size() == int(G['step%d/current_size' % step])
        assert rb.get_transitions_stored() == int(G['step%d/n_stored' % step])
        np.testing.assert_array_equal(rb.buffers['o'][:rb.current_size, 0, :3], G['step%d/o_rows' % step])
    assert rb.full
    tr = rb.sample(64, task_to_replay=1)
    want = sub(G, 'sample/')
    for k in want:
        np.testing.assert_array_equal(np.asarray(tr[k], dtype=np.float64), want[k].astype(np.float64), err_msg=k)
    for k in rb.buffers:
        np.testing.assert_array_equal(rb.buffers[k][:rb.current_size], G['final/' + k])


def test_competence_queue_matches_reference():
    G = load_golden('queues')
    q = CompetenceQueue(window=int(G['window']))
    for i, row in enumerate(G['chunks']):
        q.update([x for x in row if x >= 0])
        assert q.size == int(G['size'][i])
        assert float(q.C) == float(G['C'][i])
        assert float(q.CP) == float(G['CP'][i])


def test_task_probabilities_match_reference_rollout():
    G = load_golden('rollout')
    nb = int(G['cfg'][0])
    n = int(G['train/n_cycles'])
    checked = 0
    for c in range(n):
        if bool(G['train/exploit%d' % c]):
            # exploit rollouts reset p to uniform *before* the rollout and recompute it afterwards
            pass
        p = task_probabilities(G['train/CP%d' % c], nb, 0.4)
        np.testing.assert_array_equal(p, G['train/p%d' % c])
        checked += 1
    assert checked == n
    # CP became non-zero at some point, so the non-uniform branch is exercised
    assert any(G['train/CP%d' % c].sum() > 0 for c in range(n))


def test_adam_matches_reference():
    G = load_golden('adam')
    theta = G['theta0'].copy()
    P = theta.shape[0]
    m = np.zeros(P, np.float32)
    v = np.zeros(P, np.float32)
    t = 0
    th32, m32, v32, t32 = theta.copy(), m.copy(), v.copy(), 0
    worst = 0.0
    for k in range(G['grads'].shape[0]):
        g = G['grads'][k]
        # NumPy>=2 promotion of the imported reference, bit for bit
        theta, m, v, t = adam_update(theta, m, v, t, g, 1e-3, nep50=True)
        np.testing.assert_array_equal(m, G['ms'][k])
        np.testing.assert_array_equal(v, G['vs'][k])
        np.testing.assert_array_equal(theta, G['thetas'][k])
        # historical float32 arithmetic (what the HIP kernel implements): within 2 float32 ulp
        th32, m32, v32, t32 = adam_update(th32, m32, v32, t32, g, 1e-3, nep50=False)
        np.testing.assert_array_equal(m32, G['ms'][k])
        np.testing.assert_array_equal(v32, G['vs'][k])
        ulp = np.spacing(np.abs(G['thetas'][k]).astype(np.float32))
        worst = max(worst, float(np.max(np.abs(th32 - G['thetas'][k]) / ulp)))
    assert worst <= 2.0 * (G['grads'].shape[0]), worst   # <= 2 ulp drift per step, accumulated


def test_normalizer_accumulation_matches_reference():
    G = load_golden('normalizer')
    nz = Normalizer(G['vs'].shape[2])
    for k in range(G['vs'].shape[0]):
        nz.update(G['vs'][k])
        np.testing.assert_array_equal(nz._mpi_average(nz.local_sum.copy()), G['sums'][k])
        np.testing.assert_array_equal(nz._mpi_average(nz.local_sumsq.copy()), G['sumsqs'][k])
        np.testing.assert_array_equal(nz._mpi_average(nz.local_count.copy()), G['counts'][k])


def test_mpi_moments_single_rank():
    G = load_golden('mpi_moments')
    for x, mean, std in zip(G['xs'], G['means'], G['stds']):
        np.testing.assert_allclose(np.mean(x), mean, rtol=1e-14)
        np.testing.assert_allclose(np.std(x), std, rtol=1e-12)


def test_sagg_riac_matches_reference():
    """curious_amd.active_goal_sampling.SAGG_RIAC replays the reference's run (tools/gen_golden.py::gen_sagg_riac):
    same split decisions, region tree, sampling probabilities and sampled goals on the same seeded NumPy stream."""
    from curious_amd.active_goal_sampling import SAGG_RIAC
    d = load_golden('sagg_riac')
    np.random.seed(77)
    sel = SAGG_RIAC(d['lo'], d['hi'])
    for i in range(int(d['n_rounds'])):
        goals = [g for g in d['goals_%d' % i]]
        new_split, order = sel.update(goals, d['comp_%d' % i].tolist())
        assert bool(new_split) == bool(d['splits'][i]), i
        np.testing.assert_array_equal(np.array(order if order is not None else [], np.int64), d['order_%d' % i])
        assert sel.nb_regions == d['nregs'][i]
        got = np.array([sel.sample_goal() for _ in range(3)], np.float32)
        np.testing.assert_array_equal(got, d['samples_%d' % i])
    assert d['splits'].sum() >= 2 and sel.nb_regions >= 3          # the fixture does exercise splitting
    np.testing.assert_array_equal(np.array([b.low for b in sel.get_regions]), d['final_low'])
    np.testing.assert_array_equal(np.array([b.high for b in sel.get_regions]), d['final_high'])
    np.testing.assert_array_equal(np.array(sel.probas), d['final_probas'])
    np.testing.assert_array_equal(np.array(sel.interest, np.float64), d['final_interest'])
    np.testing.assert_array_equal(np.array([len(r[0]) for r in sel.regions]), d['final_sizes'])
    assert sel.max_difference == float(d['max_difference'])
    assert np.random.uniform() == float(d['end_draw'])             # same amount of the stream consumed


DDPG_HOST = load_golden('ddpg_host')
HOST_CASES = {'arm4': (4, 40, 'curious', 'replay_task_cp_buffer', None),
              'arm8': (8, 52, 'curious', 'replay_task_cp_buffer', None),
              'arm4rand': (4, 40, 'curious', 'replay_task_random_buffer', None),
              'expert2': (4, 40, 'task_experts', 'replay_current_task_buffer', 2)}


class _Rec:
    """Stands in for a normaliser: keeps what update() was given (the generator did the same to the reference)."""
    def __init__(self): self.seen = []
    def update(self, v): self.seen.append(np.array(v, dtype=np.float64, copy=True))
    def recompute_stats(self): pass


@pytest.mark.parametrize('name', [str(n) for n in DDPG_HOST['cases']])
def test_ddpg_store_episode_and_sample_batch_match_the_reference_class(name):
    """oracle.ddpg.OracleDDPG.store_episode / sample_batch against the outputs of the reference's OWN DDPG methods
    (ddpg.py:163-223 and :251-360, executed by tools/gen_golden.py from an instance without graph or session): task
    activity + routing incl. the j < 5 rule and the aliased distractor buffers, the batch fed to the normalisers, buffer
    proportions for three CP vectors, per-buffer sampling, concat + shuffle, clip -- on the reference's random stream."""
    from oracle.ddpg import OracleDDPG, STAGE_KEYS
    from oracle.replay_buffer import ReplayBuffer as OBuf
    G = DDPG_HOST
    T, B, cap = [int(x) for x in G['cfg']]
    nb, dimo, structure, tr, t_id = HOST_CASES[name]
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=3 * nb, ag=3 * nb, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, 3 * nb), ag=(T + 1, 3 * nb), task_descr=(T, nb),
                  change=(T, 3 * nb), info_is_success=(T, 1))
    sampler = oher.make_sample_multi_task_her_transitions('her', 4, tr, make_reward_fun(ag_ids, g_ids),
                                                          tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = [OBuf(shapes, T * cap, T, sampler) for _ in range(nb + 1)]
    agent = OracleDDPG(dims, T, bufs, sampler, ag_ids, g_ids, hidden=8, batch_size=B, task_replay=tr,
                       structure=structure, t_id=t_id, weight_rng=np.random.RandomState(0))
    agent.o_stats, agent.g_stats = _Rec(), _Rec()
    ci = list(HOST_CASES).index(name)
    cps = [G['%s/sample%d/cp' % (name, k)] for k in range(3)]
    for rnd in range(2):
        ep = {k: v.astype(np.float64) for k, v in sub(G, '%s/store%d/in/' % (name, rnd)).items()}
        np.random.seed(100 * ci + rnd)
        agent.store_episode(ep, cps[rnd], 12 * (rnd + 1))
        np.testing.assert_array_equal([b.current_size for b in agent.buffer], G['%s/store%d/sizes' % (name, rnd)])
        np.testing.assert_array_equal(agent.o_stats.seen[-1], G['%s/store%d/stats_o' % (name, rnd)])
        np.testing.assert_array_equal(agent.g_stats.seen[-1], G['%s/store%d/stats_g' % (name, rnd)])
    for i in range(min(nb + 1, 6)):
        n = agent.buffer[i].current_size
        np.testing.assert_array_equal(agent.buffer[i].buffers['o'][:n], G['%s/buffer%d/o' % (name, i)])
        np.testing.assert_array_equal(agent.buffer[i].buffers['g'][:n], G['%s/buffer%d/g' % (name, i)])
    for k, cp in enumerate(cps):
        agent.cp = cp
        np.random.seed(7000 + 10 * ci + k)
        batch = agent.sample_batch()
        np.testing.assert_array_equal(agent.proportions, G['%s/sample%d/proportions' % (name, k)])
        for key, arr in zip(STAGE_KEYS, batch):
            np.testing.assert_array_equal(np.asarray(arr, dtype=np.float64), G['%s/sample%d/%s' % (name, k, key)],
                                          err_msg='%s sample %d %s' % (name, k, key))


def test_action_postprocessing_matches_the_reference_method():
    """oracle.ddpg.action_postprocess against DDPG.get_actions of the reference (ddpg.py:147-160, run by
    tools/gen_golden.py with a session object that returns given float32 policy outputs): in-place float32 noise add,
    clip, eps-greedy replacement; same amount of the NumPy stream consumed, also when both eps are 0."""
    from oracle.ddpg import action_postprocess
    G = load_golden('get_actions')
    for n in [int(x) for x in G['ns']]:
        for tag, ne, re in (('noisy', 0.2, 0.3), ('greedy', 0.0, 0.0)):
            np.random.seed(1000 + n)
            u = action_postprocess(G['n%d/pi' % n].copy(), np.random, ne, re, 1.0).astype(np.float32)
            want = G['n%d/%s/u' % (n, tag)]
            assert want.shape == ((4,) if n == 1 else (n, 4))            # ddpg.py:153-154
            np.testing.assert_array_equal(u.reshape(want.shape), want)
            assert float(np.random.uniform()) == float(G['n%d/%s/next_uniform' % (n, tag)])
            np.testing.assert_array_equal(G['n%d/%s/Q' % (n, tag)], G['n%d/Qin' % n])
