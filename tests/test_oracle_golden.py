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
