"""Round-2 parity cases on a real MI355X: batched task experts (BASELINE configs[4]), the configs[0] plumbing run
through launch(), the Arm8 / 1 024-env acting path (configs[2]), reference-format persistence, the host-evaluated reward."""
import csv
import os
import pickle

import numpy as np
import pytest
import torch

from test_gpu_agent import T, build_pair, synth_episodes, tables

pytestmark = pytest.mark.gpu


def _expert_kit(nb=4, dimo=40, batch_size=256, hidden=256, cap_eps=64, normalize_obs=False):
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    G = 3 * nb
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    tr = 'replay_current_task_buffer'
    sampler = make_sample_multi_task_her_transitions('her', 4, tr, sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = make_pooled_buffers(shapes, T * cap_eps, T, sampler, nb + 1, alias_from=5)
    gamma = 1. - 1. / T

    def make(t_id, use_graph=False, **hooks):
        from curious_amd.ddpg import DDPG
        return DDPG(input_dims=dims, hidden=hidden, layers=3,
                    network_class='curious_amd.actor_critic:MultiTaskActorCritic', polyak=0.95, batch_size=batch_size,
                    Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1., clip_obs=200.,
                    scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None, relative_goals=False,
                    clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=normalize_obs,
                    sample_transitions=sampler, gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids, tasks_g_id=g_ids,
                    task_replay=tr, eps_task=0.4, structure='task_experts', t_id=t_id, seed=10 + t_id,
                    rng_mode='device', use_graph=use_graph, **hooks)
    return make, bufs, dims, shapes, (ag_ids, g_ids), tr


@pytest.mark.parametrize('use_graph', [False, True])
def test_batched_experts_equal_sequential_experts_and_oracle(use_graph, route):
    """curious_ddpg_update_experts: N = 4 experts in one launch sequence per update, bit-identical to 4 DDPG(t_id=i)
    objects updated one after the other (train.py:65-121, ddpg.py:302-318,335); every expert's loss within 1e-5 of the
    float64 oracle on the batch it drew."""
    from curious_amd.experts import ExpertBank
    from oracle.ddpg import OracleDDPG
    from curious_amd import ops
    from oracle.ddpg import STAGE_KEYS
    nb = 4
    rng = np.random.RandomState(2)
    ep = synth_episodes(rng, 40, nb, 40)
    groups = []
    for batched in (True, False):
        make, bufs, dims, shapes, ids, tr = _expert_kit()
        if batched:
            bank = ExpertBank(lambda t, **h: make(t, use_graph=use_graph, **h), nb)
            xs = list(bank)
        else:
            bank = None
            xs = [make(t, use_graph=use_graph) for t in range(nb)]
        np.random.seed(1)
        xs[0].store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(nb), 40)
        groups.append((bank, xs))
    (bank, bx), (_, sx) = groups
    assert all(b.current_size > 0 for b in bx[0].buffer[1:nb + 1])
    # identical starting points
    for a, b in zip(bx, sx):
        assert torch.equal(a.theta, b.theta) and a.scope == b.scope
    assert not torch.equal(bx[0].theta, bx[1].theta)                   # experts differ (seed + t_id)
    n_up = 23                                                          # singles to even parity + 2 chains + singles
    bank.train_batches(1)
    for x in sx:
        x.train_batches(1)
    # the oracle sees the batch every expert will consume in update 2 (it sits staged now) with the current parameters
    want = []
    for e, x in enumerate(bx):
        batch = [v.cpu().numpy().astype(np.float64) for v in x._layout.batch_views(x._staged, STAGE_KEYS).values()]
        orc = OracleDDPG(dims, T, [None] * (nb + 1), None, ids[0], ids[1], hidden=256, batch_size=256,
                         task_replay=tr, structure='task_experts', t_id=e, weight_rng=np.random.RandomState(0),
                         dtype=np.float64)
        orc.theta = ops.unpad_params(x.net_cfg, x.theta.cpu().numpy()).astype(np.float64)
        orc.theta_target = ops.unpad_params(x.net_cfg, x.theta_target.cpu().numpy()).astype(np.float64)
        want.append(float(orc.grads(batch)['Q_loss']))
    bank.train_batches(1)
    for x in sx:
        x.train_batches(1)
    for e, x in enumerate(bx):
        got = float(x._losses[0])
        assert abs(got - want[e]) <= 1e-5 * abs(want[e]), (e, got, want[e])
    bank.train_batches(n_up - 2)
    for x in sx:
        x.train_batches(n_up - 2)
    bank.update_target_net()
    for x in sx:
        x.update_target_net()
    torch.cuda.synchronize()
    assert bank.batched
    for a, b in zip(bx, sx):
        assert a.Q_adam.t == b.Q_adam.t == n_up and int(a._step_ctr) == n_up
        assert torch.equal(a.theta, b.theta) and torch.equal(a._m, b._m) and torch.equal(a._v, b._v)
        assert torch.equal(a.theta_target, b.theta_target)
        assert torch.equal(a._staged, b._staged)                       # next batch drawn with the same key
        assert float(a._losses[0]) == float(b._losses[0]) and torch.equal(a._Q_pi, b._Q_pi)
    # every expert samples its own buffer and relabels to its own task (ddpg.py:335)
    for e, x in enumerate(bx):
        td = x._layout.batch_views(x._staged, ['task_descr'])['task_descr'].cpu().numpy()
        assert x.proportions[e + 1] == 256
        assert np.all(td.sum(axis=1) == 1)


@pytest.mark.parametrize('case', range(int(os.environ.get('CURIOUS_FUZZ_EXPERTS', 6))))
def test_batched_experts_random_banks(case):
    """Batched == sequential over seeded banks of 2-8 experts (= tasks; with more than 5 the distractor buffers alias),
    observations of various widths, batches of 256 / 512, graph replay or eager, an odd number of updates: parameters,
    moments, target networks and the next staged batch of every expert, bit for bit."""
    from curious_amd.experts import ExpertBank
    rs = np.random.RandomState(5100 + case)
    nb = int(rs.choice([2, 3, 4, 4, 5, 6, 7, 8]))
    dimo = int(rs.randint(3 * nb + 4, min(3 * nb + 60, 124 - 4 * nb)))  # [o | td | u | g] has to fit the 128-float input row
    batch = int(rs.choice([256, 512]))
    use_graph = bool(rs.randint(0, 2))
    n_up = int(rs.choice([3, 8, 13]))
    ep = synth_episodes(np.random.RandomState(2 + case), 40, nb, dimo)
    groups = []
    for batched in (True, False):
        make, bufs, dims, shapes, ids, tr = _expert_kit(nb=nb, dimo=dimo, batch_size=batch)
        if batched:
            bank = ExpertBank(lambda t, **h: make(t, use_graph=use_graph, **h), nb)
            xs = list(bank)
        else:
            bank = None
            xs = [make(t, use_graph=use_graph) for t in range(nb)]
        np.random.seed(1)
        xs[0].store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(nb), 40)
        groups.append((bank, xs))
    (bank, bx), (_, sx) = groups
    routed = [i for i in range(1, min(nb, 5) + 1)]
    if not all(bx[0].buffer[i].current_size > 0 for i in routed):
        pytest.skip('a routed buffer stayed empty for this draw')     # (an expert without data cannot train: ddpg.py:302-318)
    trainable = [t for t in range(nb) if bx[0].buffer[t + 1].current_size > 0]
    if len(trainable) < nb:
        pytest.skip('tasks >= 5 never receive episodes (ddpg.py:183): their experts have nothing to sample')
    bank.train_batches(n_up)
    for x in sx:
        x.train_batches(n_up)
    bank.update_target_net()
    for x in sx:
        x.update_target_net()
    torch.cuda.synchronize()
    tag = 'case %d: nb %d dimo %d batch %d graph %s updates %d' % (case, nb, dimo, batch, use_graph, n_up)
    assert bank.batched, tag
    for a, b in zip(bx, sx):
        assert a.Q_adam.t == b.Q_adam.t == n_up, tag
        assert torch.equal(a.theta, b.theta) and torch.equal(a._m, b._m) and torch.equal(a._v, b._v), tag
        assert torch.equal(a.theta_target, b.theta_target) and torch.equal(a._staged, b._staged), tag
    bank.check_faults()


def test_batched_experts_fall_back_outside_the_lean_route():
    """Shapes the batched launch does not cover (hidden 64): the bank reports it and updates the experts one by one."""
    from curious_amd.experts import ExpertBank
    make, bufs, dims, shapes, ids, tr = _expert_kit(hidden=64, batch_size=64)
    bank = ExpertBank(lambda t, **h: make(t, **h), 2)
    ref = [make(t) for t in range(2)]
    # (the reference experts above share `bufs` with the bank: one store feeds both groups)
    ep = synth_episodes(np.random.RandomState(3), 20, 4, 40)
    np.random.seed(1)
    bank[0].store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(4), 20)
    bank.train_batches(3)
    for x in ref:
        x.train_batches(3)
    assert not bank.batched
    for a, b in zip(bank, ref):
        assert torch.equal(a.theta, b.theta)


# progress.csv columns of the reference: train.py:170-193 + ddpg.py:469-479 + rollout.py:451-483
def _expected_keys(nb, structure):
    keys = ['epoch', 'test/success_rate', 'test/avg_reward', 'test/mean_Q', 'test/episode', 'train/success_rate',
            'train/avg_reward', 'train/episode', 'stats_o/mean', 'stats_o/std', 'stats_g/mean', 'stats_g/std', 'Time']
    if structure in ('curious', 'task_experts'):
        for i in range(nb):
            keys += ['train/C_task%d' % i, 'train/CP_task%d' % i, 'train/%%_task%d' % i, 'train/p_task%d' % i,
                     'test/C_task%d' % i]
    if structure == 'task_experts':
        keys.append('IND_TASK_rollout')
    return set(keys)


@pytest.mark.parametrize('structure,extra', [
    ('curious', dict(task_selection='random')),                       # BASELINE configs[0]
    ('task_experts', dict(task_selection='random')),
    ('task_experts', dict(task_selection='random', experts_update='batched')),
    ('flat', dict(task_selection='random')),
    ('curious', dict(task_selection='active_competence_progress', normalize_obs=True)),   # --normalize_obs True
])
def test_launch_runs_every_structure_and_logs_the_reference_columns(tmp_path, structure, extra):
    """experiment.train.launch() end to end (train.py:217-339 -> train() :49-166): configs[0] = num_cpu 1, structure
    curious, task_selection random, rollout_batch_size 2; plus the other structures the CLI offers."""
    from curious_amd.experiment import config, train
    config.CACHED_ENVS.clear()
    over = dict(rollout_batch_size=2, n_cycles=2, n_batches=5, n_test_rollouts=2, rng_mode='device', use_graph=True,
                batch_size=256)
    extra = dict(extra)
    task_selection = extra.pop('task_selection')
    normalize_obs = extra.pop('normalize_obs', False)
    over.update(extra)
    task_replay = 'replay_task_cp_buffer' if structure == 'curious' else \
        'replay_current_task_buffer' if structure == 'task_experts' else ''
    if structure == 'flat':
        over.update(rng_mode='numpy', use_graph=False)
    best = train.launch(env='MultiTaskFetchArm4-v5', trial_id=0, n_epochs=2, num_cpu=1, seed=5, policy_save_interval=1,
                        clip_return=1, normalize_obs=normalize_obs, structure=structure, task_selection=task_selection,
                        goal_selection='random', goal_replay='her', task_replay=task_replay, save_policies=True,
                        override_params=over, save_root=str(tmp_path) + '/')
    assert 0.0 <= best <= 1.0
    run_dir = os.path.join(str(tmp_path), 'MultiTaskFetchArm4-v5', '0')
    rows = list(csv.DictReader(open(os.path.join(run_dir, 'progress.csv'))))
    assert len(rows) == 3                                              # epoch -1 (before training), 0, 1
    assert set(rows[-1].keys()) == _expected_keys(4, structure)
    assert [int(float(r['epoch'])) for r in rows] == [-1, 0, 1]
    assert all(np.isfinite(float(r['stats_o/std'])) for r in rows)
    # policy_{best,latest,N}.pkl cadence (train.py:195-205) + the weights files next to them
    for name in ('policy_best.pkl', 'policy_latest.pkl', 'policy_0.pkl', 'policy_1.pkl', 'params.json'):
        assert os.path.exists(os.path.join(run_dir, name)), name
    pol = pickle.load(open(os.path.join(run_dir, 'policy_latest.pkl'), 'rb'))
    first = pol[0] if isinstance(pol, list) else pol
    assert first.info['env_name'] == 'MultiTaskFetchArm4-v5'          # play.py:27 reads it
    logger_dir = run_dir
    assert os.path.exists(os.path.join(logger_dir, 'policy_latest.pkl0_weights.pkl' if isinstance(pol, list)
                                       else 'policy_latest.pkl_weights.pkl'))


@pytest.mark.parametrize('env_name,nb,dimo,B', [('MultiTaskFetchArm4-v5', 4, 40, 48),
                                                 ('MultiTaskFetchArm8-v5', 8, 52, 48),
                                                 ('MultiTaskFetchArm8-v5', 8, 52, 37)])
@pytest.mark.parametrize('normalize', [False, True])
def test_fused_acting_equals_unfused_lean_and_generic(env_name, nb, dimo, B, normalize, route):
    """curious_policy_act_env_step == get_actions + env.step_all, bit for bit, eager and as a replayed hipGraph, on both
    env sizes and both routes: `rows` = policy_rows_kernel when B % 4 == 0 (generic kernels for B = 37); `tiled` =
    fwd_l01<1|2> + fwd_hot<DOT> + act_step<PART> when B % 16 == 0, generic kernels otherwise.  normalize: networks with
    input normalisation (--normalize_obs) -- the fused entry points then take the normalisers' statistics
    (curious_policy_*_stats) and apply them to every observation the env step hands to the next acting step."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    G = 3 * nb
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    recs = []
    for mode in ('fused_graph', 'fused_eager', 'unfused'):
        agent, _ = build_pair(nb, dimo, rng_mode='device', use_graph=(mode == 'fused_graph'), normalize_obs=normalize)
        if normalize:                                                # statistics that really move and scale the inputs
            rs = np.random.RandomState(31)
            for nz in (agent.o_stats, agent.g_stats):
                d = nz.size
                nz.state[2 * d + 1:3 * d + 1] = torch.from_numpy((rs.randn(d) * 0.2).astype(np.float32)).cuda()
                nz.state[3 * d + 1:] = torch.from_numpy((0.3 + rs.rand(d)).astype(np.float32)).cuda()
        if mode == 'unfused':
            agent.can_act_and_step = lambda env, compute_Q: False
        w = RolloutWorker(EnvFactory(env_name), agent, dims, logger, T=T, rollout_batch_size=B, noise_eps=0.2,
                          random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                          queue_length=6, eval=False)
        w.seed(5)
        np.random.seed(8)
        for _ in range(2):
            ep, _, _ = w.generate_rollouts()
        torch.cuda.synchronize()
        recs.append(ep.records.clone())
    assert torch.equal(recs[0], recs[1]) and torch.equal(recs[1], recs[2])
    off_u = w.benv.layout.off['u']
    u = recs[0][:, :T, off_u:off_u + 4]
    assert float(u.abs().max()) <= 1.0 and float(u.abs().sum()) > 0


def test_arm8_batched_rollout_matches_oracle_env_and_policy():
    """configs[2] env on the lean acting path (B = 32: fwd_l01<2>, 8 tasks): every step against the oracle env and the
    oracle policy, like test_batched_rollout_matches_oracle does for Arm4."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    from oracle.env import SyntheticMultiTaskArm
    nb, dimo, B = 8, 52, 32
    agent, oracle = build_pair(nb, dimo)
    dims = dict(o=dimo, u=4, g=24, ag=24, task_descr=nb, info_is_success=1)
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm8-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                      queue_length=4, eval=False)
    w.seed(77)
    envs = [SyntheticMultiTaskArm(nb, dimo, T, seed=77, env_id=i) for i in range(B)]
    np.random.seed(50)
    rs = np.random.get_state()
    ep, CP, n_ep = w.generate_rollouts()
    rec = {k: v.cpu().numpy() for k, v in ep.items()}
    np.random.set_state(rs)
    exploit = np.random.random() < 0.1                                 # rollout.py:183-186
    tasks = np.random.choice(range(nb), p=np.ones(nb) / nb, size=B)
    goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
    obs = []
    for i, e in enumerate(envs):
        e.reset()
        obs.append(e.reset_task_goal(goals[i], task=int(tasks[i])))
    o = np.stack([x['observation'] for x in obs])
    g = np.stack([x['desired_goal'] for x in obs])
    td = np.stack([x['mask'] for x in obs])
    np.testing.assert_array_equal(rec['o'][:, 0], o)
    for t in range(T):
        ne, re = (0., 0.) if exploit else (0.2, 0.3)
        ou = oracle.get_actions(o, o[:, :24], g, task_descr=td, noise_eps=ne, random_eps=re)
        np.testing.assert_allclose(rec['u'][:, t], ou, rtol=1e-4, atol=2e-5)
        res = [e.step(rec['u'][i, t]) for i, e in enumerate(envs)]
        o = np.stack([r[0]['observation'] for r in res])
        np.testing.assert_array_equal(rec['o'][:, t + 1], o)
        np.testing.assert_array_equal(rec['task_descr'][:, t], td)
        np.testing.assert_array_equal(rec['info_is_success'][:, t, 0],
                                      np.array([r[3]['is_success'] for r in res], np.float32))


def test_1024_env_rollout_store_and_update_properties():
    """configs[2] at its stated size: 1 024 GPU-resident Arm8 envs, CP-driven task selection, per-task CP buffers.
    Size-independent properties: record consistency (o/ag chain, change flag, one-hot task, goal on the task's slots),
    routing (every stored episode sits in the buffers of the tasks it changed), finite updates."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    nb, dimo, B = 8, 52, 1024
    agent, _ = build_pair(nb, dimo, cap_eps=4096, rng_mode='device', use_graph=True)
    dims = dict(o=dimo, u=4, g=24, ag=24, task_descr=nb, info_is_success=1)
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm8-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                      queue_length=300, eval=False)
    w.seed(3)
    np.random.seed(4)
    for cycle in range(3):
        ep, cp, n_ep = w.generate_rollouts()
        rec = {k: v.cpu().numpy() for k, v in ep.items()}
        assert rec['o'].shape == (B, T + 1, dimo) and rec['u'].shape == (B, T, 4)
        np.testing.assert_array_equal(rec['ag'], rec['o'][:, :, :24])          # achieved goal = object coordinates
        np.testing.assert_array_equal(rec['change'] != 0, np.abs(rec['ag'][:, :1] - rec['ag'][:, 1:]) > 1e-3)
        assert np.all(rec['task_descr'].sum(axis=2) == 1) and np.all(np.abs(rec['u']) <= 1)
        task = rec['task_descr'][:, 0].argmax(axis=1)
        for i in (0, 511, 1023):
            off_task = np.ones(24, bool)
            off_task[3 * task[i]:3 * task[i] + 3] = False
            assert np.all(rec['g'][i][:, off_task] == 0)
        sizes0 = [b.current_size for b in agent.buffer[1:6]]
        agent.store_episode(ep, cp, n_ep)
        active = (rec['change'][:, -1].reshape(B, nb, 3) != 0).any(axis=2)
        grew = [b.current_size - s for b, s in zip(agent.buffer[1:6], sizes0)]
        assert grew == [int(active[:, j].sum()) for j in range(5)]            # only tasks j < 5 are routed (ddpg.py:183)
        cl, _ = agent.train_batches(20)
        agent.update_target_net()
    torch.cuda.synchronize()
    assert n_ep == 3 * B and np.isfinite(float(cl)) and bool(torch.isfinite(agent.theta).all())


def test_reference_format_weights_and_pickle_round_trip(tmp_path):
    """save_weights / load_weights (ddpg.py:481-509): pickled list of lists in the order main/Q, main/pi, target/Q,
    target/pi, o_stats, g_stats with the TF variable shapes; pickle.dumps(policy) (ddpg.py:511-537) reloads a policy that
    acts identically and keeps its constructor arguments."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    agent, _ = build_pair(4, 40, rng_mode='device', use_graph=True)
    ep = synth_episodes(np.random.RandomState(5), 30, 4, 40)
    np.random.seed(1)
    agent.store_episode({k: v.copy() for k, v in ep.items()}, np.array([0.2, 0.1, 0., 0.3]), 30)
    agent.train_batches(12)
    agent.update_target_net()
    path = str(tmp_path / 'policy_latest.pkl')
    agent.save_weights(path)
    saved = pickle.load(open(path + '_weights.pkl', 'rb'))
    H, O, N, G, U = 256, 40, 4, 12, 4
    q_shapes = [(O + N + U, H), (H,), (G, H), (H, H), (H,), (H, H), (H,), (H, 1), (1,)]
    pi_shapes = [(O + N, H), (H,), (G, H), (H, H), (H,), (H, H), (H,), (H, U), (U,)]
    assert [len(x) for x in saved] == [9, 9, 9, 9, 5, 5]
    assert [a.shape for a in saved[0]] == q_shapes and [a.shape for a in saved[2]] == q_shapes
    assert [a.shape for a in saved[1]] == pi_shapes and [a.shape for a in saved[3]] == pi_shapes
    assert [a.shape for a in saved[4]] == [(O,), (O,), (1,), (O,), (O,)]      # sum, sumsq, count, mean, std
    assert [a.shape for a in saved[5]] == [(G,), (G,), (1,), (G,), (G,)]
    assert not np.array_equal(saved[0][3], saved[2][3])                       # main and target differ after training
    fresh, _ = build_pair(4, 40, rng_mode='device', use_graph=True, seed=99)
    assert not torch.equal(fresh.theta, agent.theta)
    fresh.load_weights(path)
    assert torch.equal(fresh.theta, agent.theta) and torch.equal(fresh.theta_target, agent.theta_target)
    assert torch.equal(fresh.o_stats.state, agent.o_stats.state) and torch.equal(fresh.g_stats.state, agent.g_stats.state)
    # pickled policy: same actions, constructor arguments kept (use_graph used to be dropped by a substring filter)
    clone = pickle.loads(pickle.dumps(agent))
    assert clone.use_graph is True and clone.rng_mode == 'device' and clone.scope == agent.scope
    assert clone.buffer is None if hasattr(clone, 'buffer') else True
    rng = np.random.RandomState(1)
    o, g = rng.randn(7, 40).astype(np.float32), rng.randn(7, 12).astype(np.float32)
    td = np.eye(4, dtype=np.float32)[rng.randint(4, size=7)]
    u1, q1 = agent.get_actions(o, o[:, :12], g, task_descr=td, compute_Q=True)
    u2, q2 = clone.get_actions(o, o[:, :12], g, task_descr=td, compute_Q=True)
    np.testing.assert_array_equal(u1, u2)
    np.testing.assert_array_equal(q1, q2)
    # RolloutWorker.save_policy (rollout.py:425-433) writes both files and surfaces errors
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dict(o=40, u=4, g=12, ag=12, task_descr=4,
                      info_is_success=1), logger, T=T, rollout_batch_size=4, structure='curious', eval=True)
    p2 = str(tmp_path / 'policy_best.pkl')
    w.save_policy(p2)
    assert os.path.exists(p2) and os.path.exists(p2 + '_weights.pkl')
    with pytest.raises(Exception):
        w.save_policy(str(tmp_path / 'no_such_dir' / 'x.pkl'))


def test_host_evaluated_reward_matches_the_kernel_reward():
    """Real-env adapter (config.py:158-159, her.py:166-176): a reward_fun without .spec is evaluated on the host per
    sampled batch.  With the synthetic env's own compute_reward as that callable, batches and losses must equal the
    kernel-evaluated path bit for bit; a different reward (dense negative distance) flows through unchanged."""
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import SyntheticArmEnv, sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    nb, dimo = 4, 40
    G = 12
    ag_ids, g_ids = tables(nb)
    env = SyntheticArmEnv('MultiTaskFetchArm4-v5')
    calls = []

    def host_sparse(ag_2, g, task_descr=None, info=None):
        calls.append((ag_2.dtype, g.shape, sorted(info.keys())))
        return env.compute_reward(achieved_goal=ag_2, goal=g, task_descr=task_descr, info=info)

    def host_dense(ag_2, g, task_descr=None, info=None):
        t = task_descr.argmax(axis=1)
        return -np.stack([np.linalg.norm(ag_2[i, 3 * t[i]:3 * t[i] + 3] - g[i, 3 * t[i]:3 * t[i] + 3])
                          for i in range(len(t))]).reshape(-1, 1)

    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    gamma = 1. - 1. / T
    agents = []
    for rf in (sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)), host_sparse, host_dense):
        sampler = make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', rf, tasks_ag_id=ag_ids,
                                                         tasks_g_id=g_ids)
        bufs = make_pooled_buffers(shapes, T * 64, T, sampler, nb + 1, alias_from=5)
        agents.append(DDPG(input_dims=dims, hidden=256, layers=3,
                           network_class='curious_amd.actor_critic:MultiTaskActorCritic', polyak=0.95, batch_size=256,
                           Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1., clip_obs=200.,
                           scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None, relative_goals=False,
                           clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False,
                           sample_transitions=sampler, gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids,
                           tasks_g_id=g_ids, task_replay='replay_task_cp_buffer', eps_task=0.4, structure='curious',
                           rng_mode='numpy', seed=3))
    ep = synth_episodes(np.random.RandomState(6), 30, nb, dimo)
    cp = np.array([0.3, 0.1, 0.0, 0.2])
    for a in agents:
        np.random.seed(1)
        a.store_episode({k: v.copy() for k, v in ep.items()}, cp, 30)
    outs = []
    for a in agents:
        np.random.seed(2)
        b = [x.cpu().numpy() for x in a.sample_batch()]
        np.random.seed(3)
        cl, _ = a.train()
        outs.append((b, float(cl)))
    kern, hs, hd = outs
    for x, y in zip(kern[0], hs[0]):
        np.testing.assert_array_equal(x, y)
    assert kern[1] == hs[1]
    assert calls and calls[0][0] == np.float64 and calls[0][2] == ['is_success']
    assert set(np.unique(kern[0][7])) <= {0.0, -1.0}
    for x, y in zip(kern[0][:7], hd[0][:7]):                          # everything but r is the same batch
        np.testing.assert_array_equal(x, y)
    r = hd[0][7]
    assert r.shape == (256, 1) and np.all(r <= 0) and len(np.unique(r)) > 10 and np.isfinite(hd[1])
    # ReplayBuffer.sample (replay_buffer.py:37-55) goes through the same host call
    np.random.seed(4)
    tr = agents[2].buffer[1].sample(32, task_to_replay=0)
    assert len(np.unique(tr['r'].cpu().numpy())) > 4


def test_torch_custom_op_face_matches_the_ctypes_face():
    """torch.ops.curious_hip.* (curious_amd/torch_ops.py) call the same library entry points as curious_amd.ops."""
    import curious_amd.torch_ops  # noqa: F401  (registers the ops)
    from curious_amd import ops
    agent, _ = build_pair(4, 40, rng_mode='numpy')
    ep = synth_episodes(np.random.RandomState(5), 20, 4, 40)
    np.random.seed(1)
    agent.store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(4), 20)
    np.random.seed(2)
    agent.stage_batch()
    cl, qpi, _, _ = agent._grads()
    c = agent.net_cfg
    cfg_i = [c.dimo, c.dimg, c.dimu, c.dimtd, c.hidden, c.layers, c.modular, c.clip_pos_returns, c.normalize_obs]
    cfg_f = [c.max_u, c.gamma, c.clip_return, c.action_l2, c.norm_clip]
    BL = agent._layout.c_batch_layout()
    bl = [getattr(BL, f[0]) for f in BL._fields_]
    grad = torch.zeros_like(agent.grad)
    losses, Q_pi = torch.ops.curious_hip.ddpg_grads(cfg_i, cfg_f, agent.theta, agent.theta_target, agent._staged, bl,
                                                    grad)
    assert float(losses[0]) == float(cl) and torch.equal(Q_pi, qpi) and torch.equal(grad, agent.grad)
    o = agent._staged[:, :40].contiguous()
    g = agent._staged[:, 48:60].contiguous()
    td = agent._staged[:, 40:44].contiguous()
    pi, Q = torch.ops.curious_hip.policy_forward(cfg_i, cfg_f, agent.theta, o, g, td, 200.0, True)
    u2, q2 = agent.get_actions(o, o[:, :12], g, task_descr=td, compute_Q=True)
    assert torch.equal(pi, u2) and torch.equal(Q, q2)
    tgt = agent.theta_target.clone()
    torch.ops.curious_hip.polyak_update(tgt, agent.theta, 0.95)
    agent.update_target_net()
    assert torch.equal(tgt, agent.theta_target)
    s1 = torch.ops.curious_hip.param_checksum(agent.theta)
    s2 = torch.zeros(2, dtype=torch.int64, device='cuda')
    ops.param_checksum(agent.theta, s2)
    assert torch.equal(s1, s2)
    with pytest.raises(Exception):
        torch.ops.curious_hip.polyak_update(tgt.cpu(), agent.theta.cpu(), 0.95)      # no CPU implementation


@pytest.mark.parametrize('name', ['arm4', 'arm8', 'arm4rand', 'expert2'])
def test_store_episode_and_sample_batch_match_the_reference_class(name):
    """curious_amd.DDPG.store_episode / sample_batch (numpy stream) against tests/golden/ddpg_host.npz = the outputs of the
    reference's own DDPG.store_episode / sample_batch (ddpg.py:163-223, :251-360) on the same seeds: buffer routing
    (incl. j < 5 and the aliased buffers 6..), the HER batch fed to the normalisers, proportions for three CP vectors and
    the staged arrays ag, g, o, task_descr, u, o_2, g_2, r -- bit for bit."""
    from conftest import load_golden, sub
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    G = load_golden('ddpg_host')
    cases = {'arm4': (4, 40, 'curious', 'replay_task_cp_buffer', None),
             'arm8': (8, 52, 'curious', 'replay_task_cp_buffer', None),
             'arm4rand': (4, 40, 'curious', 'replay_task_random_buffer', None),
             'expert2': (4, 40, 'task_experts', 'replay_current_task_buffer', 2)}
    Tg, B, cap = [int(x) for x in G['cfg']]
    nb, dimo, structure, tr, t_id = cases[name]
    Gd = 3 * nb
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=Gd, ag=Gd, task_descr=nb, info_is_success=1)
    shapes = dict(o=(Tg + 1, dimo), u=(Tg, 4), g=(Tg, Gd), ag=(Tg + 1, Gd), info_is_success=(Tg, 1),
                  task_descr=(Tg, nb), change=(Tg, Gd))
    sampler = make_sample_multi_task_her_transitions('her', 4, tr, sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = make_pooled_buffers(shapes, Tg * cap, Tg, sampler, nb + 1, alias_from=5)
    gamma = 1. - 1. / Tg
    agent = DDPG(input_dims=dims, hidden=64, layers=3, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                 polyak=0.95, batch_size=B, Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1.,
                 clip_obs=200., scope='ddpg', T=Tg, rollout_batch_size=2, subtract_goals=None, relative_goals=False,
                 clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False, sample_transitions=sampler,
                 gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids, tasks_g_id=g_ids, task_replay=tr, eps_task=0.4,
                 structure=structure, t_id=t_id, rng_mode='numpy', seed=1)
    ci = list(cases).index(name)
    cps = [G['%s/sample%d/cp' % (name, k)] for k in range(3)]
    for rnd in range(2):
        ep = {k: v.copy() for k, v in sub(G, '%s/store%d/in/' % (name, rnd)).items()}
        np.random.seed(100 * ci + rnd)
        agent.store_episode(ep, cps[rnd], 12 * (rnd + 1))
        np.testing.assert_array_equal([agent.buffer[i].current_size for i in range(nb + 1)],
                                      G['%s/store%d/sizes' % (name, rnd)])
        cols = agent._layout.batch_cols
        sb = agent._stats_batch.cpu().numpy().astype(np.float64)
        np.testing.assert_array_equal(sb[:, cols['o'][0]:cols['o'][0] + dimo], G['%s/store%d/stats_o' % (name, rnd)])
        np.testing.assert_array_equal(sb[:, cols['g'][0]:cols['g'][0] + Gd], G['%s/store%d/stats_g' % (name, rnd)])
    for i in range(min(nb + 1, 6)):
        n = agent.buffer[i].current_size
        v = agent.buffer[i].buffers
        np.testing.assert_array_equal(v['o'][:n].cpu().numpy().astype(np.float64), G['%s/buffer%d/o' % (name, i)])
        np.testing.assert_array_equal(v['g'][:n].cpu().numpy().astype(np.float64), G['%s/buffer%d/g' % (name, i)])
    for k, cp in enumerate(cps):
        agent.cp = cp
        np.random.seed(7000 + 10 * ci + k)
        got = agent.sample_batch()
        np.testing.assert_array_equal(agent.proportions, G['%s/sample%d/proportions' % (name, k)])
        for key, arr in zip(['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r'], got):
            np.testing.assert_array_equal(arr.cpu().numpy().astype(np.float64), G['%s/sample%d/%s' % (name, k, key)],
                                          err_msg='%s sample %d %s' % (name, k, key))


def test_action_noise_kernel_matches_the_reference_method():
    """curious_action_noise (parity mode: host-drawn randn / binomial / uniform in the reference's order) against the
    outputs of the reference's own DDPG.get_actions post-processing (tests/golden/get_actions.npz), bit for bit."""
    from conftest import load_golden
    from curious_amd import ops
    G = load_golden('get_actions')
    for n in [int(x) for x in G['ns']]:
        for tag, ne, re in (('noisy', 0.2, 0.3), ('greedy', 0.0, 0.0)):
            np.random.seed(1000 + n)
            randn = np.random.randn(n, 4)                                  # ddpg.py:149
            binom = np.random.binomial(1, re, n).astype(np.float64)       # ddpg.py:152
            unif = np.random.uniform(low=-1.0, high=1.0, size=(n, 4))     # ddpg.py:114-115
            assert float(np.random.uniform()) == float(G['n%d/%s/next_uniform' % (n, tag)])
            u = torch.from_numpy(G['n%d/pi' % n].copy()).cuda()
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            ops.action_noise(u, n, 4, ne * 1.0, re, 1.0, d(randn.reshape(-1)), d(binom), d(unif.reshape(-1)))
            np.testing.assert_array_equal(u.cpu().numpy().reshape(G['n%d/%s/u' % (n, tag)].shape),
                                          G['n%d/%s/u' % (n, tag)])


@pytest.mark.parametrize('env_name,nb,dimo,B,layers', [('MultiTaskFetchArm4-v5', 4, 40, 64, 3),
                                                        ('MultiTaskFetchArm8-v5', 8, 52, 64, 3),
                                                        ('MultiTaskFetchArm4-v5', 4, 40, 30, 3),
                                                        ('MultiTaskFetchArm4-v5', 4, 40, 48, 2),   # one hidden matrix: no x2
                                                        ('MultiTaskFetchArm4-v5', 4, 40, 64, 4)])  # 3 hidden: streaming kernel
def test_rollout_entry_point_equals_one_launch_per_step(env_name, nb, dimo, B, layers, route):
    """curious_policy_rollout (all T steps in one launch on the row-local route, B % 4 == 0 -- with 2 or 3 layers the
    weights-resident kernel, groups of 4 workgroups per 4 envs; the launches of the single-step entry point otherwise)
    == T x curious_policy_act_env_step, bit for bit: episode records, last actions, success / NaN flags, final env state
    -- over two consecutive episodes (the episode counter feeds the env's streams)."""
    from curious_amd import ops
    from curious_amd.envs import EnvFactory, REWARD_EPS
    outs = []
    for mode in ('rollout', 'steps'):
        agent, _ = build_pair(nb, dimo, rng_mode='device', use_graph=False, layers=layers)
        env = EnvFactory(env_name).make_batched(B)
        env.seed(11)
        rs = np.random.RandomState(3)
        ws = torch.empty(ops.workspace_floats(agent.net_cfg, B), dtype=torch.float32, device='cuda')
        u = torch.zeros([B, 4], dtype=torch.float32, device='cuda')
        seed, recs = 12345, []
        for ep in range(2):
            env.reset_all(rs.randint(0, nb, B), rs.uniform(-1, 1, (B, 3)).astype(np.float32))
            args = (agent.net_cfg, agent.theta, B, agent.clip_obs, ws, 0.2, 0.3, seed)
            tail = (env.o, env.ag, env.g, env.td, env.staging, REWARD_EPS)
            if mode == 'rollout':
                ops.policy_rollout(*args, 1 + ep * T, u, env._cfg, env.layout, env.env_id0, env.episode, env.tasks, 0, T,
                                   *tail, flags=env.flags)
            else:
                for t in range(T):
                    ops.policy_act_env_step(*args, 1 + ep * T + t, u, env._cfg, env.layout, env.env_id0, env.episode,
                                            env.tasks, t, *tail, flags=env.flags)
            torch.cuda.synchronize()
            recs.append((env.staging.clone(), u.clone(), env.flags.clone(), env.o.clone(), env.ag.clone()))
        outs.append(recs)
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    off_u = env.layout.off['u']
    acts = outs[0][1][0][:, :T, off_u:off_u + 4]
    assert float(acts.abs().max()) <= 1.0 and float(acts.abs().sum()) > 0


@pytest.mark.parametrize('case', range(int(os.environ.get('CURIOUS_FUZZ_ROLLOUT', 8))))
def test_rollout_entry_point_random_sizes(case, monkeypatch):
    """Seeded sweep of curious_policy_rollout against one launch per step, bit for bit, over the sizes that decide its
    route: 4-300 envs (the weights-resident kernel needs n % 4 == 0 and n <= CUs; 260 and 300 envs stream, 30 and 37 take
    the generic launches), 2-4 layers, Arm4 / Arm8 and synthetic arms of 3 / 5 / 6 / 7 tasks whose widths are no multiple of
    4, exploration noise on and off."""
    from curious_amd import ops
    from curious_amd.envs import EnvFactory, REWARD_EPS
    from curious_amd import envs as envs_mod
    # synthetic arms of other widths than the two named configurations (widths that are no multiple of 4 included)
    extra = {'SynthArm3': (3, 21, 50), 'SynthArm5': (5, 43, 50), 'SynthArm6': (6, 33, 50), 'SynthArm7': (7, 46, 50)}
    monkeypatch.setattr(envs_mod, 'ENV_CONFIGS', dict(envs_mod.ENV_CONFIGS, **extra))
    rs0 = np.random.RandomState(2100 + case)
    shapes = [('MultiTaskFetchArm4-v5', 4, 40), ('MultiTaskFetchArm8-v5', 8, 52)] + [(k, v[0], v[1]) for k, v in extra.items()]
    env_name, nb, dimo = shapes[int(rs0.randint(0, len(shapes)))]
    B = int(rs0.choice([4, 8, 12, 30, 37, 100, 252, 256, 260, 300]))
    layers = int(rs0.choice([2, 3, 3, 4]))
    noise, reps = [(0.2, 0.3), (0.0, 0.0), (0.05, 1.0)][int(rs0.randint(0, 3))]
    normalize = bool(rs0.randint(0, 2))
    outs = []
    for mode in ('rollout', 'steps'):
        agent, _ = build_pair(nb, dimo, rng_mode='device', use_graph=False, layers=layers, normalize_obs=normalize)
        stats = {}
        if normalize:
            rs1 = np.random.RandomState(77 + case)
            for nz in (agent.o_stats, agent.g_stats):
                d = nz.size
                nz.state[2 * d + 1:3 * d + 1] = torch.from_numpy((rs1.randn(d) * 0.2).astype(np.float32)).cuda()
                nz.state[3 * d + 1:] = torch.from_numpy((0.3 + rs1.rand(d)).astype(np.float32)).cuda()
            stats = dict(o_stats=agent.o_stats.state, g_stats=agent.g_stats.state)
        env = EnvFactory(env_name).make_batched(B)
        env.seed(11 + case)
        rs = np.random.RandomState(3)
        ws = torch.zeros(ops.workspace_floats(agent.net_cfg, B), dtype=torch.float32, device='cuda')
        u = torch.zeros([B, 4], dtype=torch.float32, device='cuda')
        recs = []
        for ep in range(2):
            env.reset_all(rs.randint(0, nb, B), rs.uniform(-1, 1, (B, 3)).astype(np.float32))
            args = (agent.net_cfg, agent.theta, B, agent.clip_obs, ws, noise, reps, 999 + case)
            tail = (env.o, env.ag, env.g, env.td, env.staging, REWARD_EPS)
            if mode == 'rollout':
                ops.policy_rollout(*args, 1 + ep * T, u, env._cfg, env.layout, env.env_id0, env.episode, env.tasks, 0, T,
                                   *tail, flags=env.flags, **stats)
            else:
                for t in range(T):
                    ops.policy_act_env_step(*args, 1 + ep * T + t, u, env._cfg, env.layout, env.env_id0, env.episode,
                                            env.tasks, t, *tail, flags=env.flags, **stats)
            torch.cuda.synchronize()
            assert float(env.flags[B]) == 0.0                        # neither a NaN word nor a member that gave up (2)
            recs.append((env.staging.clone(), u.clone(), env.flags.clone(), env.o.clone(), env.ag.clone()))
        outs.append(recs)
    tag = 'case %d: %s B %d layers %d noise %s normalize %s' % (case, env_name, B, layers, (noise, reps), normalize)
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y), tag


@pytest.mark.parametrize('use_graph', [False, True])
def test_async_store_equals_the_host_routed_cycle(use_graph):
    """async_store: the cycle rollout -> store_episode -> train_batches with the episodes routed on the device and the
    rollout flags read one cycle late == the cycle that waits for the flags and routes on the host, bit for bit -- replay
    storage, buffer sizes, sampling tables, parameters, competence state -- through the phases in which the async form
    does not apply (task buffers still empty, exploit rollouts) and the ones in which it does, full buffers included
    (random slots: the same Philox draws on the host and on the device)."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    nb, dimo, B = 4, 40, 16
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    res, used = [], []
    for async_store in (False, True):
        agent, _ = build_pair(nb, dimo, cap_eps=100, rng_mode='device', use_graph=use_graph)
        agent.async_store = async_store
        w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                          noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                          queue_length=6, eval=False)
        w.seed(5)
        np.random.seed(8)
        snaps, n_async = [], 0
        for c in range(14):
            if c == 3:                                               # from here on every task buffer holds episodes
                agent.store_episode(synth_episodes(np.random.RandomState(21), 24, nb, dimo), w.CP, w.n_episodes)
            ep, cp, n_ep = w.generate_rollouts()
            n_async += int(getattr(w, '_pending', None) is not None)
            agent.store_episode(ep, cp, n_ep)
            agent.train_batches(12)
            agent.update_target_net()
            if c in (3, 7, 13):
                w.settle()
                agent.settle()
                torch.cuda.synchronize()
                nb1 = nb + 1
                snaps.append(dict(sizes=[agent.buffer[i].current_size for i in range(nb1)],
                                  stored=[agent.buffer[i].n_transitions_stored for i in range(nb1)],
                                  storage=[agent._pool.storage[agent.buffer[i].pool_index, :agent.buffer[i].current_size]
                                           .clone() for i in range(nb1)], theta=agent.theta.clone(),
                                  target=agent.theta_target.clone(), tables=agent._tables.clone(),
                                  cp=np.asarray(w.CP).copy(), p=np.asarray(w.p).copy(), n_ep=w.n_episodes,
                                  succ=list(w.success_history), tasks=list(w.task_history),
                                  o_stats=agent.o_stats.state.clone()))
        res.append(snaps)
        used.append(n_async)
    assert used[0] == 0 and 5 <= used[1] <= 12                   # the async form ran, and not in every cycle
    for i, (a, b) in enumerate(zip(res[0], res[1])):
        assert a['sizes'] == b['sizes'] and a['stored'] == b['stored'] and a['n_ep'] == b['n_ep'], i
        assert a['succ'] == b['succ'] and a['tasks'] == b['tasks'], i
        assert np.array_equal(a['cp'], b['cp']) and np.array_equal(a['p'], b['p']), i
        for k in ('theta', 'target', 'tables', 'o_stats'):
            assert torch.equal(a[k], b[k]), (i, k)
        assert all(torch.equal(x, y) for x, y in zip(a['storage'], b['storage'])), i     # the filled slots
    assert max(res[1][-1]['sizes'][1:]) == 100                   # a full buffer: random slots, async and host-routed alike


@pytest.mark.parametrize('E,cap', [(300, 150), (1500, 400), (2048, 5000)])
def test_route_store_kernel_matches_the_host_routing(E, cap):
    """curious_route_store_episodes against the host routing of DDPG.store_episode (fits case) for random activity
    patterns: pair list, new sizes, stored records; 8 tasks (only the first 5 are routed), 300 episodes (two scan passes)
    up to the 2048 the entry point accepts, buffers that overflow into random slots with many / few collisions inside the
    batch; and the NaN word that makes it store nothing."""
    from curious_amd import ops
    from curious_amd.layout import RecordLayout
    rng = np.random.RandomState(4)
    nb, Tn = 8, 5                                                 # 150 slots: ~120 routed episodes per task overflow it
    shapes = dict(o=(Tn + 1, 6), u=(Tn, 4), g=(Tn, 3), ag=(Tn + 1, 3), info_is_success=(Tn, 1), task_descr=(Tn, nb),
                  change=(Tn, 3))
    lay = RecordLayout(shapes, Tn)
    dev = torch.device('cuda', 0)
    staging = torch.randn([E, Tn + 1, lay.row_stride], device=dev)
    active = torch.from_numpy((rng.rand(E, nb) < 0.4).astype(np.int32)).to(dev)
    alias = torch.tensor([0, 1, 2, 3, 4, 5, 5, 5, 5], dtype=torch.int32, device=dev)
    for skip_val in (0.0, 1.0):
        storage = torch.zeros([6, cap, Tn + 1, lay.row_stride], device=dev)
        cur0 = rng.randint(cap - 149, cap - 100, nb + 1).astype(np.int32)
        cur = torch.from_numpy(cur0.copy()).to(dev)
        src = torch.zeros(E * 5, dtype=torch.int32, device=dev)
        dst = torch.zeros(E * 5, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        skip = torch.tensor([skip_val], device=dev)
        ops.route_store_episodes(storage, staging, lay, active.reshape(-1), nb, 5, E, cur, alias, cap, 77, 3, skip, src,
                                 dst, cnt)
        torch.cuda.synchronize()
        a = active.cpu().numpy().astype(bool)
        want_src, want_dst, want_cur, overflowed = [], [], cur0.copy(), 0
        ref = torch.zeros([6 * cap, Tn + 1, lay.row_stride], device=dev)
        if skip_val == 0.0:
            for j in range(5):
                eps = np.nonzero(a[:, j])[0]
                free = max(0, cap - cur0[1 + j])
                slots = np.arange(cur0[1 + j], cur0[1 + j] + min(eps.size, free)).tolist()
                if eps.size > free:                              # the rule of replay_buffer.py:90-109, episode by episode
                    slots += ops.store_slots_host(77, 3, j, cap, eps[free:]).tolist()
                    overflowed += 1
                want_cur[1 + j] = min(cap, cur0[1 + j] + eps.size)
                for e, sl in zip(eps.tolist(), slots):           # sequential: the later writer of a slot wins
                    ref[sl + int(alias[1 + j]) * cap] = staging[e]
                want_src += eps.tolist()
                want_dst += [sl + int(alias[1 + j]) * cap for sl in slots]
            assert overflowed >= 3
        k = int(cnt)
        assert k == len(want_src)
        got_src, got_dst = src[:k].cpu().tolist(), dst[:k].cpu().tolist()
        assert got_dst == want_dst and all(g in (w, -1) for g, w in zip(got_src, want_src))
        alive = [d for g, d in zip(got_src, got_dst) if g >= 0]
        assert len(alive) == len(set(alive)) == len(set(want_dst))         # one surviving pair per destination
        assert cur.cpu().numpy().tolist() == want_cur.tolist()
        assert torch.equal(storage.reshape(6 * cap, Tn + 1, lay.row_stride), ref)


@pytest.mark.parametrize('case', range(int(os.environ.get('CURIOUS_FUZZ_ROUTE', 10))))
def test_route_store_kernel_random_shapes(case):
    """Seeded sweep of curious_route_store_episodes against the sequential rule of ddpg.py:178-197 +
    replay_buffer.py:90-109: 1-8 tasks (only the first 5 routed, buffers 6.. alias buffer 5), 1-2 048 episodes, capacities
    from a handful of slots (every episode fights for one) to room for all, buffers empty / half full / full, sparse to dense
    activity: pair list, sizes and the stored records."""
    from curious_amd import ops
    from curious_amd.layout import RecordLayout
    rng = np.random.RandomState(3100 + case)
    nb = int(rng.randint(1, 9))
    nr = min(nb, 5)
    E = int(rng.choice([1, 2, 7, 255, 256, 257, 700, 2048]))
    cap = int(rng.choice([3, 17, 200, 3000, 30000]))
    Tn = 3
    dimg = 3 * nb
    shapes = dict(o=(Tn + 1, 6), u=(Tn, 4), g=(Tn, dimg), ag=(Tn + 1, dimg), info_is_success=(Tn, 1),
                  task_descr=(Tn, nb), change=(Tn, dimg))
    lay = RecordLayout(shapes, Tn)
    dev = torch.device('cuda', 0)
    staging = torch.randn([E, Tn + 1, lay.row_stride], device=dev)
    active_np = (rng.rand(E, nb) < rng.choice([0.02, 0.4, 1.0])).astype(np.int32)
    active = torch.from_numpy(active_np).to(dev)
    alias_np = np.array([min(i, 5) for i in range(nb + 1)], np.int32)
    nphys = int(alias_np.max()) + 1
    alias = torch.from_numpy(alias_np).to(dev)
    fill = rng.choice(['empty', 'half', 'full', 'mixed'])
    cur0 = dict(empty=np.zeros(nb + 1), half=np.full(nb + 1, cap // 2), full=np.full(nb + 1, cap),
                mixed=rng.randint(0, cap + 1, nb + 1))[fill].astype(np.int32)
    for i in range(6, nb + 1):
        cur0[i] = cur0[5]                                            # logical buffers 6.. ARE buffer 5
    storage = torch.zeros([nphys, cap, Tn + 1, lay.row_stride], device=dev)
    cur = torch.from_numpy(cur0.copy()).to(dev)
    src = torch.zeros(max(1, E * nr), dtype=torch.int32, device=dev)
    dst = torch.zeros(max(1, E * nr), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    skip = torch.zeros(1, device=dev)
    seed, call = 1234567 + case, 5 + case
    ops.route_store_episodes(storage, staging, lay, active.reshape(-1), nb, nr, E, cur, alias, cap, seed, call, skip, src,
                             dst, cnt)
    torch.cuda.synchronize()
    a = active_np.astype(bool)
    want_src, want_dst, want_cur = [], [], cur0.copy()
    ref = torch.zeros([nphys * cap, Tn + 1, lay.row_stride], device=dev)
    for j in range(nr):
        eps = np.nonzero(a[:, j])[0]
        free = max(0, cap - int(cur0[1 + j]))
        slots = np.arange(cur0[1 + j], cur0[1 + j] + min(eps.size, free)).tolist()
        if eps.size > free:
            slots += ops.store_slots_host(seed, call, j, cap, eps[free:]).tolist()
        want_cur[1 + j] = min(cap, int(cur0[1 + j]) + eps.size)
        for e, sl in zip(eps.tolist(), slots):
            ref[sl + int(alias_np[1 + j]) * cap] = staging[e]
        want_src += eps.tolist()
        want_dst += [sl + int(alias_np[1 + j]) * cap for sl in slots]
    k = int(cnt)
    tag = 'case %d: nb %d E %d cap %d %s' % (case, nb, E, cap, fill)
    assert k == len(want_src), tag
    got_src, got_dst = src[:k].cpu().tolist(), dst[:k].cpu().tolist()
    assert got_dst == want_dst and all(g in (w, -1) for g, w in zip(got_src, want_src)), tag
    alive = [d for g, d in zip(got_src, got_dst) if g >= 0]
    assert len(alive) == len(set(alive)) == len(set(want_dst)), tag
    assert cur.cpu().numpy()[1:1 + nr].tolist() == want_cur[1:1 + nr].tolist(), tag
    assert torch.equal(storage.reshape(nphys * cap, Tn + 1, lay.row_stride), ref), tag
