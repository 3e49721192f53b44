"""Agent-level parity on a real MI355X: curious_amd.DDPG / ReplayBuffer / RolloutWorker against the oracle on
identical seeds (bit-exact sampling, relabelling and rewards; losses within 1e-5 relative)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = 50


def tables(nb):
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb)]
    return ids, [list(x) for x in ids]


def synth_episodes(rng, E, nb, dimo):
    G = 3 * nb
    o = np.empty([E, T + 1, dimo], np.float32)
    o[:, 0] = rng.randn(E, dimo).astype(np.float32)
    steps = (0.01 * rng.randn(E, T, dimo)).astype(np.float32)
    frozen = rng.rand(E, 1, dimo) < 0.5
    steps = np.where(frozen, np.float32(0), steps)
    for t in range(T):
        o[:, t + 1] = o[:, t] + steps[:, t]
    ag = o[:, :, :G].copy()
    task = rng.randint(nb, size=E)
    td = np.zeros([E, T, nb], np.float32)
    td[np.arange(E), :, task] = 1
    g = np.zeros([E, T, G], np.float32)
    for e in range(E):
        sl = slice(3 * task[e], 3 * task[e] + 3)
        g[e, :, sl] = ag[e, rng.randint(T + 1), sl] + (0.03 * rng.randn(3)).astype(np.float32)
    u = rng.uniform(-1, 1, [E, T, 4]).astype(np.float32)
    change = (np.abs(ag[:, :1] - ag[:, 1:]) > 1e-3)
    succ = rng.randint(2, size=[E, T, 1]).astype(np.float32)
    return dict(o=o, u=u, g=g, ag=ag, task_descr=td, change=change, info_is_success=succ)


def build_pair(nb, dimo, cap_eps=64, batch_size=256, seed=3, rng_mode='numpy', use_graph=False, hidden=256, layers=3,
               normalize_obs=False, relative_goals=False):
    """(product agent, oracle agent) with identical weights and empty buffers (normalize_obs: the product agent only --
    the oracle agent has no input normalisation; its networks do, tests/test_gpu_kernels.py)."""
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    G = 3 * nb
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1),
                  task_descr=(T, nb), change=(T, G))
    sampler = make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer',
                                                     sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = make_pooled_buffers(shapes, T * cap_eps, T, sampler, nb + 1, alias_from=5)
    gamma = 1. - 1. / T
    agent = DDPG(input_dims=dims, hidden=hidden, layers=layers, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                 polyak=0.95, batch_size=batch_size, Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1.,
                 action_l2=1., clip_obs=200., scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None,
                 relative_goals=relative_goals, clip_pos_returns=True, clip_return=1. / (1. - gamma),
                 normalize_obs=normalize_obs,
                 sample_transitions=sampler, gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids, tasks_g_id=g_ids,
                 task_replay='replay_task_cp_buffer', eps_task=0.4, structure='curious', rng_mode=rng_mode, seed=seed,
                 use_graph=use_graph)
    osampler = oher.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer',
                                                           make_reward_fun(ag_ids, g_ids), tasks_ag_id=ag_ids,
                                                           tasks_g_id=g_ids)
    obufs = [OBuf(shapes, T * cap_eps, T, osampler) for _ in range(nb + 1)]
    oracle = OracleDDPG(dims, T, obufs, osampler, ag_ids, g_ids, hidden=hidden, layers=layers, batch_size=batch_size,
                        weight_rng=np.random.RandomState(seed), relative_goals=relative_goals)
    return agent, oracle


def theta_of(agent):
    from curious_amd import ops
    return ops.unpad_params(agent.net_cfg, agent.theta.cpu().numpy())


@pytest.mark.parametrize('nb,dimo', [(4, 40), (8, 52)])
def test_store_sample_train_match_oracle(nb, dimo):
    agent, oracle = build_pair(nb, dimo)
    np.testing.assert_array_equal(theta_of(agent), oracle.theta)
    rng = np.random.RandomState(5)
    cp = rng.rand(nb) * (rng.rand(nb) > 0.3)
    # ---- store: 3 batches, the last ones overflow the 64-episode buffers -> random slots from the NumPy stream
    for k in range(3):
        ep = synth_episodes(rng, 40, nb, dimo)
        np.random.seed(100 + k)
        agent.store_episode({k2: v.copy() for k2, v in ep.items()}, cp, 40 * (k + 1))
        np.random.seed(100 + k)
        oracle.store_episode({k2: v.astype(np.float64) for k2, v in ep.items()}, cp, 40 * (k + 1))
    for i in range(1, nb + 1):
        assert agent.buffer[i].current_size == oracle.buffer[i].current_size
        assert agent.buffer[i].n_transitions_stored == oracle.buffer[i].n_transitions_stored
        E = oracle.buffer[i].current_size
        for key, v in agent.buffer[i].buffers.items():
            np.testing.assert_array_equal(v[:E].cpu().numpy().astype(np.float64), oracle.buffer[i].buffers[key][:E],
                                          err_msg='buffer %d key %s' % (i, key))
    assert agent.buffer[0].current_size == 0                          # buffer 0 is never written (ddpg.py:191-192)
    if nb == 8:
        assert agent.buffer[6] is agent.buffer[5] and agent.buffer[8] is agent.buffer[5]
        assert agent.buffer[5].current_size > 0                       # task 4 -> buffer 5; tasks >= 5 never routed
    # normaliser statistics (float64 tree sum vs NumPy's sequential sum: 1e-6 relative)
    for nz, onz in ((agent.o_stats, oracle.o_stats), (agent.g_stats, oracle.g_stats)):
        np.testing.assert_allclose(nz.mean.cpu().numpy(), onz.mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nz.std.cpu().numpy(), onz.std, rtol=1e-5, atol=1e-6)
        assert float(nz.state[2 * nz.size]) == float(onz.count[0])
    # ---- sample_batch: bit-exact (indices, relabels, rewards, clip, shuffle)
    np.random.seed(7)
    got = [x.cpu().numpy() for x in agent.sample_batch()]
    np.random.seed(7)
    want = oracle.sample_batch()
    np.testing.assert_array_equal(agent.proportions, oracle.proportions)
    assert (agent.proportions > 0).sum() >= 2                         # really a multi-buffer mix
    for name, a, b in zip(['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r'], got, want):
        np.testing.assert_array_equal(a.astype(np.float64), np.asarray(b, dtype=np.float64), err_msg=name)
    # ---- train: losses within 1e-5 relative, parameters stay together
    np.random.seed(11)
    o_losses = []
    for _ in range(4):
        ql, qpi = oracle.train()
        o_losses.append((float(ql), qpi.copy()))
    oracle.update_target_net()
    np.random.seed(11)
    for k in range(4):
        cl, al = agent.train()
        assert abs(float(cl) - o_losses[k][0]) <= 1e-5 * abs(o_losses[k][0]), (k, float(cl), o_losses[k][0])
        np.testing.assert_allclose(al.cpu().numpy(), o_losses[k][1], rtol=1e-4, atol=1e-5)
    agent.update_target_net()
    th = theta_of(agent)
    assert np.abs(th - oracle.theta).max() <= 2e-5                    # 4 Adam steps of size 1e-3
    from curious_amd import ops
    np.testing.assert_allclose(ops.unpad_params(agent.net_cfg, agent.theta_target.cpu().numpy()), oracle.theta_target,
                               rtol=0, atol=2e-6)
    assert agent.Q_adam.t == 4 and agent.pi_adam.t == 4


def test_replay_buffer_class_matches_golden():
    """curious_amd.ReplayBuffer + sampler as standalone objects against the reference's golden vectors."""
    from conftest import load_golden, sub
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import ReplayBuffer
    G = load_golden('replay_buffer')
    nb, dimo, Tg, cap, seed = [int(x) for x in G['cfg']]
    ag_ids, g_ids = tables(nb)
    fn = make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer',
                                                sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    shapes = dict(o=(Tg + 1, dimo), u=(Tg, 4), g=(Tg, 12), ag=(Tg + 1, 12), task_descr=(Tg, nb), change=(Tg, 12),
                  info_is_success=(Tg, 1))
    rb = ReplayBuffer(shapes, Tg * cap, Tg, fn)
    np.random.seed(seed)
    for step, inc in enumerate(G['incs']):
        rb.store_episode(sub(G, 'step%d/in/' % step))
        assert rb.get_current_episode_size() == int(G['step%d/current_size' % step])
        assert rb.get_transitions_stored() == int(G['step%d/n_stored' % step])
        np.testing.assert_array_equal(rb.buffers['o'][:rb.current_size, 0, :3].cpu().numpy().astype(np.float64),
                                      G['step%d/o_rows' % step])
    assert rb.full
    tr = rb.sample(64, task_to_replay=1)
    want = sub(G, 'sample/')
    assert set(want.keys()) <= set(tr.keys())
    for k in want:
        np.testing.assert_array_equal(tr[k].cpu().numpy().astype(np.float64).reshape(want[k].shape),
                                      want[k].astype(np.float64), err_msg=k)
    for k, v in rb.buffers.items():
        np.testing.assert_array_equal(v[:rb.current_size].cpu().numpy().astype(np.float64), G['final/' + k])


def test_get_actions_matches_oracle_numpy_stream():
    agent, oracle = build_pair(4, 40)
    rng = np.random.RandomState(0)
    for n in (1, 2, 17):
        o = (rng.randn(n, 40) * 2).astype(np.float32)
        ag = rng.randn(n, 12).astype(np.float32)
        g = rng.randn(n, 12).astype(np.float32)
        td = np.eye(4, dtype=np.float32)[rng.randint(4, size=n)]
        np.random.seed(n)
        u, Q = agent.get_actions(o, ag, g, task_descr=td, noise_eps=0.2, random_eps=0.3, compute_Q=True)
        s1 = np.random.uniform()
        np.random.seed(n)
        ou, oQ = oracle.get_actions(o, ag, g, task_descr=td, noise_eps=0.2, random_eps=0.3, compute_Q=True)
        s2 = np.random.uniform()
        assert s1 == s2                                               # same amount of stream consumed
        assert u.shape == ou.shape                                    # 1-D when n == 1 (ddpg.py:153-154)
        np.testing.assert_allclose(u, ou, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(Q, oQ, rtol=1e-4, atol=1e-5)


def test_device_rng_graph_equals_eager_and_learns():
    """Throughput mode: hipGraph replay of [sample -> grads -> Adam] is bit-identical to eager launches."""
    a_graph, _ = build_pair(4, 40, rng_mode='device', use_graph=True)
    a_eager, _ = build_pair(4, 40, rng_mode='device', use_graph=False)
    rng = np.random.RandomState(9)
    cp = np.array([0.5, 0.2, 0.0, 0.1])
    ep = synth_episodes(rng, 48, 4, 40)
    for a in (a_graph, a_eager):
        np.random.seed(1)
        a.store_episode({k: v.copy() for k, v in ep.items()}, cp, 48)
    first = None
    for k in range(30):
        le, _ = a_eager.train()
        if first is None:
            first = float(le)
    # 7 single-update graphs (leaves the staging parity odd), then train_batches: one more single update to get back to
    # parity 0, then ONE chained graph of the 22 remaining updates
    for k in range(7):
        a_graph.train()
    lg, _ = a_graph.train_batches(23)
    torch.cuda.synchronize()
    assert list(a_graph._chains) == [22] and all(g is not None for g in a_graph._graphs)
    assert torch.equal(a_graph.theta, a_eager.theta)
    assert torch.equal(a_graph._m, a_eager._m) and torch.equal(a_graph._v, a_eager._v)
    assert torch.equal(a_graph._staged, a_eager._staged)             # the batch of update 31 is already staged
    assert float(lg) == float(le)
    assert np.isfinite(float(le)) and float(le) < first              # critic loss goes down on a fixed buffer
    assert int(a_graph._step_ctr) == 30 == a_graph.Q_adam.t
    # the device-drawn batch is a valid sample: rows come from the stored episodes, rewards are 0/-1
    b = a_eager.sample_batch()
    r = b[7].cpu().numpy()
    assert set(np.unique(r)) <= {0.0, -1.0}
    td = b[3].cpu().numpy()
    assert np.all(td.sum(axis=1) == 1)


def _fused_vs_unfused(nb, dimo, kw, n_updates=9):
    a_f, _ = build_pair(nb, dimo, rng_mode='device', use_graph=False, **kw)
    a_u, _ = build_pair(nb, dimo, rng_mode='device', use_graph=False, **kw)
    rng = np.random.RandomState(5)
    cp = np.array([0.3, 0.0, 0.2, 0.1] * 3)[:nb]
    ep = synth_episodes(rng, 40, nb, dimo)
    for a in (a_f, a_u):
        np.random.seed(2)
        a.store_episode({k: v.copy() for k, v in ep.items()}, cp, 40)
    if kw.get('normalize_obs'):
        assert float(a_f.o_stats.mean.abs().sum()) > 0 and a_f.normalize_obs      # non-trivial statistics feed the nets
    for k in range(n_updates):
        lf, qf = a_f.train()
        if a_u._tables_dirty:
            a_u._refresh_device_tables()
        if a_u._alpha_filled == 0:
            a_u._fill_alpha_table()
        batch_u = a_u._sample_packed().clone()
        lu, qu, _, _ = a_u._grads()
        a_u._update(use_table=True)
        torch.cuda.synchronize()
        assert torch.equal(a_f._pp[k & 1], batch_u)                  # update k consumed the same batch
        assert torch.equal(a_f.grad, a_u.grad)
        assert torch.equal(a_f.theta, a_u.theta) and torch.equal(a_f._m, a_u._m) and torch.equal(a_f._v, a_u._v)
        assert float(lf) == float(lu) and torch.equal(qf, qu)
    assert float(a_f.theta.abs().sum()) != float(a_f.theta_target.abs().sum())   # the parameters did move


@pytest.mark.parametrize('shape', ['default', 'small', 'layers4', 'layers2', 'normalize'])
def test_fused_update_equals_unfused_sequence(shape, route):
    """curious_ddpg_update (Adam in the weight-gradient launch + next gather riding along) against the three separate
    launches it replaces -- her_sample, ddpg_grads, adam_update -- bit for bit: lean kernels, the generic fallback, and
    4 layers per network (the row-local pass with the generic weight gradients + the stand-alone optimiser)."""
    kw = dict(default={}, small=dict(batch_size=64, hidden=64), layers4=dict(layers=4), layers2=dict(layers=2),
              normalize=dict(normalize_obs=True))[shape]   # --normalize_obs: statistics of the stored episodes feed the nets
    _fused_vs_unfused(4, 40, kw)


@pytest.mark.parametrize('case', range(int(os.environ.get('CURIOUS_FUZZ_UPDATE', 8))))
def test_fused_update_random_shapes(case):
    """The same identity over seeded random agents: 1-10 tasks, observations of 31-90 floats, batches of 64-512, hidden
    64 / 128 / 256, 2-4 layers, input normalisation on / off -- whichever kernels the shape selects for the fused form
    (row-local + lean tail, row-local + generic tail + stand-alone Adam, tiled) against the unfused sequence."""
    rs = np.random.RandomState(4100 + case)
    nb = int(rs.randint(1, 11))
    dimo = int(rs.randint(3 * nb + 4, 3 * nb + 61))                  # the synthetic episodes keep ag = o[:3 nb]
    kw = dict(batch_size=int(rs.choice([64, 128, 256, 512])), hidden=int(rs.choice([64, 128, 256, 256])),
              layers=int(rs.randint(2, 5)), normalize_obs=bool(rs.randint(0, 2)))
    _fused_vs_unfused(nb, dimo, kw, n_updates=5)


def test_step_size_ring_refill_under_chained_graphs():
    """4 200 updates cross the 4 096-entry ring of Adam step sizes: chained hipGraph replay stays bit-identical to
    eager launches across the refill (and to the float64 host formula of mpi_adam.py:30 at the end)."""
    a_graph, _ = build_pair(4, 40, rng_mode='device', use_graph=True, batch_size=64, hidden=64)
    a_eager, _ = build_pair(4, 40, rng_mode='device', use_graph=False, batch_size=64, hidden=64)
    rng = np.random.RandomState(3)
    ep = synth_episodes(rng, 32, 4, 40)
    for a in (a_graph, a_eager):
        np.random.seed(4)
        a.store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(4), 32)
    a_graph.train_batches(4200)
    for _ in range(4200):
        a_eager.train()
    torch.cuda.synchronize()
    assert a_graph.Q_adam.t == a_eager.Q_adam.t == 4200 == int(a_graph._step_ctr)
    assert torch.equal(a_graph.theta, a_eager.theta)
    assert torch.isfinite(a_graph.theta).all()
    # the ring was refilled BEHIND a run of updates (DDPG._keep_alpha_ahead), never in front of one: the entries the last
    # updates read are the float64 host formula rounded to float32
    from curious_amd.ddpg import ALPHA_TAB
    tab = a_graph._alpha_tab.cpu().numpy()
    for t in (4097, 4150, 4200):
        assert tab[(t - 1) % ALPHA_TAB, 0] == np.float32(a_graph.Q_adam.alpha(a_graph.Q_lr, t))
        assert tab[(t - 1) % ALPHA_TAB, 1] == np.float32(a_graph.pi_adam.alpha(a_graph.pi_lr, t))
    assert a_graph._alpha_filled >= 4200 + 100


def test_rank_paths_match_single_rank():
    """The multi-rank update paths (split graphs + eager RCCL all-reduce; all-reduce captured in the graph), run on
    a one-rank RCCL communicator, leave bit-identical parameters to the fused single-rank path."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    # eager collective forced, captured collective forced, and the default: decided by the collective self-test of
    # curious_amd.dist.captured_allreduce_ok
    for extra in ({}, {'CURIOUS_FORCE_DIST': '1', 'CURIOUS_GRAPH_ALLREDUCE': '0'},
                  {'CURIOUS_FORCE_DIST': '1', 'CURIOUS_GRAPH_ALLREDUCE': '1'}, {'CURIOUS_FORCE_DIST': '1'}):
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.pop('CURIOUS_GRAPH_ALLREDUCE', None)
        env.update(extra)
        out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rank_path_check.py')], env=env, cwd=root,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('DIGEST')][-1].split()
        assert line[2] == '35'
        digests.append(line[1])
    assert digests[0] == digests[1] == digests[2] == digests[3], digests


def test_two_ranks_share_one_gpu_over_gloo():
    """World size 2 on ONE GPU (gloo carries the collectives; RCCL refuses two ranks per device): private buffers and
    RNG streams per rank, summed gradients, pipelined update graphs -- both ranks must end with bit-identical
    parameters (and pass check_synced on the way), which differ from a single-rank run."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    import tempfile
    prefix = os.path.join(tempfile.mkdtemp(), 'digest')
    # (two processes on ONE device: the weights-resident rollout needs all CUs for one launch -- two such launches from two
    #  processes can starve each other of CUs until they time out; one process per GPU, the deployment model, cannot)
    env = dict(os.environ, CURIOUS_DIST_BACKEND='gloo', CURIOUS_RANK_CHECK_CYCLES='2', CURIOUS_RANK_CHECK_OUT=prefix,
               CURIOUS_RESIDENT='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CURIOUS_FORCE_DIST', 'CURIOUS_GRAPH_ALLREDUCE'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(root, 'tools', 'rank_path_check.py')], env=env, cwd=root, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    found = [open('%s.rank%d' % (prefix, r)).read().split() for r in range(2)]
    assert found[0][1] == found[1][1] and len(found[0][1]) == 64, found
    assert found[0][2] == found[1][2] == '235'


def test_batched_rollout_matches_oracle(route):
    """GPU-resident rollout (actor forward + noise + env step kernels) against the oracle env + oracle policy."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    from oracle.env import SyntheticMultiTaskArm
    from oracle.ddpg import action_postprocess
    agent, oracle = build_pair(4, 40)
    B = 6
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                      queue_length=4, eval=False)
    w.seed(77)
    envs = [SyntheticMultiTaskArm(4, 40, T, seed=77, env_id=i) for i in range(B)]
    for cycle in range(3):
        np.random.seed(50 + cycle)
        ep, CP, n_ep = w.generate_rollouts()
        rec = {k: v.cpu().numpy() for k, v in ep.items()}
        # ---- oracle side, same stream
        np.random.seed(50 + cycle)
        exploit = np.random.random() < 0.1
        p = np.ones(4) / 4 if (exploit or cycle == 0) else w_p_before
        tasks = np.random.choice(range(4), p=p, size=B)
        goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
        obs = []
        for i, e in enumerate(envs):
            e.reset()
            obs.append(e.reset_task_goal(goals[i], int(tasks[i])))
        o = np.stack([x['observation'] for x in obs])
        g = np.stack([x['desired_goal'] for x in obs])
        td = np.stack([x['mask'] for x in obs])
        np.testing.assert_array_equal(rec['o'][:, 0], o)
        for t in range(T):
            ne, re = (0., 0.) if exploit else (0.2, 0.3)
            ou = oracle.get_actions(o, o[:, :12], g, task_descr=td, noise_eps=ne, random_eps=re)
            # the product's actions (MFMA forward) differ from the NumPy forward in the last bits; feed the
            # product's recorded action to the oracle env so that env parity is checked exactly step by step
            np.testing.assert_allclose(rec['u'][:, t], ou, rtol=1e-4, atol=2e-5)
            res = [e.step(rec['u'][i, t]) for i, e in enumerate(envs)]
            o = np.stack([r[0]['observation'] for r in res])
            np.testing.assert_array_equal(rec['o'][:, t + 1], o)
            np.testing.assert_array_equal(rec['g'][:, t], g)
            np.testing.assert_array_equal(rec['task_descr'][:, t], td)
            np.testing.assert_array_equal(rec['info_is_success'][:, t, 0],
                                          np.array([r[3]['is_success'] for r in res], np.float32))
        w_p_before = w.p.copy()
        assert n_ep == B * (cycle + 1)
        assert bool(w.exploit) == bool(exploit)
    # the staging block feeds store_episode directly
    agent.store_episode(ep, CP, n_ep)


def test_task_experts_share_buffers_and_match_oracle():
    """structure='task_experts' (train.py:65-121, ddpg.py:302-318,335): one DDPG per task on shared buffers, each
    sampling from buffer t_id+1 and relabelling to its own task."""
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    nb, dimo = 4, 40
    G = 12
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    tr = 'replay_current_task_buffer'
    sampler = make_sample_multi_task_her_transitions('her', 4, tr, sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = make_pooled_buffers(shapes, T * 64, T, sampler, nb + 1, alias_from=5)
    osampler = oher.make_sample_multi_task_her_transitions('her', 4, tr, make_reward_fun(ag_ids, g_ids),
                                                           tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    obufs = [OBuf(shapes, T * 64, T, osampler) for _ in range(nb + 1)]
    gamma = 1. - 1. / T
    experts, oracles = [], []
    for t_id in range(2):
        experts.append(DDPG(input_dims=dims, hidden=64, layers=3,
                            network_class='curious_amd.actor_critic:MultiTaskActorCritic', polyak=0.95, batch_size=128,
                            Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1., clip_obs=200.,
                            scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None, relative_goals=False,
                            clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False,
                            sample_transitions=sampler, gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids,
                            tasks_g_id=g_ids, task_replay=tr, eps_task=0.4, structure='task_experts', t_id=t_id,
                            seed=10 + t_id))
        oracles.append(OracleDDPG(dims, T, obufs, osampler, ag_ids, g_ids, hidden=64, batch_size=128,
                                  task_replay=tr, structure='task_experts', t_id=t_id,
                                  weight_rng=np.random.RandomState(10 + t_id)))
    assert experts[0].buffer[1] is experts[1].buffer[1] and experts[0].scope == 'ddpg0'
    rng = np.random.RandomState(2)
    ep = synth_episodes(rng, 30, nb, dimo)
    np.random.seed(1)
    experts[0].store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(nb), 30)
    np.random.seed(1)
    oracles[0].store_episode({k: v.astype(np.float64) for k, v in ep.items()}, np.zeros(nb), 30)
    for t_id in range(2):
        np.random.seed(40 + t_id)
        got = [x.cpu().numpy() for x in experts[t_id].sample_batch()]
        np.random.seed(40 + t_id)
        want = oracles[t_id].sample_batch()
        np.testing.assert_array_equal(experts[t_id].proportions, oracles[t_id].proportions)
        assert experts[t_id].proportions[t_id + 1] == 128
        for name, a, b in zip(['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r'], got, want):
            np.testing.assert_array_equal(a.astype(np.float64), np.asarray(b, dtype=np.float64), err_msg=name)
        # every HER-relabelled row now belongs to this expert's task
        np.random.seed(50 + t_id)
        ql, _ = oracles[t_id].train()
        np.random.seed(50 + t_id)
        cl, _ = experts[t_id].train()
        assert abs(float(cl) - float(ql)) <= 1e-5 * abs(float(ql))


@pytest.mark.parametrize('task_replay', ['replay_cp_task_transition', 'replay_random_task_transition',
                                         'replay_current_task_transition'])
def test_single_buffer_replay_modes_match_oracle(task_replay):
    """The single-buffer task-replay strategies (train.py:38-45, her.py:138-164, ddpg.py:288-299)."""
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import ReplayBuffer
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    nb, dimo, G = 4, 40, 12
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    sampler = make_sample_multi_task_her_transitions('her', 4, task_replay,
                                                     sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    buf = ReplayBuffer(shapes, T * 48, T, sampler)
    gamma = 1. - 1. / T
    agent = DDPG(input_dims=dims, hidden=64, layers=3, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                 polyak=0.95, batch_size=128, Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1.,
                 action_l2=1., clip_obs=200., scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None,
                 relative_goals=False, clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False,
                 sample_transitions=sampler, gamma=gamma, buffers=buf, tasks_ag_id=ag_ids, tasks_g_id=g_ids,
                 task_replay=task_replay, eps_task=0.4, structure='curious', seed=4)
    osampler = oher.make_sample_multi_task_her_transitions('her', 4, task_replay, make_reward_fun(ag_ids, g_ids),
                                                           tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    oracle = OracleDDPG(dims, T, OBuf(shapes, T * 48, T, osampler), osampler, ag_ids, g_ids, hidden=64,
                        batch_size=128, task_replay=task_replay, weight_rng=np.random.RandomState(4))
    rng = np.random.RandomState(6)
    cp = np.array([0.3, 0.0, 0.1, 0.2])
    for k in range(2):                                                 # 2 x 30 episodes into 48 slots -> overflow
        ep = synth_episodes(rng, 30, nb, dimo)
        np.random.seed(20 + k)
        agent.store_episode({k2: v.copy() for k2, v in ep.items()}, cp, 30)
        np.random.seed(20 + k)
        oracle.store_episode({k2: v.astype(np.float64) for k2, v in ep.items()}, cp, 30)
    assert agent.buffer.current_size == oracle.buffer.current_size == 48
    E = 48
    for key, v in agent.buffer.buffers.items():
        np.testing.assert_array_equal(v[:E].cpu().numpy().astype(np.float64), oracle.buffer.buffers[key][:E])
    np.random.seed(77)
    got = [x.cpu().numpy() for x in agent.sample_batch()]
    s1 = np.random.uniform()
    np.random.seed(77)
    want = oracle.sample_batch()
    s2 = np.random.uniform()
    assert s1 == s2
    for name, a, b in zip(['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r'], got, want):
        np.testing.assert_array_equal(a.astype(np.float64), np.asarray(b, dtype=np.float64), err_msg=name)
    np.random.seed(78)
    ql, _ = oracle.train()
    np.random.seed(78)
    cl, _ = agent.train()
    assert abs(float(cl) - float(ql)) <= 1e-5 * abs(float(ql))


def test_generic_env_path_drives_gpu_agent():
    """Real-env adapter flow: a Python list of host envs (here the NumPy oracle arm) stepped by the
    reference's loop, actions from the GPU policy, NumPy episode dicts stored and trained on."""
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    from oracle.env import SyntheticMultiTaskArm
    agent, oracle = build_pair(4, 40, batch_size=64)
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    counter = [0]

    def make_env():
        e = SyntheticMultiTaskArm(4, 40, T, seed=3, env_id=counter[0])
        counter[0] += 1
        return e
    np.random.seed(5)
    w = RolloutWorker(make_env, agent, dims, logger, T=T, rollout_batch_size=3, noise_eps=0.2, random_eps=0.3,
                      structure='curious', task_selection='active_competence_progress', queue_length=6, eval=False)
    assert not w.batched
    for c in range(3):
        ep, CP, n_ep = w.generate_rollouts()
        assert ep['o'].shape == (3, T + 1, 40) and ep['o'].dtype == np.float32 and ep['change'].dtype == bool
        assert np.abs(ep['u']).max() <= 1.0
        agent.store_episode(ep, CP, n_ep)
    assert agent.buffer[1].current_size > 0                           # the gripper moved: Reach episodes stored
    ev = RolloutWorker(make_env, agent, dims, logger, T=T, rollout_batch_size=2, exploit=True, compute_Q=True,
                       structure='curious', task_selection='active_competence_progress', queue_length=6, eval=True)
    ev.generate_rollouts()
    assert np.isfinite(ev.current_mean_Q())
    first = None
    for k in range(20):
        cl, _ = agent.train()
        first = float(cl) if first is None else first
    agent.update_target_net()
    assert np.isfinite(float(cl)) and float(cl) < first


def test_training_state_checkpoint_resumes_bit_exactly(tmp_path):
    """save -> continue == load into a fresh job -> continue (parameters, Adam state, buffers, queues)."""
    from curious_amd.checkpoint import load_training_state, save_training_state
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)

    def job():
        agent, _ = build_pair(4, 40, rng_mode='device', use_graph=True, batch_size=64, hidden=64)
        w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=8,
                          noise_eps=0.2, random_eps=0.3, structure='curious',
                          task_selection='active_competence_progress', queue_length=6, eval=False)
        w.seed(11)
        return agent, w

    def cycles(agent, w, n):
        for _ in range(n):
            ep, cp, n_ep = w.generate_rollouts()
            agent.store_episode(ep, cp, n_ep)
            for _ in range(7):
                agent.train()
            agent.update_target_net()
    a1, w1 = job()
    np.random.seed(123)
    cycles(a1, w1, 6)                                                 # 48 episodes: the gripper task's buffer (64 slots) ...
    path = str(tmp_path / 'state.pt')
    save_training_state(path, a1, [w1])
    cycles(a1, w1, 4)                                                 # ... overflows after the save: random slots, whose
    a2, w2 = job()                                                    # Philox call index is part of the checkpoint
    np.random.seed(999)                                               # overwritten by the checkpoint
    load_training_state(path, a2, [w2])
    cycles(a2, w2, 4)
    for x in (w1, w2, a1, a2):
        x.settle()
    torch.cuda.synchronize()
    assert a1.buffer[1].current_size == a1.buffer[1].size == 64
    assert torch.equal(a1.theta, a2.theta) and torch.equal(a1.theta_target, a2.theta_target)
    assert torch.equal(a1._m, a2._m) and torch.equal(a1._v, a2._v)
    assert torch.equal(a1.o_stats.state, a2.o_stats.state)
    assert a1.Q_adam.t == a2.Q_adam.t == 70 and w1.n_episodes == w2.n_episodes
    for b1, b2 in zip(a1.buffer[1:], a2.buffer[1:]):
        assert b1.current_size == b2.current_size
        assert torch.equal(b1.records[:b1.current_size], b2.records[:b2.current_size])
    np.testing.assert_array_equal(w1.p, w2.p)


def test_active_goal_selection_feeds_sagg_riac():
    """goal_selection='active' (rollout.py:81-87,121-128,357-365): goals come from the per-task SAGG-RIAC selectors,
    are applied `directly`, and exploit rollouts feed (goal, success) back into the selector of their task."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    agent, _ = build_pair(4, 40, rng_mode='device')
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=64, exploit=True,
                      structure='curious', task_selection='active_competence_progress', goal_selection='active',
                      queue_length=6, eval=False)
    w.seed(5)
    w._decide_exploit = lambda: setattr(w, 'exploit', True)        # rollout.py:184 draws this with p = 0.1; pin it
    np.random.seed(11)
    n_per_task = np.zeros(4, int)
    for _ in range(5):
        ep, _, _ = w.generate_rollouts()
        torch.cuda.synchronize()
        g = w.benv.g.cpu().numpy()
        tasks = w.benv.tasks.cpu().numpy()
        for b in range(64):
            own = g[b, 3 * tasks[b]:3 * tasks[b] + 3]
            assert np.all(np.abs(own) <= 0.5) and np.count_nonzero(g[b]) <= 3       # goal only on the task's slots
            n_per_task[tasks[b]] += 1
    for task in range(4):
        sel = w.goal_selectors[task]
        stored = sum(len(r[0]) for r in sel.regions)
        assert stored == n_per_task[task] > 0 and len(w.split_histories[task]) == 5
        assert abs(sum(sel.probas) - 1.0) < 1e-9


def test_evaluator_rollout_graph_equals_eager():
    """Evaluator rollouts (exploit, compute_Q): the hipGraph replay of the noise-free rollout writes the same episode
    records and reports the same mean Q as the eager launches."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    recs, qs = [], []
    for use_graph in (True, False):
        agent, _ = build_pair(4, 40, rng_mode='device', use_graph=use_graph)
        w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=48,
                          exploit=True, compute_Q=True, structure='curious', task_selection='active_competence_progress',
                          queue_length=6, eval=True)
        w.seed(5)
        np.random.seed(8)
        for _ in range(3):
            ep, _, _ = w.generate_rollouts()
        torch.cuda.synchronize()
        # (round 5: the evaluator's rollout is the fused one-launch rollout, its Q values come from the recorded rows --
        #  DDPG.rollout_q_sum; the launch-per-step evaluator is compared with it in tests/test_gpu_round5.py)
        assert any(k[0] == id(w.benv) for k in getattr(agent, '_roll_graphs', {})) == use_graph
        recs.append(ep.records.clone())
        qs.append(list(w.Q_history))
    assert torch.equal(recs[0], recs[1])
    assert qs[0] == qs[1] and len(qs[0]) == 3 and np.isfinite(qs[0]).all()


def test_fused_act_and_step_equals_unfused(route):
    """curious_policy_act_env_step == get_actions + env.step_all, bit for bit (throughput mode)."""
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    recs = []
    for fused in (True, False):
        agent, _ = build_pair(4, 40, rng_mode='device')
        if not fused:
            agent.can_act_and_step = lambda env, compute_Q: False
        w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=37,
                          noise_eps=0.2, random_eps=0.3, structure='curious',
                          task_selection='active_competence_progress', queue_length=6, eval=False)
        w.seed(5)
        np.random.seed(8)
        for _ in range(2):
            ep, _, _ = w.generate_rollouts()
        torch.cuda.synchronize()
        recs.append(ep.records.clone())
    assert torch.equal(recs[0], recs[1])
    u = recs[0][:, :T, 64:68]
    assert float(u.abs().max()) <= 1.0 and float(u.abs().sum()) > 0
