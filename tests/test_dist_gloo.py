"""The N > 1 path on CPU: world_size-2 gloo process groups exercising curious_amd.dist and every piece of host
logic that sits on top of a collective (SURVEY 2.3 C1-C13 replacements).  GPU kernels are not involved; where a
computing agent is needed the oracle (NumPy) plays that role, with its all-reduce hooks wired to curious_amd.dist."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from curious_amd import dist
    dist.init_from_env(backend='gloo')
    try:
        out[rank] = globals()[fn_name](rank, world)
    finally:
        torch.distributed.destroy_process_group()


def run_world(fn_name, world):
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, fn_name, out), nprocs=world, join=True)
    return dict(out)


def run2(fn_name):
    return run_world(fn_name, 2)


# ------------------------------------------------------------------ collectives
def _collectives(rank, world):
    from curious_amd import dist
    from curious_amd.util import mpi_average
    assert dist.is_distributed() and dist.world_size() == 2 and dist.rank() == rank
    g = torch.full([1000], float(rank + 1))
    dist.allreduce_sum_(g)                                            # C1/C2: SUM, not mean
    theta = torch.arange(10, dtype=torch.float32) * (rank + 1)
    dist.broadcast_(theta, 0)                                         # C3
    rec = np.array([[rank, 1.0, 1.0], [rank, 0.0, 1.0]])
    allrec = dist.allgather_numpy(rec)                                # C9
    # (blocks of unequal length: processes that stand for different numbers of virtual ranks, dist.virtual_layout)
    urec = dist.allgather_numpy(np.full([3 - rank, 2], float(rank)), uneven=True)
    avg = mpi_average([float(rank), float(rank) + 2.0])              # C11
    obj = dist.broadcast_object({'i_policy': 7 + rank}, 0)            # C12
    return dict(g=g.numpy().copy(), theta=theta.numpy().copy(), allrec=allrec, avg=avg, obj=obj, urec=urec)


def test_collectives_world2():
    out = run2('_collectives')
    for r in (0, 1):
        np.testing.assert_array_equal(out[r]['g'], np.full(1000, 3.0, np.float32))
        np.testing.assert_array_equal(out[r]['theta'], np.arange(10, dtype=np.float32))
        np.testing.assert_array_equal(out[r]['allrec'], np.array([[0, 1, 1], [0, 0, 1], [1, 1, 1], [1, 0, 1]], float))
        np.testing.assert_array_equal(out[r]['urec'], np.array([[0, 0]] * 3 + [[1, 1]] * 2, float))
        assert out[r]['avg'] == (0 + 2 + 1 + 3) / 4.0
        assert out[r]['obj'] == {'i_policy': 7}


# ------------------------------------------------------------------ data-parallel DDPG arithmetic
def _make_agent(rank, world, allreduce):
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    from test_gpu_agent import synth_episodes, tables, T
    nb, dimo = 4, 40
    ids, _ = tables(nb)
    G = 12
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    rng = np.random.RandomState(1000 + rank)                          # rank-local data and sampler stream
    sampler = oher.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', make_reward_fun(ids, ids),
                                                          tasks_ag_id=ids, tasks_g_id=ids, rng=rng)
    bufs = [OBuf(shapes, T * 32, T, sampler, rng=rng) for _ in range(nb + 1)]
    agent = OracleDDPG(dims, T, bufs, sampler, ids, ids, hidden=32, batch_size=64, rng=rng,
                       weight_rng=np.random.RandomState(5), allreduce_sum=allreduce, comm_size=world)
    agent.store_episode({k: v.astype(np.float64) for k, v in synth_episodes(rng, 8, nb, dimo).items()},
                        np.zeros(nb), 8)
    return agent


def _dp_train(rank, world):
    """Two ranks, private buffers, gradients SUMMED by curious_amd.dist (ddpg.py:452, mpi_adam.py:26) and normaliser
    sums AVERAGED (normalizer.py:84-94)."""
    from curious_amd import dist

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.allreduce_sum_(t)
        return t.numpy()
    agent = _make_agent(rank, world, allreduce)
    losses = [float(agent.train()[0]) for _ in range(3)]
    return dict(theta=agent.theta.copy(), o_mean=agent.o_stats.mean.copy(), o_count=agent.o_stats.count.copy(),
                losses=losses)


def test_data_parallel_sum_of_gradients_and_mean_of_stats():
    out = run2('_dp_train')
    # replicas stay bit-identical (what check_synced asserts, mpi_adam.py:42-50)
    np.testing.assert_array_equal(out[0]['theta'], out[1]['theta'])
    np.testing.assert_array_equal(out[0]['o_mean'], out[1]['o_mean'])
    # single-process emulation of the same two ranks: identical data / sampler streams, gradients summed by hand
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    a0 = _make_agent(0, 2, None)
    a1 = _make_agent(1, 2, None)
    from oracle.optim import adam_update
    for k in range(3):
        b0, b1 = a0.sample_batch(), a1.sample_batch()
        g0, g1 = a0.grads(b0), a1.grads(b1)
        Qg = g0['Q_grad'] + g1['Q_grad']
        pig = g0['pi_grad'] + g1['pi_grad']
        for a in (a0, a1):
            PQ = a.math.P_Q
            th, m, v, a.t_Q = adam_update(a.theta[:PQ], a.m[:PQ], a.v[:PQ], a.t_Q, Qg, a.Q_lr)
            a.theta[:PQ], a.m[:PQ], a.v[:PQ] = th, m, v
            th, m, v, a.t_pi = adam_update(a.theta[PQ:], a.m[PQ:], a.v[PQ:], a.t_pi, pig, a.pi_lr)
            a.theta[PQ:], a.m[PQ:], a.v[PQ:] = th, m, v
        assert abs(float(g0['Q_loss']) - out[0]['losses'][k]) < 1e-6 * max(1.0, abs(out[0]['losses'][k]))
        assert abs(float(g1['Q_loss']) - out[1]['losses'][k]) < 1e-6 * max(1.0, abs(out[1]['losses'][k]))
    np.testing.assert_allclose(a0.theta, out[0]['theta'], rtol=0, atol=1e-6)
    # count after one store on both ranks: 1 + mean(400, 400) = 401
    assert float(out[0]['o_count'][0]) == 1.0 + 8 * 50


# ------------------------------------------------------------------ competence queues / task probabilities
def _rollout_ranks(rank, world):
    from curious_amd.rollout import RolloutWorker
    from curious_amd import logger
    from oracle.env import SyntheticMultiTaskArm
    from test_host_logic import FakePolicy
    from conftest import load_golden
    G = load_golden('rollout')
    nb, dimo, T, B = [int(x) for x in G['cfg']]
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    counter = [0]

    def make_env():
        e = SyntheticMultiTaskArm(nb, dimo, T, seed=rank, env_id=rank * 100 + counter[0])
        counter[0] += 1
        return e
    np.random.seed(7 + 1000000 * rank)                               # train.py:242
    w = RolloutWorker(make_env, FakePolicy(G['A']), dims, logger, T=T, rollout_batch_size=B, noise_eps=0.2,
                      random_eps=0.3, structure='curious', task_selection='active_competence_progress',
                      queue_length=4, eval=False)
    hist = []
    for c in range(40):
        ep, CP, n_ep = w.generate_rollouts()
        hist.append((bool(w.exploit), np.asarray(w.p).copy(), np.asarray(CP).copy(), np.asarray(w.C).copy(), n_ep,
                     [q.size for q in w.competence_computers]))
    return hist


def test_competence_allgather_makes_ranks_agree():
    out = run2('_rollout_ranks')
    h0, h1 = out[0], out[1]
    saw_asym = False
    for (e0, p0, cp0, c0, n0, s0), (e1, p1, cp1, c1, n1, s1) in zip(h0, h1):
        # exploit flags are rank-local draws (rollout.py:184) yet queues, C, CP agree on every rank (C9 -> C10)
        np.testing.assert_array_equal(cp0, cp1)
        np.testing.assert_array_equal(c0, c1)
        assert s0 == s1 and n0 == n1
        if not e0 and not e1:
            np.testing.assert_array_equal(p0, p1)
        saw_asym |= (e0 != e1)
    assert h0[-1][4] == 40 * 3 * 2                                    # n_episodes counts both ranks (rollout.py:313)
    assert saw_asym
    assert any(cp.sum() > 0 for _, _, cp, _, _, _ in h0)


def _distinct_seed_check(rank, world):
    """train.py:207-212: ranks must hold different NumPy streams."""
    from curious_amd import dist
    np.random.seed(5 + 1000000 * rank)
    local = float(np.random.uniform())
    root = dist.broadcast_object(local, 0)
    return (local, root)


def test_rank_seeds_differ():
    out = run2('_distinct_seed_check')
    assert out[1][0] != out[1][1] and out[0][0] == out[0][1]


# ------------------------------------------------------------------ world sizes 4 and 8 (the 1/2/4/8-GPU job shapes)
def _dp_exchanges(rank, world):
    """The three exchange steps of the data-parallel path at a given world size: C1+C2 gradient SUM on the fused
    vector (mpi_adam.py:26), C5 normaliser accumulators averaged over ranks (normalizer.py:84-94) through the oracle
    normaliser wired to curious_amd.dist, C9 competence all-gather feeding identical queues on every rank
    (rollout.py:332-356 -> RolloutWorker._finish_rollout)."""
    from curious_amd import dist, logger
    from curious_amd.rollout import RolloutWorker
    from oracle.normalizer import Normalizer
    from oracle.env import SyntheticMultiTaskArm

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.allreduce_sum_(t)
        return t.numpy()
    rng = np.random.RandomState(100 + rank)
    g = torch.from_numpy(rng.randn(294_661 // 64).astype(np.float32))
    g_local = g.clone()
    dist.allreduce_sum_(g)
    nz = Normalizer(5, 0.01, 5.0, allreduce, world)
    rows = rng.randn(10 + rank, 5)                                   # ranks contribute different row counts
    nz.update(rows)
    nz.recompute_stats()
    nb, dimo, T, B = 4, 40, 5, 3

    class Pol:
        def get_actions(self, o, ag, g, **kw):
            return np.zeros([o.shape[0], 4], np.float32)
    counter = [0]

    def make_env():
        e = SyntheticMultiTaskArm(nb, dimo, T, seed=rank, env_id=rank * 100 + counter[0])
        counter[0] += 1
        return e
    np.random.seed(9 + 1000000 * rank)
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    w = RolloutWorker(make_env, Pol(), dims, logger, T=T, rollout_batch_size=B, structure='curious',
                      task_selection='active_competence_progress', queue_length=4, eval=False)
    for _ in range(25):
        w.generate_rollouts()
    return dict(g=g.numpy().copy(), g_local=g_local.numpy().copy(), rows=rows, mean=nz.mean.copy(), std=nz.std.copy(),
                count=float(nz.count[0]), CP=np.asarray(w.CP).copy(), C=np.asarray(w.C).copy(), n_ep=w.n_episodes,
                qsizes=[q.size for q in w.competence_computers])


@pytest.mark.parametrize('world', [4, 8])
def test_exchange_steps_at_world_4_and_8(world):
    out = run_world('_dp_exchanges', world)
    total = np.sum([out[r]['g_local'].astype(np.float64) for r in range(world)], axis=0)
    for r in range(world):
        np.testing.assert_array_equal(out[r]['g'], out[0]['g'])                  # replicas agree bit for bit
        np.testing.assert_allclose(out[r]['g'], total, rtol=1e-6, atol=1e-6)     # and hold the SUM, not the mean
        np.testing.assert_array_equal(out[r]['mean'], out[0]['mean'])
        np.testing.assert_array_equal(out[r]['CP'], out[0]['CP'])
        np.testing.assert_array_equal(out[r]['C'], out[0]['C'])
        assert out[r]['qsizes'] == out[0]['qsizes'] and out[r]['n_ep'] == 25 * 3 * world
    # normaliser: count = 1 + mean over ranks of the local counts, sums likewise (normalizer.py:37-39,84-94)
    counts = [out[r]['rows'].shape[0] for r in range(world)]
    assert out[0]['count'] == pytest.approx(1.0 + np.mean(counts), rel=1e-6)
    s = np.mean([out[r]['rows'].astype(np.float32).sum(axis=0) for r in range(world)], axis=0)
    np.testing.assert_allclose(out[0]['mean'], s / out[0]['count'], rtol=1e-5, atol=1e-6)
    assert sum(out[0]['qsizes']) > 0


# ------------------------------------------------------------------ data-parallel batched experts (configs[4] at N > 1)
def _expert_block_sum(rank, world):
    """N task experts per rank; the reference all-reduces every expert's Q and pi gradient on its own (2 N Allreduces per
    round of updates: train.py:65-121, mpi_adam.py:21-35).  ExpertBank keeps the N gradient vectors in one [N, P] block
    (padded like the device layout) and sums it over the ranks with ONE collective; both ways must give every expert
    the same parameters, bit for bit."""
    from curious_amd import dist
    from oracle.ddpg import OracleDDPG, STAGE_KEYS
    nb, dimo, B, Tn, hidden = 3, 10, 16, 5, 24
    G = 3 * nb
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb)]
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.allreduce_sum_(t)
        return t.numpy()

    def experts(hook):
        return [OracleDDPG(dims, Tn, [None] * (nb + 1), None, ids, ids, hidden=hidden, layers=2, batch_size=B,
                           structure='task_experts', t_id=e, task_replay='replay_current_task_buffer',
                           weight_rng=np.random.RandomState(7 + e), allreduce_sum=hook) for e in range(nb)]
    ref, blk = experts(allreduce), experts(None)
    P_Q, P = ref[0].math.P_Q, ref[0].theta.shape[0]
    off_pi = (P_Q + 63) & ~63
    stride = off_pi + ((P - P_Q + 63) & ~63)                        # the padded device layout [Q | pad | pi | pad]
    rng = np.random.RandomState(50 + rank)                            # rank-private data
    n_collectives = [0]
    for step in range(3):
        block = np.zeros([nb, stride], np.float32)
        outs = []
        for e in range(nb):
            shapes = dict(ag=G, g=G, o=dimo, task_descr=nb, u=4, o_2=dimo, g_2=G, r=1)
            batch = [rng.randn(B, shapes[k]).astype(np.float32) for k in STAGE_KEYS]
            ref[e].train([b.copy() for b in batch])                  # 2 all-reduces inside
            out = blk[e].grads(batch)
            block[e, :P_Q] = out['Q_grad']
            block[e, off_pi:off_pi + P - P_Q] = out['pi_grad']
            outs.append(out)
        block = allreduce(block)                                     # ONE all-reduce of N x P
        n_collectives[0] += 1
        for e in range(nb):
            blk[e]._allreduce_sum = None
            g = dict(Q_grad=block[e, :P_Q], pi_grad=block[e, off_pi:off_pi + P - P_Q])
            orig = blk[e].grads
            blk[e].grads = lambda b, g=g, o=outs[e]: dict(o, **g)    # feed train() the summed gradients
            blk[e].train([None] * len(STAGE_KEYS))
            blk[e].grads = orig
    return dict(ref=[x.theta.copy() for x in ref], blk=[x.theta.copy() for x in blk], pads=float(np.abs(
        block[:, P_Q:off_pi]).sum()), n=n_collectives[0])


def test_expert_gradient_block_is_summed_by_one_collective():
    out = run2('_expert_block_sum')
    for e in range(3):
        np.testing.assert_array_equal(out[0]['ref'][e], out[0]['blk'][e])        # one N x P sum == 2 N separate sums
        np.testing.assert_array_equal(out[0]['blk'][e], out[1]['blk'][e])        # and the replicas agree
    assert not np.array_equal(out[0]['blk'][0], out[0]['blk'][1])                # experts stay distinct
    assert out[0]['pads'] == 0.0 and out[0]['n'] == 3


# ------------------------------------------------------------------ bench.py's multi-process CPU baseline leg
def test_cpu_baseline_ranks_leg_of_bench():
    """bench.py cpu_baseline_ranks (SURVEY 8d ii): R single-threaded oracle processes, gradients summed over gloo."""
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline_ranks(2, budget_s=2.0)
    assert 'error' not in out, out
    assert out['cores'] == 2 and out['kind'] == 'port' and out['value'] > 0 and out['env_steps_per_sec'] > 0
    assert 'gloo' in out['sample']


def test_captured_allreduce_is_off_without_rccl(monkeypatch):
    """dist.captured_allreduce_ok: forced by the environment, otherwise False unless an RCCL process group exists."""
    from curious_amd import dist
    monkeypatch.setenv('CURIOUS_GRAPH_ALLREDUCE', '1')
    assert dist.captured_allreduce_ok() is True
    monkeypatch.setenv('CURIOUS_GRAPH_ALLREDUCE', '0')
    assert dist.captured_allreduce_ok() is False
    monkeypatch.delenv('CURIOUS_GRAPH_ALLREDUCE')
    monkeypatch.setattr(dist, '_CAPTURED_OK', None)
    assert dist.captured_allreduce_ok() is False                   # no process group in this process


def test_virtual_layout_covers_exactly_the_ranks_asked_for():
    """dist.virtual_layout (--num_cpu R on W processes): the processes' rank counts add up to R, their blocks of global
    ranks are contiguous and in process order, no process stands for more than one rank above another; R <= W: one rank
    per process (the reference's own layout)."""
    from curious_amd import dist
    for R in (2, 3, 5, 8, 19, 24, 64):
        for W in (1, 2, 3, 4, 8):
            lay = [dist.virtual_layout(R, W, r) for r in range(W)]
            if R <= W:
                assert lay == [(1, r, W) for r in range(W)]
                continue
            assert sum(v for v, _, _ in lay) == R and all(t == R for _, _, t in lay)
            assert [b for _, b, _ in lay] == [sum(v for v, _, _ in lay[:r]) for r in range(W)]
            assert max(v for v, _, _ in lay) - min(v for v, _, _ in lay) <= 1
            assert [v for v, _, _ in lay] == sorted((v for v, _, _ in lay), reverse=True)
    assert [dist.virtual_layout(19, 8, r)[0] for r in range(8)] == [3, 3, 3, 2, 2, 2, 2, 2]
