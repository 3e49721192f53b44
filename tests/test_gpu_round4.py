"""Round 4: the several-rank arithmetic of the HIP path against the oracle's two-rank model, the scaling harness run end
to end on two ranks, and the fault paths that keep a job alive."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_KEYS = ['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r']


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _two_rank_env(**extra):
    """Two processes on ONE device: gloo carries the collectives (RCCL refuses two ranks per device), and the
    weights-resident rollout -- which needs every CU for one launch -- is switched off."""
    env = dict(os.environ, CURIOUS_DIST_BACKEND='gloo', CURIOUS_RESIDENT='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CURIOUS_FORCE_DIST', 'CURIOUS_GRAPH_ALLREDUCE'):
        env.pop(k, None)
    env.update(extra)
    return env


def _launch2(script_args, env, timeout=900):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


# ------------------------------------------------------------------ world-2 parity against the oracle
def _oracle_agent(seed):
    from oracle.ddpg import OracleDDPG
    from test_gpu_agent import T, tables
    nb, dimo = 4, 40
    ids, _ = tables(nb)
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    return OracleDDPG(dims, T, [], None, ids, ids, batch_size=256, weight_rng=np.random.RandomState(seed))


def _recompute_pair(nzs):
    """normalizer.py:84-94 for two ranks in one process: every rank's three Allreduce(SUM) calls return the sum of both
    ranks' local accumulators, then / comm size."""
    tot = [nzs[0].local_sum + nzs[1].local_sum, nzs[0].local_sumsq + nzs[1].local_sumsq,
           nzs[0].local_count + nzs[1].local_count]
    for nz in nzs:
        q = [t.copy() for t in tot]
        nz._allreduce = lambda x, q=q: q.pop(0)
        nz._comm_size = 2
        nz.recompute_stats()


@pytest.mark.parametrize('mode,graph', [('single', 0), ('single', 1), ('experts', 0), ('experts', 1)])
def test_two_rank_update_and_normaliser_match_the_oracle_two_rank_model(mode, graph):
    """mpi_adam.py:21-50, ddpg.py:452-453 (gradients SUMMED over ranks), normalizer.py:84-94 (sums AVERAGED over ranks).
    Two ranks of the product on one GPU (gloo), private data per rank; the batches the device drew are replayed through
    two oracle agents whose gradients are summed by hand.  Losses within 1e-5 relative on every rank and update, theta /
    m / v / target together after 8 updates, normaliser state = 1 + mean of the rank counts etc.; for the batched
    experts the same per expert (ONE all-reduce of the [4, P] block on the product side)."""
    from oracle.optim import adam_update, polyak_update
    prefix = os.path.join(tempfile.mkdtemp(), 'w2')
    out = _launch2([os.path.join(ROOT, 'tests', 'rank_parity_worker.py'), prefix, mode, str(graph)], _two_rank_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rec = [np.load('%s.rank%d.npz' % (prefix, r)) for r in range(2)]
    n_exp = 4 if mode == 'experts' else 1
    n_upd = int(rec[0]['n_updates'])
    assert n_upd == 8
    # ---- replicas agree bit for bit (what check_synced asserts) and the ranks really saw different data
    for e in range(n_exp):
        for name in ('theta', 'm', 'v', 'target'):
            np.testing.assert_array_equal(rec[0]['%s_%d' % (name, e)], rec[1]['%s_%d' % (name, e)])
    assert not np.array_equal(rec[0]['batch_0_0_o'], rec[1]['batch_0_0_o'])
    # ---- normalisers: both stores
    from oracle.normalizer import Normalizer as ONorm
    o_nz = [ONorm(40, 0.01, 5) for _ in range(2)]
    g_nz = [ONorm(12, 0.01, 5) for _ in range(2)]
    for tag in ('a', 'b'):
        for r in range(2):
            o_nz[r].update(np.clip(rec[r]['stats_o_' + tag].astype(np.float64), -200, 200))
            g_nz[r].update(np.clip(rec[r]['stats_g_' + tag].astype(np.float64), -200, 200))
        _recompute_pair(o_nz)
        _recompute_pair(g_nz)
        for r in range(2):
            for nz, key in ((o_nz[r], 'o_state_' + tag), (g_nz[r], 'g_state_' + tag)):
                d = nz.size
                st = rec[r][key]
                np.testing.assert_allclose(st[:d], nz.sum, rtol=1e-5, atol=1e-4, err_msg=key)
                np.testing.assert_allclose(st[d:2 * d], nz.sumsq, rtol=1e-5, atol=1e-4, err_msg=key)
                assert float(st[2 * d]) == float(nz.count[0]), key  # 1 + mean over ranks of the rows fed
                np.testing.assert_allclose(st[2 * d + 1:3 * d + 1], nz.mean, rtol=1e-5, atol=1e-6, err_msg=key)
                np.testing.assert_allclose(st[3 * d + 1:], nz.std, rtol=1e-5, atol=1e-6, err_msg=key)
    assert float(o_nz[0].count[0]) == 1.0 + 2 * 24 * 50            # two stores of 24 episodes x T on each rank
    # ---- updates, step by step from the product's own state (no drift between two trajectories): per expert two oracle
    # ranks evaluate their batch at theta_k, the gradients are SUMMED (not averaged), one Adam step follows
    for e in range(n_exp):
        a = _oracle_agent(3 + e)
        np.testing.assert_array_equal(rec[0]['theta_pre_0_%d' % e], a.theta)      # same initial weights as the oracle's
        target0 = a.theta.copy()                                     # ddpg.py:459-460; constant until update_target_net
        for k in range(n_upd):
            th, m, v = (rec[0]['%s_pre_%d_%d' % (name, k, e)] for name in ('theta', 'm', 'v'))
            np.testing.assert_array_equal(th, rec[1]['theta_pre_%d_%d' % (k, e)])
            outs = []
            for r in range(2):
                batch = dict(zip(STAGE_KEYS, [rec[r]['batch_%d_%d_%s' % (k, e, key)] for key in STAGE_KEYS]))
                outs.append(a.math.losses_and_grads(th, target0, batch))
                want, got = float(outs[r]['Q_loss']), float(rec[r]['loss_%d_%d' % (k, e)])
                assert abs(got - want) <= 1e-5 * abs(want), (e, k, r, got, want)
                np.testing.assert_allclose(rec[r]['qpi_%d_%d' % (k, e)], outs[r]['Q_pi'], rtol=1e-4, atol=2e-5)
            PQ = a.math.P_Q
            nxt = [np.empty_like(th), np.empty_like(m), np.empty_like(v)]
            for sl, key, lr in ((slice(0, PQ), 'Q_grad', a.Q_lr), (slice(PQ, None), 'pi_grad', a.pi_lr)):
                g = outs[0][key] + outs[1][key]                      # mpi_adam.py:26: SUM over ranks
                nxt[0][sl], nxt[1][sl], nxt[2][sl], _ = adam_update(th[sl], m[sl], v[sl], k, g, lr)
            suffix = ('_pre_%d_%d' % (k + 1, e)) if k + 1 < n_upd else ('_%d' % e)
            got = [rec[0][name + suffix] for name in ('theta', 'm', 'v')]
            np.testing.assert_allclose(got[1], nxt[1], rtol=0, atol=2e-5 * np.abs(nxt[1]).max())
            np.testing.assert_allclose(got[2], nxt[2], rtol=0, atol=4e-5 * np.abs(nxt[2]).max())
            assert np.abs(got[0] - nxt[0]).max() <= 2e-5, (e, k)     # one Adam step of size 1e-3
            assert (np.abs(got[0] - nxt[0]) > 2e-6).mean() < 1e-3, (e, k)
            if k == 0:
                # SUM, not mean, and not one rank alone: either would leave first moments half as large
                half = adam_update(th[:PQ], m[:PQ], v[:PQ], 0, outs[0]['Q_grad'], a.Q_lr)[1]
                assert np.abs(got[1][:PQ] - half).max() > 0.2 * np.abs(half).max()
        np.testing.assert_allclose(rec[0]['target_%d' % e], polyak_update(target0, rec[0]['theta_%d' % e], 0.95),
                                   rtol=0, atol=1e-6)


# ------------------------------------------------------------------ the scaling harness itself
def test_bench_runs_end_to_end_on_two_ranks():
    """`bench.py --gpus 2` exactly as the driver's scaling run launches it (torch.distributed.run, one rank per process),
    with gloo in place of RCCL because both ranks share this box's one GPU: the line must carry n_gpus = 2, the
    several-rank dominant kernel, the all-reduce timed alone, the communicator's settings, identical replicas -- and
    the processes must shut down cleanly."""
    out = _launch2([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--prefill', '256'],
                   _two_rank_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-1500:]   # stdout = rank 0's ONE line, nothing
    #                                                                  else (gloo's own announcements go to stderr)
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['warmup'] == 1 and rec['scaling'] == 'weak'
    assert rec['config']['parallelism'] == 'dp2'
    assert rec['value'] > 0 and abs(rec['value'] - 2 * 3 * 100 * 256 / (rec['ms_per_step'] * 3e-3)) < 1e-3 * rec['value']
    assert rec['roofline']['kernel'] == 'ddpg_rows_her_kernel', rec['roofline']
    assert rec['roofline']['traffic_source']
    c = rec['collectives']
    assert c['rccl']['backend'] == 'gloo' and c['rccl']['world'] == 2 and c['rccl']['captured'] is False
    assert c['allreduce_us'] > 0 and c['allreduce_bytes'] >= 4 * 294661
    assert c['replicas_identical'] is True and len(c['replica_checksums']) == 2
    assert c['replica_checksums'][0] == c['replica_checksums'][1]
    # the side measurement of the fused IPC all-reduce + Adam path, run in child processes: ran, nobody gave up
    assert 'error' not in c['ipc_probe'], c['ipc_probe']
    assert c['ipc_probe']['world'] == 2 and c['ipc_probe']['ms_per_step'] > 0 and c['ipc_probe']['wait_gave_up'] == 0


# ------------------------------------------------------------------ faults: the job goes on
def test_a_handoff_fault_on_one_rank_freezes_and_resumes_every_rank_alike():
    """the guard of the Q' hand-off used to be rank-local -- the faulted rank skipped its optimiser,
    the healthy ranks applied the NaN-summed gradient.  Now the fault word rides through the gradient all-reduce as a
    padding element of the gradient vector (curious_transposed_t.fault_flag): rank 1 loses a producer in one update, BOTH
    ranks skip that update and every later one, both raise HandoffFault at the same cycle count, both resume, and the
    replicas are bit-identical throughout (mpi_adam.py:42-50)."""
    from curious_amd.ddpg import FAULT_CHECK_EVERY
    prefix = os.path.join(tempfile.mkdtemp(), 'fault')
    out = _launch2([os.path.join(ROOT, 'tests', 'rank_parity_worker.py'), prefix, 'fault', '0'], _two_rank_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rec = [np.load('%s.rank%d.npz' % (prefix, r)) for r in range(2)]
    for tag in ('good', 'hit', 'frozen', 'cleared', 'resumed'):
        np.testing.assert_array_equal(rec[0]['theta_' + tag], rec[1]['theta_' + tag], err_msg=tag)
        np.testing.assert_array_equal(rec[0]['m_' + tag], rec[1]['m_' + tag], err_msg=tag)
    for r in range(2):
        assert np.isfinite(rec[r]['theta_resumed']).all()
        for tag in ('hit', 'frozen', 'cleared'):                     # nothing moved from the faulted update on
            np.testing.assert_array_equal(rec[r]['theta_good'], rec[r]['theta_' + tag], err_msg=tag)
            np.testing.assert_array_equal(rec[r]['m_good'], rec[r]['m_' + tag], err_msg=tag)
        assert int(rec[r]['fault_good']) == 0
        assert int(rec[r]['fault_frozen']) > 0                       # the healthy rank raised its own word as well
        assert int(rec[r]['fault_cleared']) == 0
        assert list(rec[r]['raised']) == [FAULT_CHECK_EVERY]         # every rank in the same cycle, once
        assert not np.array_equal(rec[r]['theta_good'], rec[r]['theta_resumed'])
    assert int(rec[1]['fault_hit']) == 4 and int(rec[0]['fault_hit']) == 1     # 4 waves gave up / raised by the optimiser


def _job(seed=11, B=16, use_graph=True, async_store=False):
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from test_gpu_agent import T, build_pair, synth_episodes
    nb, dimo = 4, 40
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    agent, _ = build_pair(nb, dimo, cap_eps=200, rng_mode='device', use_graph=use_graph, seed=seed)
    agent.async_store = async_store
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='random', queue_length=6,
                      eval=False)
    w.seed(seed)
    w._decide_exploit = lambda: None
    agent.store_episode(synth_episodes(np.random.RandomState(21), 24, nb, dimo), w.CP, 24)
    return agent, w


def _cycles(agent, w, n, n_batches=4):
    for _ in range(n):
        ep, cp, n_ep = w.generate_rollouts()
        agent.store_episode(ep, cp, n_ep)
        agent.train_batches(n_batches)
        agent.update_target_net()
    w.settle(); agent.settle()
    torch.cuda.synchronize()


@pytest.mark.parametrize('use_graph', [False, True])
def test_a_void_resident_rollout_is_redone_on_the_streaming_kernel_with_the_same_numbers(use_graph):
    """a time-out of the weights-resident rollout used to end the job ("rerun with CURIOUS_RESIDENT=0").
    Now the worker switches the process to the streaming kernel, warns once and generates the SAME rollout again (same
    tasks / goals / episode numbers / noise counters): the run ends bit-identical to one that never saw the fault."""
    from curious_amd import ops
    assert ops.get_option('resident') == 1
    np.random.seed(5)
    ref_agent, ref_w = _job(use_graph=use_graph)
    _cycles(ref_agent, ref_w, 3)
    try:
        np.random.seed(5)
        agent, w = _job(use_graph=use_graph)
        _cycles(agent, w, 1)
        agent.drop_rollout_graphs()                                  # (the injection is a launch argument: capture anew)
        with ops.option('fault_inject', 1), ops.option('res_spins', 20000):
            ep, cp, n_ep = w.generate_rollouts()                     # void -> switched -> redone, inside this call
        assert ops.get_option('resident') == 0                      # the process stays on the streaming kernel
        agent.store_episode(ep, cp, n_ep)
        agent.train_batches(4)
        agent.update_target_net()
        _cycles(agent, w, 1)
    finally:
        ops.set_option('resident', 1)
    for name in ('theta', '_m', '_v', 'theta_target'):
        assert torch.equal(getattr(agent, name), getattr(ref_agent, name)), name
    assert [b.current_size for b in agent.buffer] == [b.current_size for b in ref_agent.buffer]
    for b, rb in zip(agent.buffer[1:], ref_agent.buffer[1:]):         # (the storage beyond the stored episodes is never
        assert torch.equal(b.records[:b.current_size], rb.records[:rb.current_size])   # initialised)
    assert torch.equal(w.benv.staging, ref_w.benv.staging)
    assert w.n_episodes == ref_w.n_episodes and agent._noise_counter == ref_agent._noise_counter


def test_a_void_resident_rollout_on_the_async_path_is_replaced_a_cycle_late():
    """async_store: the void launch is seen when its flags are settled, a cycle late -- the device dropped its episodes,
    the worker switches to the streaming kernel and generates + stores a replacement at once; the job goes on."""
    from curious_amd import ops
    np.random.seed(5)
    agent, w = _job(async_store=True)
    try:
        _cycles(agent, w, 2)
        sizes = sum(b.current_size for b in agent.buffer[1:5])
        n_ep0 = w.n_episodes
        agent.drop_rollout_graphs()
        with ops.option('fault_inject', 1), ops.option('res_spins', 20000):
            ep, cp, n_ep = w.generate_rollouts()
            assert getattr(w, '_pending', None) is not None           # returned without waiting
            agent.store_episode(ep, cp, n_ep)
            torch.cuda.synchronize()
        assert float(w.benv.flags[w.benv.n]) == 2.0
        agent.train_batches(4)
        agent.update_target_net()
        agent.settle()
        assert sum(b.current_size for b in agent.buffer[1:5]) == sizes       # dropped on the device
        _cycles(agent, w, 1)                                         # settles the void rollout: replacement + this cycle's
        assert ops.get_option('resident') == 0
        assert sum(b.current_size for b in agent.buffer[1:5]) > sizes
        assert w.n_episodes == n_ep0 + 2 * 16                        # the lost rollout does not count, its replacement does
        assert torch.isfinite(agent.theta).all()
    finally:
        ops.set_option('resident', 1)


@pytest.mark.parametrize('use_graph', [False, True])
def test_guarded_updates_replay_a_faulted_run_bit_identically(use_graph):
    """fault_check='sync' (DDPG.train_batches_guarded): a run of updates in which a consumer of Q' gave up is replayed
    once from the parameters, moments and step counter it started with -- same batches, same step sizes: the job ends
    bit-identical to one that never faulted; a fault in the replay is raised."""
    from curious_amd import ops
    from curious_amd.ddpg import HandoffFault
    from test_gpu_round3 import _filled_agent
    ref, _ = _filled_agent(use_graph=use_graph)
    ref.train_batches(3)
    ref.train_batches(12)
    ref.train_batches(4)
    agent, _ = _filled_agent(use_graph=use_graph)
    agent.train_batches(3)
    plain = agent.train_batches
    calls = []

    def faulty_first(n):
        if not calls:
            saved = (agent._graphs, agent._chains)
            agent._graphs, agent._chains = [None, None], None          # captured with the injection baked in: dropped below
            with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
                out = plain(n)
                torch.cuda.synchronize()
            agent._graphs, agent._chains = saved
        else:
            out = plain(n)
        calls.append(n)
        return out
    agent.train_batches = faulty_first
    with pytest.warns(UserWarning, match='replaying'):
        agent.train_batches_guarded(12)
    assert calls == [12, 12]
    agent.train_batches = plain
    agent.train_batches_guarded(4)                                   # a clean run: no replay
    torch.cuda.synchronize()
    for name in ('theta', '_m', '_v', '_step_ctr'):
        assert torch.equal(getattr(agent, name), getattr(ref, name)), name
    assert agent.Q_adam.t == ref.Q_adam.t == 19
    # a fault that repeats in the replay is raised
    with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
        agent._graphs, agent._chains = [None, None], None
        with pytest.warns(UserWarning), pytest.raises(HandoffFault):
            agent.train_batches_guarded(2)
    agent._graphs, agent._chains = [None, None], None
    agent.check_faults()


def test_the_training_loop_survives_an_isolated_handoff_fault():
    """experiment.train.FaultTolerance (the default, asynchronous form): the verdict arrives cycles later, the loop logs
    it and goes on from the last good parameters; a second fault within the window is raised."""
    from curious_amd import ops
    from curious_amd.ddpg import FAULT_CHECK_EVERY, HandoffFault
    from curious_amd.experiment.train import FaultTolerance
    np.random.seed(5)
    agent, w = _job(use_graph=False)
    ft = FaultTolerance(window=4 * FAULT_CHECK_EVERY)

    def cycle(inject=False):
        ft.tick()
        ep, cp, n_ep = w.generate_rollouts()
        ft.call(agent.store_episode, ep, cp, n_ep)
        if inject:
            with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
                ft.call(agent.train_batches, 3)
                torch.cuda.synchronize()
        else:
            ft.call(agent.train_batches, 3)
        ft.call(agent.update_target_net)
    for _ in range(2):
        cycle()
    torch.cuda.synchronize()
    good = agent.theta.clone()
    cycle(inject=True)
    for _ in range(3 * FAULT_CHECK_EVERY):
        cycle()
    torch.cuda.synchronize()
    assert ft.count == 1                                             # seen once, survived
    assert torch.isfinite(agent.theta).all() and not torch.equal(good, agent.theta)
    agent.check_faults()
    cycle(inject=True)                                               # again, inside the window: raised
    with pytest.raises(HandoffFault):
        for _ in range(3 * FAULT_CHECK_EVERY):
            cycle()


# ------------------------------------------------------------------ relative goals in the fused acting kernels
@pytest.mark.parametrize('normalize_obs', [False, True])
def test_relative_goals_in_the_fused_rollout(normalize_obs):
    """ddpg.py:118-127 with relative_goals: the policy sees clip(g - ag), and ag changes with every env step.  Round 3
    acted through one plain forward + noise + env-step launch per step in this mode; now curious_policy_rollout_stats /
    curious_policy_act_env_step_stats carry the flag: the goal part of the policy's input row is recomputed from the new
    achieved goal inside the env step.  (a) the one-launch rollout == one fused launch per step == the unfused sequence,
    bit for bit, with exploration noise; (b) without noise the recorded actions are the oracle policy's on the recorded
    observations (1e-4) and the oracle env reproduces the recorded episode exactly."""
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from oracle.env import SyntheticMultiTaskArm
    from test_gpu_agent import T, build_pair
    nb, dimo, B = 4, 40, 8
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)

    def job(noise, seed=9):
        agent, oracle = build_pair(nb, dimo, rng_mode='device', use_graph=False, relative_goals=True,
                                   normalize_obs=normalize_obs, seed=seed)
        if normalize_obs:                                            # non-trivial statistics
            rng = np.random.RandomState(4)
            for nz in (agent.o_stats, agent.g_stats):
                nz.update(torch.as_tensor(rng.randn(300, nz.size).astype(np.float32) * 0.5 + 0.1, device='cuda'))
                nz.recompute_stats()
        w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                          noise_eps=0.2 if noise else 0.0, random_eps=0.3 if noise else 0.0, structure='curious',
                          task_selection='random', queue_length=6, eval=False)
        w.seed(13)
        w._decide_exploit = lambda: None
        return agent, oracle, w

    # ---- (a) three ways of stepping, with noise
    recs = []
    for mode in ('rollout', 'per_step', 'unfused'):
        agent, _, w = job(True)
        assert agent.relative_goals and agent.can_act_and_step(w.benv, False)
        np.random.seed(31)
        if mode == 'rollout':
            ep, _, _ = w.generate_rollouts()
        else:
            env = w.benv
            tasks = np.random.choice(range(nb), p=w.p, size=B)
            goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
            env.reset_all(tasks, goals)
            for t in range(T):
                if mode == 'per_step':
                    agent.act_and_step(env, t, noise_eps=0.2, random_eps=0.3)
                else:
                    u = agent.get_actions(env.o, env.ag, env.g, task_descr=env.td, noise_eps=0.2, random_eps=0.3)
                    env.step_all(u, t)
            ep = env.episode_views()
        torch.cuda.synchronize()
        recs.append({k: v.clone() for k, v in ep.items()})
    for k in recs[0]:
        assert torch.equal(recs[0][k], recs[1][k]), ('rollout vs per-step', k)
        assert torch.equal(recs[0][k], recs[2][k]), ('rollout vs unfused', k)
    assert float(recs[0]['u'].abs().sum()) > 0

    # ---- (b) no noise: against the oracle policy (relative goals) and the oracle env
    if normalize_obs:
        return              # (the oracle AGENT has no input normalisation -- its networks do: tests/test_gpu_kernels.py)
    agent, oracle, w = job(False)
    np.random.seed(31)
    ep, _, _ = w.generate_rollouts()
    rec = {k: v.cpu().numpy() for k, v in ep.items()}
    envs = [SyntheticMultiTaskArm(nb, dimo, T, seed=13, env_id=i) for i in range(B)]
    np.random.seed(31)
    tasks = np.random.choice(range(nb), p=np.ones(nb) / nb, size=B)
    goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
    obs = []
    for i, e in enumerate(envs):
        e.reset()
        obs.append(e.reset_task_goal(goals[i], int(tasks[i])))
    o = np.stack([x['observation'] for x in obs])
    g = np.stack([x['desired_goal'] for x in obs])
    td = np.stack([x['mask'] for x in obs])
    np.testing.assert_array_equal(rec['o'][:, 0], o)
    moved = 0.0
    for t in range(T):
        ou = oracle.get_actions(o, o[:, :12], g, task_descr=td, noise_eps=0., random_eps=0.)
        np.testing.assert_allclose(rec['u'][:, t], ou, rtol=1e-4, atol=2e-5, err_msg='step %d' % t)
        res = [e.step(rec['u'][i, t]) for i, e in enumerate(envs)]
        o_new = np.stack([r[0]['observation'] for r in res])
        np.testing.assert_array_equal(rec['o'][:, t + 1], o_new)
        moved += np.abs(o_new[:, :12] - o[:, :12]).sum()
        o = o_new
    assert moved > 0                                                 # the achieved goals did change: g - ag was re-derived


# ------------------------------------------------------------------ batched experts with input normalisation
@pytest.mark.parametrize('use_graph', [False, True])
def test_batched_experts_with_input_normalisation_equal_sequential_experts(use_graph):
    """actor_critic.py:76-83 + train.py:285-291: every task expert is a DDPG of its own -- with --normalize_obs each has
    its own two normalisers, fed only by the rollouts IT collected.  Round 3's ExpertBank refused normalize_obs; now the
    experts' normaliser state lives in their slab rows and the batched launches read expert e's statistics at e x stride.
    Bit-identical to the same experts updated one after the other (whose kernel is checked against the oracle with
    normalisation in tests/test_gpu_kernels.py)."""
    from curious_amd.experts import ExpertBank
    from test_gpu_agent import synth_episodes
    from test_gpu_round2 import _expert_kit
    nb = 4
    groups = []
    for batched in (True, False):
        make, bufs, dims, shapes, ids, tr = _expert_kit(normalize_obs=True)
        if batched:
            bank = ExpertBank(lambda t, **h: make(t, use_graph=use_graph, **h), nb)
            xs = list(bank)
        else:
            bank, xs = None, [make(t, use_graph=use_graph) for t in range(nb)]
        rng = np.random.RandomState(2)
        for e in (0, 2, 3, 0):                                       # stores through different experts: different statistics
            np.random.seed(1 + e)
            xs[e].store_episode({k: v.copy() for k, v in synth_episodes(rng, 16, nb, 40).items()}, np.zeros(nb), 16)
        groups.append((bank, xs))
    (bank, bx), (_, sx) = groups
    for a, b in zip(bx, sx):
        assert torch.equal(a.o_stats.state, b.o_stats.state) and torch.equal(a.g_stats.state, b.g_stats.state)
    assert not torch.equal(bx[0].o_stats.state, bx[2].o_stats.state)
    assert float(bx[1].o_stats.state[2 * 40]) == 1.0                   # expert 1 never stored: count 1, mean 0, std 1
    for n in (1, 1, 22, 3):
        bank.train_batches(n)
        for x in sx:
            x.train_batches(n)
    torch.cuda.synchronize()
    assert bank.batched
    for e, (a, b) in enumerate(zip(bx, sx)):
        for name in ('theta', '_m', '_v'):
            assert torch.equal(getattr(a, name), getattr(b, name)), (e, name)
        assert float(a._losses[0]) == float(b._losses[0]) and np.isfinite(float(a._losses[0]))
    # the statistics matter: an expert with other statistics ends elsewhere
    assert not torch.equal(bx[0].theta, bx[2].theta)
    bank.check_faults()


# ------------------------------------------------------------------ the epoch outside the cycles
def test_evaluation_rollouts_enqueued_together_equal_one_at_a_time():
    """train.py:156-158: `for _ in range(n_test_rollouts): evaluator.generate_rollouts()`.  generate_eval_rollouts(n)
    enqueues all n and waits once: same NumPy draws, same success / Q histories, same competence queues, same episode
    counts as n calls of generate_rollouts()."""
    import pickle
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from test_gpu_agent import T, build_pair
    nb, dimo, B = 4, 40, 16
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    outs = []
    for together in (False, True):
        agent, _ = build_pair(nb, dimo, rng_mode='device', use_graph=True, seed=4)
        ev = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                           exploit=True, use_target_net=False, compute_Q=True, structure='curious',
                           task_selection='active_competence_progress', queue_length=6, eval=True)
        ev.seed(21)
        ev.EVAL_SLOTS = 0          # (the one-launch-per-rollout form; several rollouts to a launch: tests/test_gpu_round6.py)
        np.random.seed(17)
        for epoch in range(2):
            ev.clear_history()
            if together:
                ev.generate_eval_rollouts(5)
            else:
                for _ in range(5):
                    ev.generate_rollouts()
        outs.append(dict(succ=list(ev.success_history), Q=list(ev.Q_history), n=ev.n_episodes, C=np.array(ev.get_C()),
                         tasks=list(ev.task_history), draw=np.random.random(), logs=ev.logs('test'),
                         staging=ev.benv.staging.clone()))
    a, b = outs
    assert a['succ'] == b['succ'] and a['Q'] == b['Q'] and a['n'] == b['n'] == 2 * 5 * B
    assert len(a['succ']) == 5 and np.isfinite(a['Q']).all()
    np.testing.assert_array_equal(a['C'], b['C'])
    assert a['tasks'] == b['tasks'] and a['draw'] == b['draw'] and a['logs'] == b['logs']
    assert torch.equal(a['staging'], b['staging'])


def test_policy_snapshot_pickles_like_the_policy_and_is_written_in_the_background(tmp_path):
    """train.py:195-205 off the training thread: RolloutWorker.save_policy hands a host snapshot (PolicySnapshot) to a
    BackgroundWriter; the files are byte for byte what the synchronous form writes, in submission order, and a failing
    job surfaces at close()."""
    import pickle
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from curious_amd.util import BackgroundWriter, PolicySnapshot
    from test_gpu_agent import T, build_pair
    agent, _ = build_pair(4, 40, rng_mode='device', seed=4)
    snap = PolicySnapshot(agent)
    assert pickle.dumps(snap) == pickle.dumps(agent)
    clone = pickle.loads(pickle.dumps(snap))
    assert type(clone) is type(agent) and torch.equal(clone.theta, agent.theta)
    dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
    ev = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=4,
                       structure='curious', task_selection='random', queue_length=6, eval=True)
    sync = str(tmp_path / 'sync.pkl')
    ev.save_policy(sync)
    ev.writer = BackgroundWriter()
    back = str(tmp_path / 'back.pkl')
    ev.save_policy(back)
    agent.theta.add_(1.0)                                            # the snapshot was taken when save_policy was called
    ev.writer.close()
    for suffix in ('', '_weights.pkl'):
        assert open(sync + suffix, 'rb').read() == open(back + suffix, 'rb').read()
    ev.writer = BackgroundWriter()
    ev.save_policy(str(tmp_path / 'no_such_dir' / 'x.pkl'))
    with pytest.raises(OSError):
        ev.writer.close()


# ------------------------------------------------------------------ fused IPC all-reduce + Adam (opt-in)
def test_ipc_allreduce_adam_on_one_rank_equals_the_standalone_optimiser():
    """curious_allreduce_adam_ipc with world = 1 (the rank's own buffers as its only peer): the sum over ranks is the
    gradient itself, so parameters, moments and the rebuilt transposed copies must equal curious_adam_update with `keep`
    bit for bit -- mpi_adam.py:29-35 once more, and the index arithmetic of slices, step-size ring and copies."""
    from curious_amd import _lib, ops
    from test_gpu_round3 import _filled_agent
    agent, _ = _filled_agent()
    agent.train_batches(3)
    torch.cuda.synchronize()
    # a gradient for the current parameters, step counter advanced as the several-rank gradient call does
    agent._train_device_prologue(1)
    agent._sample_packed()
    agent._grads_next(agent._cur)
    torch.cuda.synchronize()
    keep = agent._kept_copies()
    state = [x.clone() for x in (agent.theta, agent._m, agent._v, agent._workspace)]
    ops.adam_update(agent.theta, agent._m, agent._v, agent.grad, agent.off_pi, agent.P_total - agent.off_pi,
                    alpha_tab=agent._alpha_tab, step_ctr=agent._step_ctr, tab_base=agent._alpha_base, keep=keep)
    torch.cuda.synchronize()
    want = [x.clone() for x in (agent.theta, agent._m, agent._v, agent._workspace)]
    for x, s in zip((agent.theta, agent._m, agent._v, agent._workspace), state):
        x.copy_(s)
    flags = torch.zeros(16, dtype=torch.int32, device='cuda')
    words = torch.zeros(3, dtype=torch.int32, device='cuda')         # blocks done | a wait gave up | epoch
    words[2] = 41                                                    # (tokens are epochs, not Adam steps)
    stage = torch.zeros_like(agent.theta)                            # where the new slices land before they become theta
    peers = _lib.IpcPeers()
    peers.world, peers.rank = 1, 0
    peers.grad[0], peers.stage[0], peers.flags[0] = agent.grad.data_ptr(), stage.data_ptr(), flags.data_ptr()
    ops.allreduce_adam_ipc(peers, agent.theta, agent._m, agent._v, agent.off_pi, agent.P_total - agent.off_pi,
                           agent._alpha_tab, agent._step_ctr, agent._alpha_base, words[2:3], words[0:1], words[1:2], keep)
    torch.cuda.synchronize()
    assert int(words[1]) == 0 and int(words[0]) == 64 and int(words[2]) == 42
    assert flags[0].item() == 42 and flags[8].item() == 42             # ready / landed words carry the epoch's token
    assert torch.equal(stage, agent.theta)
    for name, got, w in zip(('theta', 'm', 'v', 'workspace'), (agent.theta, agent._m, agent._v, agent._workspace), want):
        assert torch.equal(got, w), name
    assert not torch.equal(agent.theta, state[0])


@pytest.mark.parametrize('world', [2, 4])
def test_ipc_allreduce_adam_ranks_on_one_gpu(world):
    """DDPG(_allreduce='ipc') (env CURIOUS_ALLREDUCE=ipc): `world` processes share this GPU, map each other's gradient /
    parameter vectors through CUDA-IPC handles and run 35 updates + 2 whole cycles of the bench job -- gradients SUMMED in
    rank order, Adam on the owned slice, new slices handed round, transposed copies rebuilt in the same kernel.  Every
    rank ends with the same parameters (check_synced on the way); at 2 ranks, where a sum of two is the same in any order,
    they are bit for bit those of the default path (gloo all-reduce + stand-alone optimiser)."""
    digests = {}
    for mode in (('ipc', 'rccl') if world == 2 else ('ipc',)):
        prefix = os.path.join(tempfile.mkdtemp(), 'digest')
        env = _two_rank_env(CURIOUS_RANK_CHECK_CYCLES='2', CURIOUS_RANK_CHECK_OUT=prefix, CURIOUS_ALLREDUCE=mode,
                            HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
               os.path.join(ROOT, 'tools', 'rank_path_check.py')]
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (mode, out.stdout[-1500:], out.stderr[-3000:])
        found = [open('%s.rank%d' % (prefix, r)).read().split() for r in range(world)]
        assert len({f[1] for f in found}) == 1 and len(found[0][1]) == 64, (mode, found)
        assert all(f[2] == '235' for f in found)
        digests[mode] = found[0][1]
    if world == 2:
        assert digests['ipc'] == digests['rccl'], digests


# ------------------------------------------------------------------ custom-op face of the three hot entry points
def test_torch_custom_op_face_of_the_hot_entry_points():
    """north_star: "exposed to Python via PyTorch-ROCm custom ops".  torch.ops.curious_hip.{her_sample, ddpg_update,
    policy_rollout} take an opaque descriptor (torch_ops.desc_create: the struct-shaped arguments, filed once) + tensors
    and call the same symbols as curious_amd.ops: two identically seeded jobs, one driven through each face, end bit for
    bit alike."""
    import curious_amd.torch_ops as T
    from curious_amd import ops
    from curious_amd.envs import REWARD_EPS
    from test_gpu_agent import T as horizon

    def run(face):
        np.random.seed(5)
        agent, w = _job(use_graph=False, seed=11)
        S = agent.sample_transitions
        env = w.benv
        # -- her_sample: one device-drawn minibatch
        agent._train_device_prologue(1)
        P = S.params(agent.clip_obs, agent.relative_goals)
        batch = agent._pp[0]
        if face == 'ops':
            ops.her_sample(agent._pool.storage, agent._pool.buf_stride, agent._layout, S.tasks, P, agent.batch_size, batch,
                           rng=agent._rng_desc)
        else:
            d = T.desc_create(layout=agent._layout, tasks=S.tasks, params=P, rng=agent._rng_desc,
                              buf_stride=agent._pool.buf_stride, n=agent.batch_size, keep=agent._tables)
            torch.ops.curious_hip.her_sample(d, agent._pool.storage, batch)
        # -- ddpg_update: three fused updates (gradients + Adam + the gather of the next batch)
        if face != 'ops':
            du = T.desc_create(cfg=agent.net_cfg, layout=agent._layout, B=agent.batch_size, tab_base=agent._alpha_base,
                               tasks=S.tasks, params=P, rng=agent._rng_desc, buf_stride=agent._pool.buf_stride)
        for k in range(3):
            p = k & 1
            if face == 'ops':
                agent._update_fused(p, chained=k > 0)
            else:
                torch.ops.curious_hip.ddpg_update(du, agent.theta, agent.theta_target, agent._pp[p], agent._workspace,
                                                  agent.grad, agent._losses, agent._Q_pi, agent._m, agent._v,
                                                  agent._step_ctr, agent._alpha_tab, agent._pp[p ^ 1], agent._pool.storage,
                                                  k > 0)
        # -- policy_rollout: a whole T-step rollout of the batched env in one launch
        env.reset_all(np.arange(env.n) % 4, np.linspace(-1, 1, 3 * env.n, dtype=np.float32).reshape(env.n, 3))
        ws = torch.zeros(ops.workspace_floats(agent.net_cfg, env.n), device='cuda')
        u = torch.empty([env.n, 4], device='cuda')
        base = torch.zeros(1, dtype=torch.int64, device='cuda')
        seed = 987654321
        if face == 'ops':
            ops.policy_rollout(agent.net_cfg, agent.theta, env.n, agent.clip_obs, ws, 0.2, 0.3, seed, 1, u, env._cfg,
                               env.layout, env.env_id0, env.episode, env.tasks, 0, horizon, env.o, env.ag, env.g, env.td,
                               env.staging, REWARD_EPS, counter_base=base, flags=env.flags)
        else:
            dr = T.desc_create(cfg=agent.net_cfg, ecfg=env._cfg, layout=env.layout, n=env.n, clip_obs=agent.clip_obs,
                               noise_scale=0.2, random_eps=0.3, seed=seed, counter=1, env_id0=env.env_id0, t0=0,
                               nsteps=horizon, reward_eps=REWARD_EPS)
            torch.ops.curious_hip.policy_rollout(dr, agent.theta, ws, u, base, env.episode, env.tasks, env.o, env.ag, env.g,
                                                 env.td, env.staging, env.flags)
            for h in (d, du, dr):
                T.desc_free(h)
        torch.cuda.synchronize()
        return dict(batch=batch.clone(), theta=agent.theta.clone(), m=agent._m.clone(), losses=agent._losses.clone(),
                    staging=env.staging.clone(), flags=env.flags.clone(), u=u.clone(), ctr=agent._step_ctr.clone())

    a, b = run('ops'), run('torch')
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(a['u'].abs().sum()) > 0 and int(a['ctr']) == 3
    with pytest.raises(Exception):
        torch.ops.curious_hip.her_sample(12345, a['batch'], a['batch'])             # unknown descriptor


# ---------------------------------------------------------------------------------------------------------------------
# the row-local launch's in-kernel synchronisation (one workgroup barrier per layer, partial tiles alternating between two
# LDS buffers, the Q' hand-off between workgroups): the same gradient call many times, every output bit for bit the first's
@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['single', 'experts'])
def test_repeated_gradient_calls_are_bit_identical(mode):
    import subprocess
    n = '3000' if mode == 'single' else '1000'
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'rows_determinism.py'), n] + (['experts'] if mode == 'experts' else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-400:] + p.stderr[-400:]
    assert 'all identical to the first' in p.stdout


# ---------------------------------------------------------------------------------------------------------------------
# speed-only options of the update's two launches change no number
@pytest.mark.gpu
@pytest.mark.parametrize('option', ['rows_pre', 'rows_xcd', 'dw_xcd'])
def test_placement_and_argument_options_change_no_result(option):
    """rows_pre (the row-local launch's role / first loads from leading kernel arguments, mlp_rows.h RowsPre), rows_xcd and
    dw_xcd (XCD-aware block placement of the two launches): 13 updates, eager and as chained graphs, with the option off
    end bit for bit where they end with it on -- parameters, moments, gradients, losses, the next staged batch."""
    from curious_amd import ops
    from test_gpu_round3 import _filled_agent
    assert ops.get_option(option) == 1
    agents = []
    for on in (1, 0):
        with ops.option(option, on):
            a, _ = _filled_agent(use_graph=False)
            g, _ = _filled_agent(use_graph=True)
            for _ in range(7):
                a.train()
            g.train_batches(7)
            a.train_batches(6)
            g.train_batches(6)
            torch.cuda.synchronize()
            a.check_faults()
            g.check_faults()
            agents.append((a, g))
    for x, y in zip(agents[0], agents[1]):
        assert torch.equal(x.theta, y.theta) and torch.equal(x._m, y._m) and torch.equal(x._v, y._v)
        assert torch.equal(x._staged, y._staged) and torch.equal(x.grad, y.grad)
        assert torch.equal(x._losses, y._losses) and torch.equal(x._Q_pi, y._Q_pi)
        assert int(x._step_ctr) == int(y._step_ctr) == 13
    assert torch.equal(agents[0][0].theta, agents[0][1].theta)      # eager == graph
