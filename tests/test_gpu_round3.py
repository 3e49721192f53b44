"""Round-3 GPU tests: the guard on the in-kernel Q' hand-off (fault injection), NaN rollouts on the device-routed
path, the wide-observation normaliser fallback, run-time options."""
import numpy as np
import pytest
import torch

from test_gpu_agent import T, build_pair, synth_episodes
from test_gpu_round2 import _expert_kit

pytestmark = pytest.mark.gpu


def _filled_agent(**kw):
    agent, _ = build_pair(4, 40, rng_mode='device', **kw)
    ep = synth_episodes(np.random.RandomState(5), 40, 4, 40)
    np.random.seed(2)
    agent.store_episode({k: v.copy() for k, v in ep.items()}, np.array([0.3, 0.0, 0.2, 0.1]), 40)
    return agent, ep


def test_options_are_read_per_call():
    from curious_amd import _lib, ops
    assert ops.get_option('rows') == 1 and ops.get_option('fault_inject') == 0 and ops.get_option('qt_spins') == 1 << 22
    with ops.option('rows', 0):
        assert ops.get_option('rows') == 0
    assert ops.get_option('rows') == 1
    with pytest.raises(_lib.CuriousHipError):
        ops.set_option('no_such_option', 1)
    with pytest.raises(_lib.CuriousHipError):
        ops.set_option('xcd_map', 3)


@pytest.mark.parametrize('use_graph', [False, True])
def test_handoff_fault_is_detected_and_the_optimiser_skipped(use_graph):
    """A target group that never publishes Q' (option fault_inject): the consumers give up after qt_spins polls, the
    loss is NaN, the workspace's fault word counts the 4 waves, the optimiser of that update -- and of every later one
    until the word is cleared -- leaves theta / m / v / the transposed copies alone; DDPG.check_faults raises and clears;
    the next update is clean.  Eager launches and a captured graph (the option is baked into the capture)."""
    from curious_amd import ops
    from curious_amd.ddpg import HandoffFault
    agent, ep = _filled_agent(use_graph=use_graph)
    agent.train_batches(3)
    agent.check_faults()                                             # nothing to report
    torch.cuda.synchronize()
    before = [x.clone() for x in (agent.theta, agent._m, agent._v, agent.theta_target)]
    kept = ops.ddpg_transposed(agent.net_cfg, agent.batch_size, agent._workspace)
    assert kept.n == 4 and kept.fault == ops.fault_word(agent.net_cfg, agent.batch_size, agent._workspace).data_ptr()
    ws_before = agent._workspace.clone()
    fault = ops.fault_word(agent.net_cfg, agent.batch_size, agent._workspace)
    with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
        if use_graph:
            agent._graphs = [None, None]                             # capture anew, with the injected fault baked in
        loss, _ = agent.train()
        torch.cuda.synchronize()
    # the 4 waves (batch rows) of row group 2 -- twice with a graph: the eager warm-up of the capture, then the replay
    assert int(fault) == (8 if use_graph else 4)
    assert not np.isfinite(float(loss))
    agent.train()                                                    # the word is sticky: this update is skipped as well
    torch.cuda.synchronize()
    if use_graph:
        agent._graphs = [None, None]                                 # (drop the graphs that carry the injection)
    for a, b in zip(before, (agent.theta, agent._m, agent._v, agent.theta_target)):
        assert torch.equal(a, b)
    off = (kept.dst[0] - agent._workspace.data_ptr()) // 4           # the transposed copies: untouched
    n_copy = 4 * 256 * 256
    assert torch.equal(ws_before[off:off + n_copy], agent._workspace[off:off + n_copy])
    # the training loop's form of the check: enqueued by update_target_net, read by the next store_episode
    agent.update_target_net()
    torch.cuda.synchronize()
    with pytest.raises(HandoffFault):
        agent.store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(4), 80)
    assert int(fault) == 0                                           # cleared by the check
    agent.check_faults()
    loss, _ = agent.train()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and not torch.equal(before[0], agent.theta)
    agent.check_faults()


def test_handoff_fault_with_batched_experts():
    """N = 4 experts: 768 workgroups compete for 256 CUs; every expert's row group 5 loses its producer.  Every expert
    reports, no expert's parameters move, the bank trains on afterwards."""
    from curious_amd import ops
    from curious_amd.ddpg import HandoffFault
    from curious_amd.experts import ExpertBank
    nb = 4
    make, bufs, dims, shapes, ids, tr = _expert_kit()
    bank = ExpertBank(lambda t, **h: make(t, **h), nb)
    ep = synth_episodes(np.random.RandomState(2), 40, nb, 40)
    np.random.seed(1)
    bank[0].store_episode({k: v.copy() for k, v in ep.items()}, np.zeros(nb), 40)
    bank.train_batches(2)
    bank.check_faults()
    torch.cuda.synchronize()
    before = bank.slab.clone()
    x0 = bank[0]
    with ops.option('fault_inject', 6), ops.option('qt_spins', 50000):
        bank.train_batches(1)
        torch.cuda.synchronize()
    assert bank.batched
    for x in bank:
        assert int(ops.fault_word(x.net_cfg, x.batch_size, x._workspace)) == 4
        n = x.theta.numel()
        for name in ('theta', '_m', '_v', 'theta_target'):
            t = getattr(x, name)
            off = (t.data_ptr() - bank.slab.data_ptr()) // 4
            assert torch.equal(before.view(-1)[off:off + n], t), name
        assert not np.isfinite(float(x._losses[0]))
    n_raised = 0
    for x in bank:
        with pytest.raises(HandoffFault):
            x.check_faults()
        n_raised += 1
    assert n_raised == nb
    bank.check_faults()                                              # all cleared
    bank.train_batches(3)
    torch.cuda.synchronize()
    bank.check_faults()
    assert all(np.isfinite(float(x._losses[0])) for x in bank)
    assert not torch.equal(before.view(-1)[:x0.theta.numel()], x0.theta)


def test_nan_rollout_on_the_device_routed_path_feeds_nothing():
    """async_store: a rollout whose observations turn NaN is neither stored nor fed to the normalisers (the reference
    regenerates it before store_episode ever sees it, rollout.py:268-271); the worker sees it a cycle late, when it
    settles the flags, and generates + stores the replacement then (round 4; round 3 only reported the loss); the NaN
    word is cleared by the next reset."""
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    nb, dimo, B = 4, 40, 16
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    agent, _ = build_pair(nb, dimo, cap_eps=100, rng_mode='device', use_graph=False)
    agent.async_store = True
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.0, structure='curious', task_selection='random', queue_length=6,
                      eval=False)
    w.seed(5)
    np.random.seed(8)
    w._decide_exploit = lambda: None                                 # no exploit rollouts (they wait for the flags)
    agent.store_episode(synth_episodes(np.random.RandomState(21), 24, nb, dimo), w.CP, 24)
    ep, cp, n_ep = w.generate_rollouts()
    assert getattr(w, '_pending', None) is not None                   # the async form applies
    agent.store_episode(ep, cp, n_ep)
    agent.train_batches(2)
    w.settle(); agent.settle()
    torch.cuda.synchronize()
    sizes = [b.current_size for b in agent.buffer]
    stats = (agent.o_stats.state.clone(), agent.g_stats.state.clone(), agent._stats_acc.clone())
    launch_reset = w.benv.launch_reset                               # (the reset launch heads the rollout: DDPG.act_rollout)

    def poisoned_reset(counter=None, delta=0):                       # one env starts from a NaN observation
        launch_reset(counter=counter, delta=delta)
        w.benv.o[3, 20] = float("nan")                               # an entry the env carries along unchanged
    w.benv.launch_reset = poisoned_reset
    ep, cp, n_ep = w.generate_rollouts()
    w.benv.launch_reset = launch_reset
    assert getattr(w, '_pending', None) is not None
    agent.store_episode(ep, cp, n_ep)
    torch.cuda.synchronize()
    assert float(w.benv.flags[B]) == 1.0
    agent.settle()
    assert [b.current_size for b in agent.buffer] == sizes            # nothing stored
    for a, b in zip(stats, (agent.o_stats.state, agent.g_stats.state, agent._stats_acc)):
        assert torch.equal(a, b)                                      # nothing accumulated
    assert torch.isfinite(agent.o_stats.state).all()
    n_before = w.n_episodes
    w.settle()                                                       # the lost rollout is replaced NOW (rollout.py:268-271)
    agent.settle()
    torch.cuda.synchronize()
    assert float(w.benv.flags[B]) == 0.0                             # the replacement's reset cleared the NaN word
    replaced = [b.current_size for b in agent.buffer]
    assert sum(replaced[1:nb + 1]) > sum(sizes[1:nb + 1])
    assert w.n_episodes == n_before                                  # - the lost rollout + its replacement
    assert not torch.equal(stats[0], agent.o_stats.state) and torch.isfinite(agent.o_stats.state).all()
    sizes = replaced
    ep, cp, n_ep = w.generate_rollouts()                             # the next reset clears the NaN word
    agent.store_episode(ep, cp, n_ep)
    w.settle(); agent.settle()
    torch.cuda.synchronize()
    assert float(w.benv.flags[B]) == 0.0
    assert sum(b.current_size for b in agent.buffer[1:nb + 1]) > sum(sizes[1:nb + 1])


def test_wide_observations_take_the_per_normaliser_path():
    """dimo + dimg > 256: store_episode feeds the two normalisers one after the other (curious_norm_update +
    curious_norm_recompute) instead of the paired launch; same statistics as the oracle."""
    nb, dimo = 4, 250
    agent, oracle = build_pair(nb, dimo, rng_mode='numpy')
    ep = synth_episodes(np.random.RandomState(7), 12, nb, dimo)
    cp = np.zeros(nb)
    np.random.seed(4)
    agent.store_episode({k: v.copy() for k, v in ep.items()}, cp, 12)
    np.random.seed(4)
    oracle.store_episode({k: v.astype(np.float64) for k, v in ep.items()}, cp, 12)
    for nz, onz in ((agent.o_stats, oracle.o_stats), (agent.g_stats, oracle.g_stats)):
        np.testing.assert_allclose(nz.mean.cpu().numpy(), onz.mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nz.std.cpu().numpy(), onz.std, rtol=1e-5, atol=1e-6)


def test_synthetic_env_refuses_observations_wider_than_its_step_kernel():
    from curious_amd import _lib, ops
    from curious_amd.layout import RecordLayout
    nb, dimo, n, Tn = 4, 132, 4, 5
    G = 3 * nb
    shapes = dict(o=(Tn + 1, dimo), u=(Tn, 4), g=(Tn, G), ag=(Tn + 1, G), info_is_success=(Tn, 1),
                  task_descr=(Tn, nb), change=(Tn, G))
    lay = RecordLayout(shapes, Tn)
    ecfg = ops.make_env_cfg(nb, dimo, Tn, 1)
    z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device='cuda')
    with pytest.raises(_lib.CuriousHipError, match='at most 128'):
        ops.env_step(ecfg, lay, 0, z(n, dtype=torch.int32), z(n, dtype=torch.int32), z(n, 4), 0, n, z(n, dimo), z(n, G),
                     z(n, G), z(n, nb), z(n, Tn + 1, lay.row_stride), 0.05)


def _run_rank_check(extra, nproc=1, timeout=900):
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    prefix = os.path.join(tempfile.mkdtemp(), 'digest')
    env = dict(os.environ, CURIOUS_RANK_CHECK_OUT=prefix)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CURIOUS_FORCE_DIST', 'CURIOUS_GRAPH_ALLREDUCE', 'CURIOUS_DIST_BACKEND'):
        env.pop(k, None)
    script = os.path.join(root, 'tools', 'rank_path_check.py')
    if nproc == 1:
        env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
               '--master-addr', '127.0.0.1', '--master-port', str(port), script]
    env.update(extra)
    if nproc > 1:
        env['CURIOUS_RESIDENT'] = '0'           # several processes on ONE device must not both claim every CU
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    return [open('%s.rank%d' % (prefix, r)).read().split() for r in range(nproc)]


def test_data_parallel_batched_experts_match_the_single_rank_bank():
    """BASELINE configs[4] with several ranks: gradient launches of all experts -> ONE all-reduce of the [N, P] block ->
    optimiser launch of all experts.  On a one-rank RCCL communicator (eager launches; split graphs around an eager
    collective; the collective captured in the chain graph; whatever the self-test picks) the parameters of all 4
    experts are bit-identical to the fused single-rank bank."""
    ex = {'CURIOUS_RANK_CHECK_STRUCTURE': 'task_experts'}
    runs = [dict(ex),
            dict(ex, CURIOUS_FORCE_DIST='1', CURIOUS_RANK_CHECK_GRAPH='0'),
            dict(ex, CURIOUS_FORCE_DIST='1', CURIOUS_GRAPH_ALLREDUCE='0'),
            dict(ex, CURIOUS_FORCE_DIST='1', CURIOUS_GRAPH_ALLREDUCE='1'),
            dict(ex, CURIOUS_FORCE_DIST='1')]
    found = [_run_rank_check(e)[0] for e in runs]
    assert all(f[2] == '35' for f in found), found
    assert len({f[1] for f in found}) == 1, found


def test_two_ranks_of_batched_experts_share_one_gpu_over_gloo():
    """World size 2 on ONE GPU over gloo: rank-private buffers / rollouts / RNG streams, the [4, P] gradient block summed
    by one collective per update, check_synced on the way: both ranks end with bit-identical experts."""
    found = _run_rank_check({'CURIOUS_RANK_CHECK_STRUCTURE': 'task_experts', 'CURIOUS_DIST_BACKEND': 'gloo',
                             'CURIOUS_RANK_CHECK_CYCLES': '2'}, nproc=2)
    assert found[0][1] == found[1][1] and len(found[0][1]) == 64, found
    assert found[0][2] == found[1][2] == '235'


def test_gradients_with_the_next_gather_riding_along(route):
    """curious_ddpg_grads(next): the HER gather of the next update's batch in spare workgroups of the row-local launch
    (the step counter's increment deferred to the weight-gradient launch) == curious_ddpg_grads followed by
    curious_her_sample, bit for bit: gradients, losses, counter, the gathered batch.  (The tiled route: the gather is a
    launch behind the gradients.)"""
    from curious_amd import ops
    agent, _ = _filled_agent(use_graph=False)
    agent._train_device_prologue(1)
    agent._sample_packed()
    S = agent.sample_transitions
    kw = dict(storage=agent._pool.storage, buf_stride=agent._pool.buf_stride, tasks=S.tasks,
              params=S.params(agent.clip_obs, agent.relative_goals), rng=agent._rng_desc)
    outs = []
    for riding in (True, False):
        agent._step_ctr.fill_(7)
        agent._pp[1].fill_(-3.0)
        agent.grad.fill_(float('nan'))
        ops.ddpg_grads(agent.net_cfg, agent.theta, agent.theta_target, agent._pp[0], agent._layout, agent.batch_size,
                       agent._workspace, agent.grad, agent._losses, agent._Q_pi, step_ctr=agent._step_ctr,
                       **(dict(next_batch=agent._pp[1], **kw) if riding else {}))
        if not riding:
            ops.her_sample(agent._pool.storage, agent._pool.buf_stride, agent._layout, S.tasks, kw['params'],
                           agent.batch_size, agent._pp[1], rng=agent._rng_desc)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in (agent.grad, agent._losses, agent._Q_pi, agent._step_ctr, agent._pp[1])])
    for a, b in zip(*outs):
        assert torch.equal(a.nan_to_num(nan=12345.0), b.nan_to_num(nan=12345.0))     # (the pads of `grad` stay NaN)
    assert int(outs[0][3]) == 8 and bool((outs[0][4][:, :40] != -3.0).all())  # counter advanced once, batch overwritten


def test_resident_rollout_reports_a_member_that_never_shows_up():
    """policy_resident_kernel: the 4 workgroups of a group wait for each other.  With one member missing (option
    fault_inject: it exits at once, like a workgroup that was never scheduled) its peers give up after res_spins polls, the
    launch ENDS (no hang), flags[n] = 2 and the worker's flag fetch raises; the next rollout is clean."""
    from curious_amd import _lib, logger, ops
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    nb, dimo, B = 4, 40, 64
    dims = dict(o=dimo, u=4, g=12, ag=12, task_descr=nb, info_is_success=1)
    agent, _ = build_pair(nb, dimo, rng_mode='device', use_graph=False)
    w = RolloutWorker(EnvFactory('MultiTaskFetchArm4-v5'), agent, dims, logger, T=T, rollout_batch_size=B,
                      noise_eps=0.2, random_eps=0.3, structure='curious', task_selection='random', queue_length=6,
                      eval=False)
    w.seed(5)
    np.random.seed(8)
    before = ops.prof_launch_counts()['policy_resident_kernel']
    ep, _, _ = w.generate_rollouts()
    torch.cuda.synchronize()
    assert ops.prof_launch_counts()['policy_resident_kernel'] == before + 1     # the resident route is the one taken
    assert float(w.benv.flags[B]) == 0.0
    from curious_amd.envs import ResidentRolloutVoid
    try:
        with ops.option('fault_inject', 3), ops.option('res_spins', 20000):
            w.benv.reset_all(np.zeros(B, np.int64), np.zeros([B, 3], np.float32))
            agent.act_rollout(w.benv, T, noise_eps=0.2, random_eps=0.3)
            with pytest.raises(ResidentRolloutVoid, match='gave up waiting'):      # a CuriousHipError
                w.benv.fetch_flags()
            assert float(w.benv.flags[B]) == 2.0
            # the worker (round 4) does not pass the error on: it switches the process to the streaming kernel and
            # generates the rollout again (tests/test_gpu_round4.py: with the same numbers)
            ep, _, _ = w.generate_rollouts()
            assert ops.get_option('resident') == 0
        torch.cuda.synchronize()
        assert float(w.benv.flags[B]) == 0.0 and bool(torch.isfinite(ep.records).all())
        assert issubclass(ResidentRolloutVoid, _lib.CuriousHipError)
    finally:
        ops.set_option('resident', 1)


def test_ipc_allreduce_adam_prototype_two_processes_one_gpu(tmp_path):
    """tools/ipc_allreduce_lab.hip ("next step"): reduce-scatter over IPC-mapped peer gradients + Adam on
    the owned slice + all-gather of the new slices as ONE kernel per rank.  Two processes share this GPU and map each
    other's buffers through hipIpc handles; every rank's parameters must equal the host model of SUM-in-rank-order + Adam
    bit for bit, and each other.  (Functional only: on one device there is no wire to time.)"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    exe = str(tmp_path / 'ipc_allreduce_lab')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-ffp-contract=off',
                    os.path.join(root, 'tools', 'ipc_allreduce_lab.hip'), '-o', exe], check=True, timeout=600)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([exe, '2'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith('OK'), (out.stdout[-1500:], out.stderr[-1500:])
    assert 'bit for bit' in out.stdout and 'replicas identical' in out.stdout


def test_training_learns_the_synthetic_arm_and_reproduces_the_committed_curve(tmp_path):
    """End to end: launch() with the throughput configuration (256 GPU-resident envs, device RNG, hipGraphs, async store,
    resident rollout) for 70 epochs of 25 cycles x 40 updates.  The test success rate must follow the committed learning
    curve (profiles/r03_learning_curve_arm4.csv, same seed: the run is deterministic) and end above 0.9 -- every kernel
    of the cycle takes part, a silent numerical regression anywhere shows up here."""
    import csv
    import os
    from curious_amd.experiment import config, train
    config.CACHED_ENVS.clear()
    over = dict(rollout_batch_size=256, n_cycles=25, n_batches=40, rng_mode='device', use_graph=True, async_store=True)
    np.random.seed(0)
    best = train.launch(env='MultiTaskFetchArm4-v5', trial_id=0, n_epochs=70, num_cpu=1, seed=1, policy_save_interval=0,
                        clip_return=1, normalize_obs=False, structure='curious',
                        task_selection='active_competence_progress', goal_selection='random', goal_replay='her',
                        task_replay='replay_task_cp_buffer', save_policies=False, override_params=over,
                        save_root=str(tmp_path) + '/')
    rows = list(csv.DictReader(open(os.path.join(str(tmp_path), 'MultiTaskFetchArm4-v5', '0', 'progress.csv'))))
    got = [float(r['test/success_rate']) for r in rows]
    assert len(got) == 71 and max(got) >= 0.95 and got[-1] >= 0.9, got[-5:]       # (`best` is only tracked when saving)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = list(csv.DictReader(open(os.path.join(root, 'profiles', 'r03_learning_curve_arm4.csv'))))
    want = [float(r['test/success_rate']) for r in ref[:71]]
    assert got == want, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:5]
