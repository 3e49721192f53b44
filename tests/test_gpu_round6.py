"""Round 6: the 16-row form of the row-local update (csrc/mlp_rows16.h: v_mfma_f32_16x16x4, the waves of a workgroup
split the output columns), resumable training jobs, bench.py on the published rank layout."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('V,graph', [(2, 0), (2, 1), (3, 0), (8, 1)])
def test_sixteen_rows_per_workgroup_match_the_oracle(V, graph):
    """Option rows16 = N: batches of >= N rows take the 16-row form (default 1 280 rows = 5 ranks; here forced from 1 row
    on).  Another order of summation over k than the 4- / 8-row forms, so the check is the oracle's: the V-rank test of
    round 5 (per-rank losses 1e-5, one oracle Adam step from the summed oracle gradients, normalisers, Polyak) run through
    the 16-row kernels -- ddpg.py:419-449, actor_critic.py:51-98, util.py:73-107, mpi_adam.py:26-35."""
    from curious_amd import ops
    from test_gpu_round5 import test_virtual_ranks_match_the_oracle_rank_model as check
    before = ops.prof_launch_counts()
    with ops.option('rows', 1), ops.option('rows16', 1):
        check(V, graph, 'rows')
    assert sum(ops.prof_launch_counts().values()) > sum(before.values())


@pytest.mark.parametrize('V', [1, 3, 19])
def test_sixteen_rows_against_the_eight_row_form(V):
    """Same batches (the draws do not depend on the form), per-rank losses equal to 1e-6 relative, Q_pi to 1e-5, the summed
    gradient behind the first update within 2e-6 of its max-norm: the two forms differ in the ORDER of every sum over k and
    in nothing else.  Also with the gather of the next batch inside the launch (graphs: ddpg_rows16_her_kernel)."""
    from curious_amd import ops
    from test_gpu_round5 import make_agent, rank_episodes
    outs = []
    for rows16 in (0, 1):
        with ops.option('rows16', rows16):
            agent = make_agent(V, use_graph=True)
            draw = rank_episodes(V, 12)
            agent.store_episode(draw(), np.array([0.3, 0.0, 0.2, 0.1]), 12 * V)
            agent.train()
            torch.cuda.synchronize()
            first = [t.cpu().numpy().copy() for t in (agent._losses, agent._Q_pi, agent._m, agent._pp[0])]
            agent.train_batches(11)
            agent.update_target_net()
            agent.train_batches(4)
            torch.cuda.synchronize()
            agent.check_faults(wait=True)
            outs.append(first + [agent.theta.cpu().numpy().copy(), agent._losses.cpu().numpy().copy()])
    a, b = outs
    np.testing.assert_array_equal(a[3], b[3])                        # the staged batch: same draws
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
    # m after the first update = 0.1 x the summed gradient
    dev = np.abs(a[2] - b[2]) / np.abs(a[2]).max()
    assert dev.max() <= 5e-4 and (dev > 2e-6).mean() <= 1e-3, (dev.max(), (dev > 2e-6).mean())
    assert np.isfinite(b[5]).all() and np.abs(a[4] - b[4]).max() <= 2e-2   # 16 Adam steps of 1e-3 apart at the very most
    assert (np.abs(a[4] - b[4]) > 1e-4).mean() < 1e-2


@pytest.mark.parametrize('V,graph', [(5, 0), (8, 1), (19, 1)])
def test_weight_gradients_as_64x64_tiles(V, graph):
    """Option dw64 = N (csrc/mlp_dw.h dw_hot_tile64; off by default -- measured no faster than the 16 x 64 tiles):
    from N batch rows on the hidden matrices' weight gradients are 64 x 64 tiles staged through LDS, every tile summed by up to 8 workgroups over segments of the rows (ticket reduction: partial
    tiles added in segment order).  Against the 16 x 64 tiles: same batches, gradients (m after the first Adam step = 0.1 x
    the summed gradient, util.py:49-53 / mpi_adam.py:31) within 2e-6 of their max-norm -- another order of the sum over the
    rows and nothing else --, transposed copies of the updated matrices kept current (the next update's backward layers
    read them); and a run reproduces itself bit for bit whichever workgroup draws a tile's last ticket."""
    from curious_amd import ops
    from test_gpu_round5 import make_agent, rank_episodes
    outs = []
    for dw64 in (0, 1280, 1280):
        with ops.option('dw64', dw64):
            agent = make_agent(V, use_graph=bool(graph))
            draw = rank_episodes(V, 12)
            agent.store_episode(draw(), np.array([0.3, 0.0, 0.2, 0.1]), 12 * V)
            agent.train()
            torch.cuda.synchronize()
            first = [t.cpu().numpy().copy() for t in (agent._m, agent._losses, agent.grad)]
            agent.train_batches(11)
            agent.update_target_net()
            agent.train_batches(4)
            torch.cuda.synchronize()
            agent.check_faults(wait=True)
            outs.append(first + [agent.theta.cpu().numpy().copy(), agent._m.cpu().numpy().copy()])
    a, b, c = outs
    np.testing.assert_array_equal(a[1], b[1])                        # the losses do not depend on the tiles
    for x, y in ((a[0], b[0]), (a[2], b[2])):
        dev = np.abs(x - y) / np.abs(x).max()
        assert dev.max() <= 2e-6, dev.max()
    assert (np.abs(a[3] - b[3]) > 1e-4).mean() < 1e-2 and np.abs(a[3] - b[3]).max() <= 2e-2
    for x, y in zip(b, c):                                           # bit for bit from run to run
        np.testing.assert_array_equal(x, y)


# ------------------------------------------------------------------ a training job that resumes
def _rows(path):
    import csv
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r.pop('Time', None)                                          # (wall clock)
    return rows


def _same_state(a, b, path='state'):
    """Two checkpoint structures hold the same thing: tensors / arrays bit for bit, containers member for member."""
    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.dtype == b.dtype and torch.equal(a, b), path
    elif isinstance(a, np.ndarray):
        np.testing.assert_array_equal(a, b, err_msg=path)
    elif isinstance(a, dict):
        assert set(a) == set(b), (path, sorted(set(a) ^ set(b)))
        for k in a:
            if k not in ('elapsed',):
                _same_state(a[k], b[k], '%s.%s' % (path, k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same_state(x, y, '%s[%d]' % (path, i))
    elif isinstance(a, float) and np.isnan(a):
        assert isinstance(b, float) and np.isnan(b), path
    elif hasattr(a, '__dict__') and not isinstance(a, type):
        _same_state(vars(a), vars(b), path)
    else:
        assert a == b, (path, a, b)


def _launch(tmp, sub, n_epochs, structure='curious', resume=None, **over):
    from curious_amd.experiment import config, train as tr
    config.CACHED_ENVS.clear()
    root = os.path.join(str(tmp), sub, '')
    params = dict(rng_mode='device', use_graph=True, async_store=True, n_cycles=3, n_batches=6, rollout_batch_size=4,
                  n_test_rollouts=2)
    params.update(over)
    tr.launch(env='MultiTaskFetchArm4-v5', trial_id=0, n_epochs=n_epochs, num_cpu=params.pop('num_cpu', 1), seed=7,
              policy_save_interval=2, clip_return=1, normalize_obs=False, structure=structure,
              task_selection=params.pop('task_selection', 'active_competence_progress'), goal_selection='random',
              goal_replay='her', task_replay=params.pop('task_replay', 'replay_task_cp_buffer'), save_policies=True,
              override_params=params, save_root=root, resume=resume)
    return os.path.join(root, 'MultiTaskFetchArm4-v5', '0')


@pytest.mark.parametrize('case', ['virtual_ranks', 'one_rank', 'task_experts_batched', 'task_experts_sequential',
                                  'task_experts_virtual_ranks'])
def test_training_job_resumes_bit_for_bit(tmp_path, case):
    """experiment.train --resume (what ddpg.py:511-513 says the reference cannot do): 5 epochs in one go
    against 3 epochs + a new job that goes on from the first one's last checkpoint for 2 more.  Checkpoints fall on the
    reference's save cadence (train.py:195-205: policy_save_interval = 2 -> epochs 0, 2, 4) and behind the last epoch.  The
    resumed job ends with progress.csv cell for cell (but the wall clock) and with the SAME training state: parameters,
    targets, Adam moments and counters, normalisers, every replay buffer of every virtual rank, competence queues, task
    probabilities, every RNG stream -- the final checkpoints of both are compared member by member, bit for bit."""
    from curious_amd.checkpoint import STATE_DIR, latest_epoch
    kw = dict(virtual_ranks=dict(num_cpu=3), one_rank=dict(rollout_batch_size=8),
              task_experts_batched=dict(structure='task_experts', experts_update='batched', task_selection='random',
                                        task_replay='replay_current_task_buffer'),
              task_experts_sequential=dict(structure='task_experts', task_selection='active_competence_progress',
                                           task_replay='replay_current_task_buffer'),
              task_experts_virtual_ranks=dict(structure='task_experts', experts_update='batched', task_selection='random',
                                              task_replay='replay_current_task_buffer', num_cpu=2))[case]
    kw = dict(kw)
    structure = kw.pop('structure', 'curious')
    straight = _launch(tmp_path, 'straight', 5, structure, **kw)
    first = _launch(tmp_path, 'resumed', 3, structure, **kw)
    assert latest_epoch(first)['epoch'] == 2
    again = _launch(tmp_path, 'resumed', 5, structure, resume=first, **kw)
    assert os.path.samefile(again, first)
    assert latest_epoch(first)['epoch'] == 4 == latest_epoch(straight)['epoch']
    a, b = _rows(os.path.join(straight, 'progress.csv')), _rows(os.path.join(first, 'progress.csv'))
    assert [r['epoch'] for r in b] == ['-1', '0', '1', '2', '3', '4']
    assert a == b
    sa = torch.load(os.path.join(straight, STATE_DIR, 'rank000_epoch000004.pt'), weights_only=False)
    sb = torch.load(os.path.join(first, STATE_DIR, 'rank000_epoch000004.pt'), weights_only=False)
    assert len(sa['buffers']) >= 4 and sa['policies'][0]['t_Q'] > 0
    if case == 'virtual_ranks':
        assert len(sa['buffers']) == 3 * 5 and len(sa['workers'][0]['vrng']) == 3      # every rank's buffers and streams
        for v in range(3):                                           # (buffer 0 is never written: ddpg.py:178-197)
            assert sum(int(bf['current_size']) > 0 for bf in sa['buffers'] if bf['slot'] // 5 == v) >= 2, v
    _same_state(sa, sb)
    # only the files of the last checkpoint are kept, and the reference-format policy files are there as before
    assert sorted(os.listdir(os.path.join(first, STATE_DIR))) == ['LATEST.json', 'rank000_epoch000004.pt']
    assert os.path.exists(os.path.join(first, 'policy_latest.pkl'))


def test_resume_refuses_another_configuration(tmp_path):
    """--resume checks the job against the params.json of the directory it resumes: another seed / layout is an error, not a
    silently different job."""
    first = _launch(tmp_path, 'a', 1)
    with pytest.raises(ValueError, match='configured differently'):
        _launch(tmp_path, 'a', 2, resume=first, rollout_batch_size=6)
    with pytest.raises(FileNotFoundError):
        _launch(tmp_path, 'b', 2, resume=os.path.join(str(tmp_path), 'nowhere'))


def test_training_job_resumes_over_two_processes(tmp_path):
    """The same over two processes (gloo, both on this GPU) with an uneven virtual layout (--num_cpu 3 = ranks {0, 1} + {2}):
    every process writes and reads its own file, rank 0 publishes the checkpoint behind a barrier."""
    import subprocess
    import sys
    from curious_amd.checkpoint import STATE_DIR
    from test_gpu_round4 import _free_port, _two_rank_env
    env = _two_rank_env()
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')

    def job(cwd, n_epochs, resume=None):
        os.makedirs(cwd, exist_ok=True)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(_free_port()), '-m', 'curious_amd.experiment.train', '--env',
               'MultiTaskFetchArm4-v5', '--num_cpu', '3', '--rollout_batch_size', '4', '--n_batches', '6', '--n_epochs',
               str(n_epochs), '--n_cycles', '3', '--seed', '1', '--policy_save_interval', '2']
        if resume:
            cmd += ['--resume', resume]
        out = subprocess.run(cmd, env=env, cwd=cwd, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        return os.path.join(cwd, 'save', 'MultiTaskFetchArm4-v5', '0')
    straight = job(os.path.join(str(tmp_path), 'straight'), 4)
    first = job(os.path.join(str(tmp_path), 'resumed'), 2)
    job(os.path.join(str(tmp_path), 'resumed'), 4, resume=first)
    assert _rows(os.path.join(straight, 'progress.csv')) == _rows(os.path.join(first, 'progress.csv'))
    for r in (0, 1):
        sa = torch.load(os.path.join(straight, STATE_DIR, 'rank%03d_epoch000003.pt' % r), weights_only=False)
        sb = torch.load(os.path.join(first, STATE_DIR, 'rank%03d_epoch000003.pt' % r), weights_only=False)
        assert sa['layout']['virtual_ranks'] == (2 if r == 0 else 1) and sa['layout']['total_ranks'] == 3
        _same_state(sa, sb)


# ------------------------------------------------------------------ task experts with virtual ranks
def _expert_kit_ranks(V, nb=4, dimo=40, batch_size=256, cap_eps=64):
    """tests/test_gpu_round2._expert_kit with V private buffer sets (config.py:210-214 runs once per MPI rank) on one pool."""
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    from test_gpu_agent import T, tables
    G = 3 * nb
    ag_ids, g_ids = tables(nb)
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    tr = 'replay_current_task_buffer'
    sampler = make_sample_multi_task_her_transitions('her', 4, tr, sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)),
                                                     tasks_ag_id=ag_ids, tasks_g_id=g_ids)
    bufs = make_pooled_buffers(shapes, T * cap_eps, T, sampler, nb + 1, alias_from=5, n_ranks=V)
    gamma = 1. - 1. / T

    def make(t_id, use_graph=False, **hooks):
        from curious_amd.ddpg import DDPG
        return DDPG(input_dims=dims, hidden=256, layers=3, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                    polyak=0.95, batch_size=batch_size, Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1.,
                    action_l2=1., clip_obs=200., scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None,
                    relative_goals=False, clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False,
                    sample_transitions=sampler, gamma=gamma, buffers=bufs, tasks_ag_id=ag_ids, tasks_g_id=g_ids,
                    task_replay=tr, eps_task=0.4, structure='task_experts', t_id=t_id, seed=10 + t_id, rng_mode='device',
                    use_graph=use_graph, virtual_ranks=V, **hooks)
    return make, bufs


@pytest.mark.parametrize('V,graph', [(2, 0), (3, 1)])
def test_task_experts_with_virtual_ranks(V, graph):
    """structure='task_experts' on V virtual ranks (train.py:65-121 runs its expert loop on every MPI rank alike; the experts'
    gradients are summed over the ranks by their MpiAdam, mpi_adam.py:26): every rank owns a buffer set shared by ITS experts,
    expert t draws 256 transitions per rank from that rank's buffer t + 1 relabelled to its own task (ddpg.py:302-318,335),
    an update consumes the V minibatches in one launch sequence.  (i) the bank (all experts in one launch sequence) == the
    experts updated one by one, bit for bit; (ii) every rank's rows come from ITS episodes and carry the expert's task;
    (iii) per-rank losses and the Adam step from the SUMMED oracle gradients against the float64 oracle, as for the curious
    agent in test_virtual_ranks_match_the_oracle_rank_model."""
    from curious_amd import ops
    from curious_amd.experts import ExpertBank
    from oracle.optim import adam_update
    from test_gpu_round4 import _oracle_agent
    from test_gpu_round5 import B, DIMO, NB, STAGE_KEYS, rank_episodes
    groups = []
    for batched in (True, False):
        make, bufs = _expert_kit_ranks(V)
        if batched:
            bank = ExpertBank(lambda t, **h: make(t, use_graph=bool(graph), **h), NB)
            xs = list(bank)
        else:
            bank, xs = None, [make(t, use_graph=bool(graph)) for t in range(NB)]
        xs[0].store_episode(rank_episodes(V, 24)(), np.zeros(NB), 24 * V)
        groups.append((bank, xs))
    (bank, bx), (_, sx) = groups
    assert all(b.current_size > 0 for bl in bx[0]._rank_buffers for b in bl[1:NB + 1])
    # ---- one update recorded for the oracle, then a run of updates
    pre = [[ops.unpad_params(x.net_cfg, t.cpu().numpy()) for t in (x.theta, x._m, x._v)] for x in bx]
    p = bank._cur
    bank.train_batches(1)
    for x in sx:
        x.train_batches(1)
    torch.cuda.synchronize()
    a = _oracle_agent(10)                                            # (network shapes only: the parameters come from the product)
    PQ = a.math.P_Q
    for e, x in enumerate(bx):
        views = x._layout.batch_views(x._pp[p])
        batch = {k: views[k].cpu().numpy().copy() for k in STAGE_KEYS}
        # (the HER samples -- future_p = 0.8 of them -- are relabelled to the expert's task, the others keep theirs: her.py:131-155)
        assert (batch['task_descr'].argmax(axis=1) == e).mean() > 0.7 and np.all(batch['task_descr'].sum(axis=1) == 1)
        assert all(x.proportions[i] == (B if i == e + 1 else 0) for i in range(NB + 1))
        assert not np.array_equal(batch['o'][:B], batch['o'][B:2 * B])
        th, m, v = pre[e]
        outs = [a.math.losses_and_grads(th, th, {k: batch[k][r * B:(r + 1) * B] for k in STAGE_KEYS}) for r in range(V)]
        losses = x._losses.cpu().numpy().reshape(V, 2)
        for r in range(V):
            assert abs(losses[r, 0] - float(outs[r]['Q_loss'])) <= 1e-5 * abs(float(outs[r]['Q_loss'])), (e, r)
            assert abs(losses[r, 1] - float(outs[r]['pi_loss'])) <= 1e-5 * abs(float(outs[r]['pi_loss'])) + 1e-7, (e, r)
        got_m = ops.unpad_params(x.net_cfg, x._m.cpu().numpy())
        for sl, key, lr in ((slice(0, PQ), 'Q_grad', 1e-3), (slice(PQ, None), 'pi_grad', 1e-3)):
            g = sum(o[key] for o in outs)                            # mpi_adam.py:26: SUM over ranks
            want_m = adam_update(th[sl], m[sl], v[sl], 0, g, lr)[1]
            dev = np.abs(got_m[sl] - want_m) / np.abs(want_m).max()
            assert dev.max() <= 5e-4 and (dev > 2e-5).mean() <= 2e-3, (e, key, dev.max(), (dev > 2e-5).mean())
    bank.train_batches(11)
    for x in sx:
        x.train_batches(11)
    bank.update_target_net()
    for x in sx:
        x.update_target_net()
    torch.cuda.synchronize()
    assert bank.batched
    for x, y in zip(bx, sx):
        assert x.Q_adam.t == y.Q_adam.t == 12
        assert torch.equal(x.theta, y.theta) and torch.equal(x._m, y._m) and torch.equal(x._v, y._v)
        assert torch.equal(x.theta_target, y.theta_target) and torch.equal(x._staged, y._staged)
        assert torch.equal(x._losses, y._losses)
    bank.check_faults()


def test_training_job_of_task_experts_with_virtual_ranks(tmp_path):
    """experiment.train --structure task_experts --experts_update batched --num_cpu 3 on one process: the whole loop (expert of
    the epoch collects on every rank, every expert is updated from every rank's buffers, the evaluator acts with one expert
    per task on the envs of all ranks); the episode count is the reference's, and a job that asks for ranks it cannot have
    is refused instead of running another job."""
    first = _launch(tmp_path, 'x', 3, 'task_experts', num_cpu=3, experts_update='batched', task_selection='random',
                    task_replay='replay_current_task_buffer')
    rows = _rows(os.path.join(first, 'progress.csv'))
    assert [r['epoch'] for r in rows] == ['-1', '0', '1', '2']
    assert int(float(rows[-1]['train/episode'])) == 3 * 4 * 3            # cycles x rollouts per rank x ranks, of the LAST expert's worker
    assert all(np.isfinite(float(r['test/mean_Q'])) for r in rows)
    with pytest.raises(ValueError, match='virtual ranks'):
        _launch(tmp_path, 'y', 1, 'task_experts', num_cpu=3, task_selection='random',
                task_replay='replay_current_task_buffer')               # (sequential experts)
    with pytest.raises(ValueError, match='virtual ranks'):
        _launch(tmp_path, 'z', 1, 'curious', num_cpu=3, rng_mode='numpy', use_graph=False, async_store=False)


# ------------------------------------------------------------------ bench.py on the published rank layout
def test_bench_lays_the_published_ranks_out_over_the_processes():
    """`bench.py --gpus 2 --num-cpu 3` (torch.distributed.run, gloo: both processes share this box's GPU): the reference's
    --num_cpu R laid out like experiment.train lays it out -- ranks {0, 1} on process 0, {2} on process 1
    (dist.virtual_layout; `--gpus 8 --num-cpu 19` = 3 3 3 2 2 2 2 2 is the published job, readme.md:16).  `value` counts
    R x 256 transitions per update -- the accounting of the one-process run `--virtual-ranks 3` --, the line names the
    layout, the replicas end identical."""
    import json
    from test_gpu_round4 import _launch2, _two_rank_env
    out = _launch2([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--num-cpu', '3', '--steps', '3', '--warmup', '1',
                    '--prefill', '256', '--no-ipc-probe'], _two_rank_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-1500:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['ranks'] == 3 and rec['config']['ranks_per_process'] == [2, 1]
    assert rec['config']['rollout_batch_size'] == 2 and '--num_cpu 3' in rec['config']['workload']
    per_s = 3 / (rec['ms_per_step'] * 3e-3)                          # cycles per second
    assert abs(rec['value'] - 3 * 100 * 256 * per_s) < 1e-3 * rec['value']          # R x 256 per update, 100 updates per cycle
    assert abs(rec['env_steps_per_sec'] - 3 * 2 * 50 * per_s) < 1e-3 * rec['env_steps_per_sec']
    c = rec['collectives']
    assert c['rccl']['world'] == 2 and c['replicas_identical'] is True


# ------------------------------------------------------------------ evaluation rollouts, several to a launch
@pytest.mark.parametrize('V,B,nb', [(1, 256, 4), (1, 16, 4), (3, 2, 4), (1, 64, 8)])
def test_evaluation_rollouts_several_to_a_launch(V, B, nb):
    """RolloutWorker.generate_eval_rollouts: the n_test_rollouts evaluation rollouts of an epoch (train.py:156-158) as a few
    launches of up to 1 024 env SLOTS (rollout k of env e = slot k' x n_envs + e, the same env id at the episode that env
    would be at) against one launch per rollout: the same draws, the same episodes (success flags, tasks, competence queues,
    counters: identical), test/mean_Q equal to 1e-6 (a mean taken in another order); twice in a row (the episodes of the
    second evaluation follow the first's), and with training rollouts in between (they advance nothing of the evaluator's).
    Arm8: the distractor tasks' random walk is keyed by (env id, episode) as well (curious_env_cfg_t.wrap)."""
    from curious_amd import logger
    from curious_amd.envs import EnvFactory
    from curious_amd.rollout import RolloutWorker
    from test_gpu_round5 import make_agent
    from test_gpu_agent import T
    name = 'MultiTaskFetchArm%d-v5' % nb
    outs = []
    for slots in (0, 1024):
        if nb == 4:
            agent = make_agent(V if V > 1 else None, use_graph=True, rollout_batch_size=B)
            dims = dict(o=40, u=4, g=12, ag=12, task_descr=4, info_is_success=1)
        else:
            from test_gpu_agent import build_pair
            agent, _ = build_pair(8, 52, rng_mode='device', use_graph=True)
            dims = dict(o=52, u=4, g=24, ag=24, task_descr=8, info_is_success=1)
        ev = RolloutWorker(EnvFactory(name), agent, dims, logger, T=T, rollout_batch_size=B, exploit=True, compute_Q=True,
                           structure='curious', task_selection='active_competence_progress', queue_length=20, eval=True)
        ev.EVAL_SLOTS = slots
        ev.seed(21)
        if V > 1:
            ev.seed_ranks([31 + v for v in range(V)])
        np.random.seed(5)
        rec = []
        for n in (10, 5):
            ev.generate_eval_rollouts(n)
            torch.cuda.synchronize()
            rec.append((list(ev.success_history)[-n:], list(ev.Q_history)[-n:], list(ev.task_history), ev.n_episodes, ev.count,
                        [(list(q.successes), q.C, q.CP) for q in ev.competence_computers],
                        ev.benv.episode.cpu().numpy().copy()))
        assert (slots > 0) == ('_eval_env' in ev.__dict__)
        outs.append(rec)
    for a, b in zip(*outs):
        assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[5] == b[5]
        np.testing.assert_array_equal(a[6], b[6])
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-7)
        assert 0.0 <= min(a[0]) and np.isfinite(a[1]).all()


@pytest.mark.parametrize('case', range(6))
def test_big_policy_forward_sixteen_rows_vs_oracle(case):
    """curious_policy_forward on >= 1 024 rows with option fwd16 takes 16 rows per workgroup (mlp_rows_act.h policy_fwd16_kernel: the
    evaluator's Q pass over the recorded rows of its rollouts, DDPG.rollout_q_sum; ddpg.py:129-147): pi and Q(pi) within
    1e-5 of the float64 oracle -- 2-3 hidden layers, 1-8 tasks, normalised inputs, relative goals, with and without Q --
    and within 1e-5 of the 4-row form (another order of the sums over k, not another result)."""
    from curious_amd import ops
    from oracle.ddpg import preprocess_og
    from oracle.networks import DDPGMath
    from test_gpu_kernels import dev
    rs = np.random.RandomState(4100 + case)
    nb, dimo = int(rs.randint(1, 9)), int(rs.randint(5, 60))
    G = 3 * nb
    n = int(rs.choice([1024, 2048, 13056, 4096 + 16]))
    layers = int(rs.randint(2, 4))
    max_u = float(rs.choice([1.0, 2.0]))
    clip = float(rs.choice([200.0, 2.0]))
    m64 = DDPGMath(dimo, G, 4, nb, 256, layers, max_u, 0.98, 50., True, 1.0, True, np.float64)
    m32 = DDPGMath(dimo, G, 4, nb, 256, layers, max_u, 0.98, 50., True, 1.0, True, np.float32)
    theta = m32.init(rs)
    norm, relative, with_q = case % 3 == 1, case % 3 == 2, case != 4
    ncfg = ops.make_net_cfg(dimo, G, 4, nb, 256, layers, True, max_u, 0.98, 50., 1.0, normalize_obs=norm, norm_clip=5.0)
    o = (rs.randn(n, dimo) * 3).astype(np.float32)
    g = rs.randn(n, G).astype(np.float32)
    ag = rs.randn(n, G).astype(np.float32)
    td = np.eye(nb, dtype=np.float32)[rs.randint(nb, size=n)]
    stats = {}
    if norm:
        for key, d in (('o', dimo), ('g', G)):
            mean, std = (rs.randn(d) * 0.3).astype(np.float32), (0.5 + rs.rand(d)).astype(np.float32)
            st = np.zeros(4 * d + 1, np.float32)
            st[2 * d] = 1
            st[2 * d + 1:3 * d + 1] = mean
            st[3 * d + 1:] = std
            stats[key] = (mean, std, dev(st))
    ws = torch.zeros(ops.workspace_floats(ncfg, n), device='cuda')
    th, do, dg, dtd, dag = dev(ops.pad_params(ncfg, theta)), dev(o), dev(g), dev(td), dev(ag)

    def run():
        pi = torch.full([n, 4], float('nan'), device='cuda')
        Q = torch.full([n, 1], float('nan'), device='cuda') if with_q else None
        ops.policy_forward(ncfg, th, do, dg, dtd, n, clip, ws, pi, Q, ag=dag, relative_goals=relative,
                           o_stats=stats['o'][2] if norm else None, g_stats=stats['g'][2] if norm else None)
        torch.cuda.synchronize()
        return pi.cpu().numpy(), (Q.cpu().numpy() if with_q else None)
    ops.prof_collect()
    ops.prof_enable(True)
    with ops.option('fwd16', 1):
        pi16, q16 = run()
    ops.prof_enable(False)
    assert ops.prof_collect().get('policy_rows_kernel', (0, 0))[0] == 1          # (one launch: both forms report under this name)
    pi4, q4 = run()                                                  # (the default: the 4-row form of the fused acting kernels)
    with ops.option('fwd16', 1), ops.option('rows16', 0):
        pi4b, _ = run()
    np.testing.assert_array_equal(pi4, pi4b)
    oc, gc = preprocess_og(o, ag, g, clip, relative)
    if norm:
        oc = np.clip((oc.astype(np.float32) - stats['o'][0]) / stats['o'][1], -5, 5)
        gc = np.clip((gc.astype(np.float32) - stats['g'][0]) / stats['g'][1], -5, 5)
    Qp, pip = m64.split(theta.astype(np.float64))
    want_pi, _, _ = m64.actor(pip, oc.astype(np.float64), td.astype(np.float64), gc.astype(np.float64))
    tag = 'case %d: n %d nb %d dimo %d layers %d' % (case, n, nb, dimo, layers)
    np.testing.assert_allclose(pi16, want_pi, rtol=1e-5, atol=2e-6 * max_u, err_msg=tag)
    np.testing.assert_allclose(pi16, pi4, rtol=1e-5, atol=2e-6 * max_u, err_msg=tag)
    assert not np.array_equal(pi16, pi4) or layers < 2, 'the 16-row form did not run'
    if with_q:
        want_Q, _ = m64.critic(Qp, oc.astype(np.float64), td.astype(np.float64), gc.astype(np.float64), want_pi / max_u)
        np.testing.assert_allclose(q16, want_Q, rtol=1e-5, atol=4e-6, err_msg=tag)
        np.testing.assert_allclose(q16, q4, rtol=1e-5, atol=4e-6, err_msg=tag)


def test_several_rank_paths_of_eight_virtual_ranks_leave_the_same_bits():
    """Eight virtual ranks (2 048 rows: the 16-row kernels, the small weight-gradient problems dealt to the XCDs in halves
    and reduced in 3 segments, mlp_dw.h dw_role) through the fused update (gradients + Adam + gather in dw_adam_her_kernel)
    and through the several-process path on a one-rank RCCL communicator (dw_all_kernel, all-reduce, adam_her_kernel; eager
    and captured): the same parameters bit for bit after 35 updates -- and other bits than with the problems kept whole
    (the balanced map did apply), within the rounding of another order of the partial sums."""
    import socket
    import subprocess
    import sys
    digests = {}
    for name, extra in (('fused', {}), ('eager', {'CURIOUS_FORCE_DIST': '1', 'CURIOUS_GRAPH_ALLREDUCE': '0'}),
                        ('captured', {'CURIOUS_FORCE_DIST': '1', 'CURIOUS_GRAPH_ALLREDUCE': '1'}),
                        ('whole', {'CURIOUS_DW_BAL': '0'})):
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   CURIOUS_RANK_CHECK_V='8')
        env.pop('CURIOUS_GRAPH_ALLREDUCE', None)
        env.update(extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rank_path_check.py')], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('DIGEST')][-1].split()
        assert line[2] == '35'
        digests[name] = (line[1], float(line[3]))
    assert digests['fused'][0] == digests['eager'][0] == digests['captured'][0], digests
    assert digests['whole'][0] != digests['fused'][0]
    assert abs(digests['whole'][1] - digests['fused'][1]) <= 1e-3 * abs(digests['fused'][1]), digests


def test_bench_default_line_carries_the_reference_regime():
    """bench.regime_line: the diagnostic the default `python bench.py` adds to its JSON line -- the reference's published
    `--num_cpu 19` job (readme.md:16) as 19 virtual ranks on this GPU, measured by a child process (here: 2 cycles)."""
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    r = bench.regime_line(types.SimpleNamespace(no_graph=False), steps=2)
    assert 'error' not in r, r
    assert r['ranks'] == 19 and r['unit'] == 'transitions/s' and r['value'] > 1e7 and r['ms_per_step'] > 0
    assert r['roofline']['bound'] == 'mfma' and 0.3 < r['roofline']['frac'] < 1.0
    assert set(r['kernels_avg_us']) >= {'ddpg_rows_kernel', 'dw_adam_her_kernel'}
