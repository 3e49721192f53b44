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
    """Option rows16 = N: batches of >= N rows take the 16-row form (default 2 048 rows = 8 ranks; here forced from 1 row
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


@pytest.mark.parametrize('case', ['virtual_ranks', 'one_rank', 'task_experts_batched', 'task_experts_sequential'])
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
                                           task_replay='replay_current_task_buffer'))[case]
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
