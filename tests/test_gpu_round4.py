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
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-1500:]                        # rank 0 prints ONE line
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
