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
