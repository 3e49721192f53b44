"""One rank of the world-2 parity tests (tests/test_gpu_round4.py): the product's several-rank update path -- HIP
gradients, the gradient all-reduce (SUM over ranks, mpi_adam.py:26-28, ddpg.py:452-453), the stand-alone optimiser, and
the normaliser all-reduce (MEAN over ranks, normalizer.py:84-94) -- on private data per rank.  Two such processes share
ONE GPU with gloo carrying the collectives (RCCL refuses two ranks on one device).  Every batch the device drew, every
loss and the final state are written to <out>.rank<r>.npz; the test replays the same batches through the oracle's
two-rank model.  The worker itself never touches oracle/.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/rank_parity_worker.py <out> <single|experts> <graph 0|1>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

NB, DIMO, SEED, B, CAP = 4, 40, 3, 256, 64
STAGE_KEYS = ['ag', 'g', 'o', 'task_descr', 'u', 'o_2', 'g_2', 'r']


def make_agent(use_graph, t_id=None, buffers=None, sampler=None, **hooks):
    from curious_amd.ddpg import DDPG
    from curious_amd.envs import sparse_reward_fun
    from curious_amd.her import make_sample_multi_task_her_transitions
    from curious_amd.replay_buffer import make_pooled_buffers
    from test_gpu_agent import T, tables
    G = 3 * NB
    ag_ids, g_ids = tables(NB)
    dims = dict(o=DIMO, u=4, g=G, ag=G, task_descr=NB, info_is_success=1)
    shapes = dict(o=(T + 1, DIMO), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, NB),
                  change=(T, G))
    experts = t_id is not None
    if sampler is None:
        sampler = make_sample_multi_task_her_transitions(
            'her', 4, 'replay_current_task_buffer' if experts else 'replay_task_cp_buffer',
            sparse_reward_fun(dict(kind='sparse_l2', eps=0.05)), tasks_ag_id=ag_ids, tasks_g_id=g_ids)
        buffers = make_pooled_buffers(shapes, T * CAP, T, sampler, NB + 1, alias_from=5)
    gamma = 1. - 1. / T
    agent = DDPG(input_dims=dims, hidden=256, layers=3, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                 polyak=0.95, batch_size=B, Q_lr=1e-3, pi_lr=1e-3, norm_eps=0.01, norm_clip=5, max_u=1., action_l2=1.,
                 clip_obs=200., scope='ddpg', T=T, rollout_batch_size=2, subtract_goals=None, relative_goals=False,
                 clip_pos_returns=True, clip_return=1. / (1. - gamma), normalize_obs=False, sample_transitions=sampler,
                 gamma=gamma, buffers=buffers, tasks_ag_id=ag_ids, tasks_g_id=g_ids,
                 task_replay='replay_current_task_buffer' if experts else 'replay_task_cp_buffer', eps_task=0.4,
                 structure='task_experts' if experts else 'curious', t_id=t_id, rng_mode='device',
                 seed=SEED + (t_id or 0), use_graph=use_graph, **hooks)
    return agent, buffers, sampler


def batch_arrays(agent, packed):
    views = agent._layout.batch_views(packed)
    return {k: views[k].cpu().numpy().copy() for k in STAGE_KEYS}


def stats_rows(agent):
    """The rows the normalisers of the last store_episode were fed with (the device drew them: ddpg.py:207-223)."""
    cols = agent._layout.batch_cols
    sb = agent._stats_batch
    return (sb[:, cols['o'][0]:cols['o'][0] + agent.dimo].cpu().numpy().copy(),
            sb[:, cols['g'][0]:cols['g'][0] + agent.dimg].cpu().numpy().copy())


def main():
    out, mode, graph = sys.argv[1], sys.argv[2], sys.argv[3] == '1'
    from curious_amd import dist
    from test_gpu_agent import synth_episodes
    dist.init_from_env()
    torch.cuda.set_device(dist.local_device_index())
    rank = dist.rank()
    assert dist.world_size() == 2 and dist.is_distributed()
    rec = {}
    rng = np.random.RandomState(50 + rank)                           # rank-private data (train.py:242-243)
    np.random.seed(900 + rank)
    cp = np.array([0.3, 0.0, 0.2, 0.1])

    if mode == 'fault':
        return fault_main(out, rank, rng, cp)
    if mode == 'uneven':
        # 3 of the reference's ranks on these 2 processes (dist.virtual_layout): the sequence of test_gpu_round5.run_virtual
        import test_gpu_round5 as R5
        V, base, total = dist.virtual_layout(3)
        assert (V, base, total) == ((2, 0, 3) if rank == 0 else (1, 2, 3))
        agent5, rec5 = R5.run_virtual(V, graph, rank_base=base, total_ranks=total)
        agent5._check_synced(wait=True)                              # mpi_adam.py:42-50
        np.savez('%s.rank%d.npz' % (out, rank), **{k: v for k, v in rec5.items() if isinstance(v, np.ndarray)})
        from curious_amd.experiment.train import shutdown
        return shutdown([agent5])
    if mode == 'single':
        agent, _, _ = make_agent(graph)
        agents, bank = [agent], None
    else:
        from curious_amd.experts import ExpertBank
        shared = {}

        def mk(i, **hooks):
            a, shared['b'], shared['s'] = make_agent(graph, t_id=i, buffers=shared.get('b'), sampler=shared.get('s'),
                                                     **hooks)
            return a
        bank = ExpertBank(mk, NB)
        agents = list(bank)

    def store(tag):
        ep = synth_episodes(rng, 24, NB, DIMO)
        agents[0].store_episode({k: v.copy() for k, v in ep.items()}, cp, 24)
        torch.cuda.synchronize()
        rec['stats_o_' + tag], rec['stats_g_' + tag] = stats_rows(agents[0])
        st = agents[0]
        rec['o_state_' + tag] = st.o_stats.state.cpu().numpy().copy()
        rec['g_state_' + tag] = st.g_stats.state.cpu().numpy().copy()

    def state(k):
        from curious_amd import ops
        for e, a in enumerate(agents):
            for name, vec in (('theta', a.theta), ('m', a._m), ('v', a._v)):
                rec['%s_pre_%d_%d' % (name, k, e)] = ops.unpad_params(a.net_cfg, vec.cpu().numpy())

    def update(k):
        state(k)                                                     # parameters and moments this update starts from
        if bank is None:
            p = agents[0]._cur
            cl, qpi = agents[0].train()
            outs = [(cl, qpi)]
        else:
            p = bank._cur
            outs = bank.train()
        torch.cuda.synchronize()
        for e, a in enumerate(agents):
            for key, v in batch_arrays(a, a._pp[p]).items():
                rec['batch_%d_%d_%s' % (k, e, key)] = v
            rec['loss_%d_%d' % (k, e)] = np.float32(float(outs[e][0]))
            rec['qpi_%d_%d' % (k, e)] = outs[e][1].cpu().numpy().copy()

    store('a')
    k = 0
    for _ in range(6):
        update(k)
        k += 1
    store('b')                                                       # a second store: the running sums accumulate
    for _ in range(2):
        update(k)
        k += 1
    for a in agents:
        a.update_target_net()
    torch.cuda.synchronize()
    from curious_amd import ops
    for e, a in enumerate(agents):
        a._check_synced(wait=True)                                   # mpi_adam.py:42-50
        rec['theta_%d' % e] = ops.unpad_params(a.net_cfg, a.theta.cpu().numpy())
        rec['target_%d' % e] = ops.unpad_params(a.net_cfg, a.theta_target.cpu().numpy())
        rec['m_%d' % e] = ops.unpad_params(a.net_cfg, a._m.cpu().numpy())
        rec['v_%d' % e] = ops.unpad_params(a.net_cfg, a._v.cpu().numpy())
    rec['n_updates'] = np.int64(k)
    np.savez(out + '.rank%d.npz' % rank, **rec)
    from curious_amd.experiment.train import shutdown
    shutdown(agents, bank)


def fault_main(out, rank, rng, cp):
    """A hand-off fault on ONE rank (rank 1 loses a producer of Q' in one update): the flag element of the gradient
    all-reduce freezes BOTH ranks from that update on, both raise HandoffFault at the same cycle count, both resume."""
    from curious_amd import ops
    from curious_amd.ddpg import FAULT_CHECK_EVERY, HandoffFault
    from test_gpu_agent import synth_episodes
    agent, _, _ = make_agent(False)
    agent.store_episode({k: v.copy() for k, v in synth_episodes(rng, 24, NB, DIMO).items()}, cp, 24)
    rec = {}

    def snap(tag):
        torch.cuda.synchronize()
        rec['theta_' + tag] = agent.theta.cpu().numpy().copy()
        rec['m_' + tag] = agent._m.cpu().numpy().copy()
        rec['fault_' + tag] = np.int64(int(ops.fault_word(agent.net_cfg, agent.batch_size, agent._workspace)))
    for _ in range(3):
        agent.train()
    snap('good')
    if rank == 1:
        with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
            agent.train()
            torch.cuda.synchronize()
    else:
        agent.train()
    snap('hit')                                                      # the faulted update: skipped on BOTH ranks
    agent.train_batches(2)
    snap('frozen')                                                   # sticky on both ranks
    raised = []
    for tick in range(1, 2 * FAULT_CHECK_EVERY + 1):
        try:
            agent.update_target_net()
        except HandoffFault:
            raised.append(tick)
    rec['raised'] = np.array(raised, np.int64)
    snap('cleared')
    agent.train_batches(3)
    snap('resumed')
    agent._check_synced(wait=True)
    agent.check_faults(wait=True)
    np.savez(out + '.rank%d.npz' % rank, **rec)
    from curious_amd.experiment.train import shutdown
    shutdown([agent], None)


if __name__ == '__main__':
    main()
