"""Training-resumable checkpoints.

The reference can only pickle a policy for acting: Adam moments (mpi_adam.py:14-16), replay buffers and competence
queues are not saved (docstring at ddpg.py:512), so a run cannot be resumed.  `save_training_state` writes everything a
bit-exact continuation needs -- parameters, target parameters, Adam m / v / t, the device step counter, normaliser
state and accumulators, every replay buffer's stored episodes and counters, competence queues / task probabilities
of the rollout workers, and the NumPy global RNG state -- next to the reference-format `*_weights.pkl`
(DDPG.save_weights, ddpg.py:481-497), which stays loadable by the reference.
"""
import numpy as np
import torch


def _buffers_of(policy):
    bufs = policy.buffer if isinstance(policy.buffer, list) else [policy.buffer]
    seen, out = set(), []
    for i, b in enumerate(bufs):
        if b is not None and id(b) not in seen:
            seen.add(id(b))
            out.append((i, b))
    return out


def policy_state(policy):
    st = dict(
        theta=policy.theta.cpu(), theta_target=policy.theta_target.cpu(), m=policy._m.cpu(), v=policy._v.cpu(),
        t_Q=policy.Q_adam.t, t_pi=policy.pi_adam.t, step_ctr=int(policy._step_ctr),
        o_stats=policy.o_stats.state.cpu(), g_stats=policy.g_stats.state.cpu(), stats_acc=policy._stats_acc.cpu(),
        cp=None if policy.cp is None else np.asarray(policy.cp, dtype=np.float64).copy(),
        noise_counter=policy._noise_counter, stats_calls=getattr(policy, '_stats_calls', 0),
        store_calls=getattr(policy, '_store_calls', 0),
        buffers=[])
    for i, b in _buffers_of(policy):
        st['buffers'].append(dict(index=i, current_size=b.current_size, n_transitions_stored=b.n_transitions_stored,
                                  records=b.records[:b.current_size].cpu()))
    return st


def load_policy_state(policy, st):
    policy.theta.copy_(st['theta'])
    policy.theta_target.copy_(st['theta_target'])
    policy._m.copy_(st['m'])
    policy._v.copy_(st['v'])
    policy.Q_adam.t, policy.pi_adam.t = st['t_Q'], st['t_pi']
    policy._step_ctr.fill_(st['step_ctr'])
    policy.o_stats.state.copy_(st['o_stats'])
    policy.g_stats.state.copy_(st['g_stats'])
    policy._stats_acc.copy_(st['stats_acc'])
    policy.cp = st['cp']
    policy._noise_counter = st['noise_counter']               # the device mirror follows lazily (DDPG.act_rollout)
    if getattr(policy, '_noise_base', None) is not None:
        policy._noise_base_val = None
    policy._stats_calls = st['stats_calls']
    policy._store_calls = st.get('store_calls', 0)             # index of the Philox slot draws (device RNG mode)
    by_index = {i: b for i, b in _buffers_of(policy)}
    for bs in st['buffers']:
        b = by_index[bs['index']]
        b.current_size, b.n_transitions_stored = bs['current_size'], bs['n_transitions_stored']
        b.records[:b.current_size].copy_(bs['records'])
    # everything derived from the counters is rebuilt lazily
    policy._tables_dirty = True
    policy._batch_stale = True
    policy._alpha_filled = 0


def worker_state(w):
    st = dict(n_episodes=w.n_episodes, C=np.asarray(w.C).copy(), CP=np.asarray(w.CP).copy(),
              success_history=list(w.success_history), reward_history=list(w.reward_history),
              Q_history=list(w.Q_history), count=w.count)
    if hasattr(w, 'competence_computers'):
        st['p'] = np.asarray(w.p).copy()
        st['queues'] = [(list(q.successes), q.C, q.CP) for q in w.competence_computers]
        st['task_history'] = list(w.task_history)
    if getattr(w, 'batched', False):
        st['env_episode'] = w.benv.episode.cpu()
    return st


def load_worker_state(w, st):
    w.n_episodes, w.C, w.CP, w.count = st['n_episodes'], st['C'], st['CP'], st['count']
    for name in ('success_history', 'reward_history', 'Q_history'):
        h = getattr(w, name)
        h.clear()
        h.extend(st[name])
    if 'queues' in st:
        w.p = st['p']
        for q, (succ, C, CP) in zip(w.competence_computers, st['queues']):
            q.successes.clear()
            q.successes.extend(succ)
            q.C, q.CP = C, CP
        w.task_history.clear()
        w.task_history.extend(st['task_history'])
    if 'env_episode' in st:
        w.benv.episode.copy_(st['env_episode'])


def save_training_state(path, policy, workers=()):
    for w in workers:                                                # async_store: flags / routing the host has not
        if hasattr(w, 'settle'):                                     # mirrored yet
            w.settle()
    if hasattr(policy, 'settle'):
        policy.settle()
    torch.cuda.synchronize()
    torch.save(dict(policy=policy_state(policy), workers=[worker_state(w) for w in workers],
                    numpy_rng=np.random.get_state()), path)


def load_training_state(path, policy, workers=()):
    ck = torch.load(path, weights_only=False)
    load_policy_state(policy, ck['policy'])
    for w, st in zip(workers, ck['workers']):
        load_worker_state(w, st)
    np.random.set_state(ck['numpy_rng'])
    torch.cuda.synchronize()
