"""HER transition samplers (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates baselines/her/her.py of the reference:
  * make_sample_her_transitions            her.py:5-68   (flat structure)
  * make_sample_multi_task_her_transitions her.py:72-185 (curious / task_experts)

The restatement is split in two so that the HIP kernel can be checked stage by stage:
  draw_*   -- consumes the NumPy legacy RNG stream in exactly the reference's order
  apply_*  -- deterministic index math, gather, goal/task relabel, reward
Unlike the reference, apply_* is vectorised (no per-sample Python loop); equality with
the imported reference is pinned by tests/golden/her_*.npz.
"""
import numpy as np


def future_p_of(goal_replay, her_replay_k):
    # her.py:15-18 / her.py:86-89
    return 1 - (1. / (1 + her_replay_k)) if goal_replay == 'her' else 0


def draw_her(rng, n_episodes, T, batch_size):
    """RNG consumption of her.py:108-117 (same for her.py:26-36).

    Returns episode_idxs[int64 B], t_samples[int64 B], u_her[f64 B], u_off[f64 B]."""
    episode_idxs = rng.randint(0, n_episodes, batch_size)          # her.py:108
    t_samples = rng.randint(T, size=batch_size)                      # her.py:109
    u_her = rng.uniform(size=batch_size)                             # her.py:115
    u_off = rng.uniform(size=batch_size)                             # her.py:116
    return episode_idxs, t_samples, u_her, u_off


def her_index_math(t_samples, u_her, u_off, T, future_p):
    """her.py:115-118: HER mask and the future time step (float64 math, truncation)."""
    her_mask = u_her < future_p
    future_offset = (u_off * (T - t_samples)).astype(int)
    future_t = t_samples + 1 + future_offset
    return her_mask, future_t


def _task_tables(tasks_ag_id, tasks_g_id):
    nb = len(tasks_g_id)
    # her.py:146-149: ag ids truncated to the goal-id length
    ag = [list(tasks_ag_id[j][:len(tasks_g_id[j])]) for j in range(nb)]
    g = [list(tasks_g_id[j]) for j in range(nb)]
    return ag, g


def apply_multi_task(episode_batch, draws, *, future_p, tasks_ag_id, tasks_g_id, task_replay,
                     reward_fun, task_to_replay=None, replay_tasks=None):
    """Deterministic part of her.py:99-183.

    replay_tasks: per-HER-sample task for the single-buffer modes that draw it from the
    RNG inside the loop (her.py:138-142); None for the multi-buffer modes."""
    episode_idxs, t_samples, u_her, u_off = draws
    T = episode_batch['u'].shape[1]
    B = episode_idxs.shape[0]
    transitions = {k: episode_batch[k][episode_idxs, t_samples].copy() for k in episode_batch}
    her_mask, future_t = her_index_math(t_samples, u_her, u_off, T, future_p)
    her_idx = np.where(her_mask)[0]
    future_ag = episode_batch['ag'][episode_idxs[her_idx], future_t[her_idx]]      # her.py:125
    ag_ids, g_ids = _task_tables(tasks_ag_id, tasks_g_id)
    multiple_buffers = ('buffer' in task_replay) or task_replay == 'hand_designed'  # her.py:94-97

    cur_task = np.argmax(transitions['task_descr'][her_idx], axis=1) if her_idx.size else \
        np.zeros(0, dtype=int)
    if task_replay != 'replay_current_task_transition':
        if multiple_buffers:
            if task_to_replay is None:
                rtask = cur_task                                   # her.py:132-134
            else:
                rtask = np.full(her_idx.size, task_to_replay, dtype=int)   # her.py:135-136
        else:
            rtask = np.asarray(replay_tasks, dtype=int)           # her.py:138-142
        g = transitions['g']
        td = transitions['task_descr']
        g[her_idx] = 0                                             # her.py:151
        td[her_idx] = 0                                            # her.py:152
        for j in range(len(g_ids)):
            rows = her_idx[rtask == j]
            if rows.size == 0:
                continue
            g[np.ix_(rows, g_ids[j])] = future_ag[rtask == j][:, ag_ids[j]]    # her.py:154
            td[rows, j] = 1                                        # her.py:155
    else:
        g = transitions['g']
        for j in range(len(g_ids)):
            sel = cur_task == j
            rows = her_idx[sel]
            if rows.size == 0:
                continue
            g[np.ix_(rows, g_ids[j])] = future_ag[sel][:, ag_ids[j]]           # her.py:164

    info = {k.replace('info_', ''): v for k, v in transitions.items() if k.startswith('info_')}
    transitions['r'] = reward_fun(ag_2=transitions['ag_2'], g=transitions['g'],
                                  task_descr=transitions['task_descr'], info=info)  # her.py:174-176
    transitions = {k: v.reshape(B, *v.shape[1:]) for k, v in transitions.items()}
    return transitions


def make_sample_multi_task_her_transitions(goal_replay, her_replay_k, task_replay, reward_fun,
                                           tasks_ag_id=None, tasks_g_id=None, rng=None):
    """Oracle counterpart of her.py:72.  `rng` defaults to the global np.random stream."""
    future_p = future_p_of(goal_replay, her_replay_k)
    nb_tasks = len(tasks_ag_id)
    multiple_buffers = ('buffer' in task_replay) or task_replay == 'hand_designed'

    def _sample(episode_batch, batch_size_in_transitions, task_to_replay=None, cp_proba=None):
        r = np.random if rng is None else rng
        T = episode_batch['u'].shape[1]
        E = episode_batch['u'].shape[0]
        draws = draw_her(r, E, T, batch_size_in_transitions)
        replay_tasks = None
        if task_replay != 'replay_current_task_transition' and not multiple_buffers:
            n_her = int((draws[2] < future_p).sum())
            if task_replay == 'replay_random_task_transition':
                replay_tasks = [r.choice(range(nb_tasks)) for _ in range(n_her)]          # her.py:139
            elif task_replay == 'replay_cp_task_transition':
                replay_tasks = [r.choice(range(nb_tasks), p=cp_proba) for _ in range(n_her)]  # :142
        return apply_multi_task(episode_batch, draws, future_p=future_p, tasks_ag_id=tasks_ag_id,
                                tasks_g_id=tasks_g_id, task_replay=task_replay, reward_fun=reward_fun,
                                task_to_replay=task_to_replay, replay_tasks=replay_tasks)

    return _sample


def apply_flat(episode_batch, draws, *, future_p, tasks_ag_id, tasks_g_id, reward_fun):
    """Deterministic part of her.py:20-66 (flat structure)."""
    episode_idxs, t_samples, u_her, u_off = draws
    T = episode_batch['u'].shape[1]
    B = episode_idxs.shape[0]
    transitions = {k: episode_batch[k][episode_idxs, t_samples].copy() for k in episode_batch}
    her_mask, future_t = her_index_math(t_samples, u_her, u_off, T, future_p)
    her_idx = np.where(her_mask)[0]
    future_ag = episode_batch['ag'][episode_idxs[her_idx], future_t[her_idx]]
    ag_id = []
    for t in range(len(tasks_g_id)):
        ag_id.extend(tasks_ag_id[t][:len(tasks_g_id[t])])            # her.py:44-46
    transitions['g'][her_idx] = future_ag[:, ag_id]                  # her.py:47
    info = {k.replace('info_', ''): v for k, v in transitions.items() if k.startswith('info_')}
    transitions['r'] = reward_fun(ag_2=transitions['ag_2'], g=transitions['g'],
                                  task_descr=None, info=info)        # her.py:56-59
    return {k: v.reshape(B, *v.shape[1:]) for k, v in transitions.items()}


def make_sample_her_transitions(goal_replay, her_replay_k, reward_fun, task_replay='',
                                tasks_ag_id=None, tasks_g_id=None, rng=None):
    """Oracle counterpart of her.py:5."""
    future_p = future_p_of(goal_replay, her_replay_k)

    def _sample(episode_batch, batch_size_in_transitions, task_to_replay=None, cp_proba=None):
        r = np.random if rng is None else rng
        T = episode_batch['u'].shape[1]
        E = episode_batch['u'].shape[0]
        draws = draw_her(r, E, T, batch_size_in_transitions)
        return apply_flat(episode_batch, draws, future_p=future_p, tasks_ag_id=tasks_ag_id,
                          tasks_g_id=tasks_g_id, reward_fun=reward_fun)

    return _sample
