"""Training-resumable checkpoints.

The reference can only pickle a policy for acting: Adam moments (mpi_adam.py:14-16), replay buffers and competence
queues are not saved (docstring at ddpg.py:511-513: "after unpickling you cannot continue training"), so a run cannot be
resumed.  Here everything a bit-exact continuation needs is written next to the reference-format `policy_*.pkl` /
`*_weights.pkl` files (DDPG.save_weights, ddpg.py:481-497: those stay loadable by the reference), on the reference's
own save cadence (train.py:195-205, `policy_save_interval`):

  per policy     parameters, target parameters, Adam m / v / t, the device step counter, normaliser state and
                 accumulators, competence progress, the counters that key the device RNG streams (noise, statistics
                 batches, slot draws)
  per buffer     stored episodes and counters of EVERY replay buffer of EVERY virtual rank of the process (config.py:210-214:
                 one set per rank), once per pool slot -- the experts of a task_experts job share theirs (train.py:285-291)
  per worker     competence queues, task probabilities, histories, SAGG-RIAC selectors, the env episode counters, the host
                 stream of every virtual rank (train.py:242-243)
  per process    NumPy / Python / torch host RNG states, what the training loop carries (epoch, best success rate, ...)

One file per process and checkpoint (`training_state/rank003_epoch000050.pt`); rank 0 writes `training_state/LATEST.json`
AFTER every process has finished its file (a barrier), then the files of older checkpoints go: whatever a job dies in the
middle of, LATEST.json names a complete set.  `experiment.train --resume DIR` continues from it.
"""
import copy
import json
import os
import random
from itertools import islice, repeat

import numpy as np
import torch

from curious_amd import dist

HISTORY_TAIL = 100             # entries of a worker's task / goal history a checkpoint keeps (worker_state)

STATE_DIR = 'training_state'
FORMAT = 2


# ---------------------------------------------------------------------------------------------------------- policies
def _buffers_of(policy):
    """[(pool slot, buffer)] of every distinct replay buffer the policy samples from: all virtual ranks' lists
    (`policy._rank_buffers`; `policy.buffer` is rank 0's alone), aliased logical buffers once."""
    lists = getattr(policy, '_rank_buffers', None)
    if lists is None:
        lists = [policy.buffer]
    seen, out = set(), []
    for bl in lists:
        for b in (bl if isinstance(bl, list) else [bl]):
            if b is not None and id(b) not in seen:
                seen.add(id(b))
                out.append((int(b.pool_index), b))
    return out


def _pooled(policies):
    """{(ordinal of the pool among the policies', pool slot): buffer}, in first-seen order."""
    pools, by_key = [], {}
    for p in policies:
        for slot, b in _buffers_of(p):
            if id(b.pool) not in pools:
                pools.append(id(b.pool))
            by_key.setdefault((pools.index(id(b.pool)), slot), b)
    return by_key


def buffers_state(policies):
    """Every buffer of the pools the policies sample from, once (experts share their buffers)."""
    return [dict(pool=pool, slot=slot, current_size=int(b.current_size), n_transitions_stored=int(b.n_transitions_stored),
                 size=int(b.size), records=b.records[:b.current_size].cpu())
            for (pool, slot), b in _pooled(policies).items()]


def load_buffers_state(policies, states):
    by_key = _pooled(policies)
    if len(states) != len(by_key):
        raise ValueError('checkpoint holds %d replay buffers, this job has %d (another --num_cpu / structure / '
                         'task_replay?)' % (len(states), len(by_key)))
    for bs in states:
        b = by_key[(bs['pool'], bs['slot'])]
        if bs['size'] != b.size:
            raise ValueError('checkpointed buffer holds %d episodes at most, this job\'s %d' % (bs['size'], b.size))
        b.current_size, b.n_transitions_stored = bs['current_size'], bs['n_transitions_stored']
        b.records[:b.current_size].copy_(bs['records'])
        b.pool.version += 1


def policy_state(policy, with_buffers=True):
    st = dict(
        theta=policy.theta.cpu(), theta_target=policy.theta_target.cpu(), m=policy._m.cpu(), v=policy._v.cpu(),
        t_Q=policy.Q_adam.t, t_pi=policy.pi_adam.t, step_ctr=int(policy._step_ctr),
        o_stats=policy.o_stats.state.cpu(), g_stats=policy.g_stats.state.cpu(), stats_acc=policy._stats_acc.cpu(),
        cp=None if policy.cp is None else np.asarray(policy.cp, dtype=np.float64).copy(),
        noise_counter=policy._noise_counter, stats_calls=getattr(policy, '_stats_calls', 0),
        store_calls=getattr(policy, '_store_calls', 0), n_episodes=getattr(policy, 'n_episodes', None),
        fault_tick=getattr(policy, '_fault_tick', 0),
        shape=(int(policy.P_total), int(policy.V), int(policy.rank_base), int(policy.total_ranks)))
    if with_buffers:
        st['buffers'] = buffers_state([policy])
    return st


def load_policy_state(policy, st):
    shape = (int(policy.P_total), int(policy.V), int(policy.rank_base), int(policy.total_ranks))
    if tuple(st.get('shape', shape)) != shape:
        raise ValueError('checkpoint of a policy with (parameters, virtual ranks, first rank, ranks) = %s, this one has %s'
                         % (tuple(st['shape']), shape))
    policy.settle()
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
    if st.get('n_episodes') is not None:
        policy.n_episodes = st['n_episodes']
    policy._fault_tick = st.get('fault_tick', 0)
    if 'buffers' in st:
        load_buffers_state([policy], st['buffers'])
    # everything derived from the counters is rebuilt lazily
    policy._tables_dirty = True
    policy._batch_stale = True
    policy._alpha_filled = 0


# ----------------------------------------------------------------------------------------------------------- workers
def _env_state(w):
    if getattr(w, 'batched', False):
        return dict(kind='batched', episode=w.benv.episode.cpu(), seed=w.benv._seed)
    envs = []
    for e in w.envs:                                                 # host envs: the synthetic arm keeps a counter per env;
        b = getattr(getattr(e, 'unwrapped', e), '_b', None)          # a real env's state is its own business (the
        envs.append(None if b is None else (b.episode.cpu(), b._seed))   # reference cannot checkpoint MuJoCo either)
    return dict(kind='list', envs=envs)


def _load_env_state(w, st):
    if st['kind'] == 'batched':
        if w.benv._seed != st['seed']:
            w.benv.seed(st['seed'])
        w.benv.episode.copy_(st['episode'])
        return
    for e, es in zip(w.envs, st['envs']):
        b = getattr(getattr(e, 'unwrapped', e), '_b', None)
        if b is not None and es is not None:
            if b._seed != es[1]:
                b.seed(es[1])
            b.episode.copy_(es[0])


def worker_state(w):
    st = dict(n_episodes=w.n_episodes, C=np.asarray(w.C).copy(), CP=np.asarray(w.CP).copy(),
              success_history=list(w.success_history), reward_history=list(w.reward_history),
              Q_history=list(w.Q_history), count=w.count, exploit=bool(w.exploit), env=_env_state(w),
              vrng=None if w._vrng is None else [r.get_state() for r in w._vrng])
    if hasattr(w, 'competence_computers'):
        st['p'] = np.asarray(w.p).copy()
        st['queues'] = [(list(q.successes), q.C, q.CP) for q in w.competence_computers]
        # The histories grow by a rollout's worth of entries per cycle for the whole job (rollout.py:392-393: unbounded upstream
        # too) and only the last 100 tasks are ever read (the '%_task' columns, rollout.py:475): their length and their last
        # HISTORY_TAIL entries are the state -- pickling two million entries took the writer thread, and the epoch beside it,
        # up to 0.7 s per checkpoint 250 epochs into a job
        for name in ('task_history', 'goal_history'):
            h = getattr(w, name)
            tail = list(islice(reversed(h), HISTORY_TAIL))[::-1]
            st[name] = dict(n=len(h), tail=[list(g) if isinstance(g, (list, tuple, np.ndarray)) else g for g in tail])
    if hasattr(w, 'goal_selectors'):                                 # SAGG-RIAC (plain Python objects)
        st['goal_selectors'] = copy.deepcopy(w.goal_selectors)      # (a snapshot: the file may be written behind the loop)
        st['split_histories'] = [list(h) for h in w.split_histories]
    return st


def load_worker_state(w, st):
    w.settle()
    w.n_episodes, w.C, w.CP, w.count = st['n_episodes'], st['C'], st['CP'], st['count']
    w.exploit = st.get('exploit', w.exploit)
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
        for name in ('task_history', 'goal_history'):
            h, saved = getattr(w, name), st.get(name, [])
            h.clear()
            if isinstance(saved, dict):                              # length + tail; the entries in front are placeholders
                from curious_amd.rollout_eval import _NOTHING
                h.extend(repeat(_NOTHING, saved['n'] - len(saved['tail'])))
                h.extend(saved['tail'])
            else:                                                    # (files written before the histories were cut)
                h.extend(saved)
    if 'goal_selectors' in st:
        w.goal_selectors = st['goal_selectors']
        for h, saved in zip(w.split_histories, st['split_histories']):
            h.clear()
            h.extend(saved)
    if st.get('vrng') is not None:
        if w._vrng is None or len(w._vrng) != len(st['vrng']):
            w._vrng = [np.random.RandomState(0) for _ in st['vrng']]
        for r, s in zip(w._vrng, st['vrng']):
            r.set_state(s)
    if 'env' in st:
        _load_env_state(w, st['env'])
        w.__dict__['_ep_host'] = None                                # (the host mirror of benv.episode: re-read on demand)
        w.__dict__.pop('_eval_env', None)
    elif 'env_episode' in st:                                        # (format 1)
        w.benv.episode.copy_(st['env_episode'])


# ------------------------------------------------------------------------------------ one policy, in one file (format 1 API)
def _settle(policies, workers):
    for w in workers:                                                # async_store: flags / routing the host has not
        if hasattr(w, 'settle'):                                     # mirrored yet
            w.settle()
    for p in policies:
        if hasattr(p, 'settle'):
            p.settle()
    torch.cuda.synchronize()


def save_training_state(path, policy, workers=()):
    _settle([policy], workers)
    torch.save(dict(policy=policy_state(policy), workers=[worker_state(w) for w in workers],
                    numpy_rng=np.random.get_state()), path)


def load_training_state(path, policy, workers=()):
    ck = torch.load(path, weights_only=False)
    load_policy_state(policy, ck['policy'])
    for w, st in zip(workers, ck['workers']):
        load_worker_state(w, st)
    np.random.set_state(ck['numpy_rng'])
    torch.cuda.synchronize()


# --------------------------------------------------------------------------------- a whole job: every process, every epoch
def _flat(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _layout_of(policies):
    p = policies[0]
    return dict(world=dist.world_size(), rank=dist.rank(), virtual_ranks=int(p.V), rank_base=int(p.rank_base),
                total_ranks=int(p.total_ranks), n_policies=len(policies))


def _file(dirpath, rank, epoch):
    return os.path.join(dirpath, STATE_DIR, 'rank%03d_epoch%06d.pt' % (rank, epoch))


def save_job_state(dirpath, epoch, policy, workers, expert_bank=None, loop=None, writer=None):
    """Collective: every process of the job writes its file for `epoch`, then rank 0 publishes LATEST.json and the files of
    earlier checkpoints are removed.  policy: a DDPG or the list of experts; workers: every RolloutWorker of the process
    (training workers first, then the evaluator); loop: what the training loop needs to go on (a picklable dict).
    writer (curious_amd.util.BackgroundWriter; one process only): the state is copied to the host here, the file -- up to
    20 GB with the reference's 19 ranks on full buffers -- and LATEST.json are written behind the training loop (several
    processes publish behind a barrier, which stays on the training thread)."""
    policies = _flat(policy)
    workers = [w for w in _flat(workers) if w is not None]
    _settle(policies, workers)
    os.makedirs(os.path.join(dirpath, STATE_DIR), exist_ok=True)
    state = dict(format=FORMAT, epoch=int(epoch), layout=_layout_of(policies),
                 policies=[policy_state(p, with_buffers=False) for p in policies], buffers=buffers_state(policies),
                 workers=[worker_state(w) for w in workers],
                 bank=None if expert_bank is None else dict(cur=int(expert_bank._cur), batched=bool(expert_bank.batched)),
                 loop=dict(loop or {}), numpy_rng=np.random.get_state(), python_rng=random.getstate(),
                 torch_rng=torch.get_rng_state())
    path = _file(dirpath, dist.rank(), epoch)
    latest = os.path.join(dirpath, STATE_DIR, 'LATEST.json')
    world, rank = dist.world_size(), dist.rank()

    def write():
        torch.save(state, path + '.tmp')
        os.replace(path + '.tmp', path)

    def publish():
        if rank == 0:
            with open(latest + '.tmp', 'w') as f:
                json.dump(dict(format=FORMAT, epoch=int(epoch), world=world), f)
            os.replace(latest + '.tmp', latest)

    def sweep():
        mine = 'rank%03d_epoch' % rank
        for name in os.listdir(os.path.join(dirpath, STATE_DIR)):
            if name.startswith(mine) and name != os.path.basename(path):
                try:
                    os.remove(os.path.join(dirpath, STATE_DIR, name))
                except OSError:
                    pass
    if writer is not None and world == 1:
        writer.submit(lambda: (write(), publish(), sweep()))          # (in this order, behind earlier jobs of the writer)
        return path
    write()
    dist.barrier()                                                   # every file of this checkpoint is complete
    publish()
    dist.barrier()                                                   # ... and published: older files may go
    sweep()
    return path


def latest_epoch(dirpath):
    """Epoch of the last complete checkpoint under `dirpath`, or None."""
    try:
        with open(os.path.join(dirpath, STATE_DIR, 'LATEST.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def load_job_state(dirpath, policy, workers, expert_bank=None):
    """Restore this process's part of the last complete checkpoint under `dirpath`; returns (epoch, loop dict).  The job
    must be laid out as the one that saved it (same processes, same virtual ranks per process)."""
    meta = latest_epoch(dirpath)
    if meta is None:
        raise FileNotFoundError('no complete checkpoint under %s (no %s/LATEST.json)' % (dirpath, STATE_DIR))
    if meta['world'] != dist.world_size():
        raise ValueError('checkpoint of a %d-process job, this job has %d' % (meta['world'], dist.world_size()))
    policies = _flat(policy)
    workers = [w for w in _flat(workers) if w is not None]
    ck = torch.load(_file(dirpath, dist.rank(), meta['epoch']), weights_only=False)
    if ck['epoch'] != meta['epoch'] or ck['layout'] != _layout_of(policies):
        raise ValueError('checkpoint file does not belong to this job: saved by %s at epoch %d, this process is %s'
                         % (ck['layout'], ck['epoch'], _layout_of(policies)))
    if len(ck['workers']) != len(workers):
        raise ValueError('checkpoint holds %d rollout workers, this job has %d' % (len(ck['workers']), len(workers)))
    for p, st in zip(policies, ck['policies']):
        load_policy_state(p, st)
    load_buffers_state(policies, ck['buffers'])
    for w, st in zip(workers, ck['workers']):
        load_worker_state(w, st)
    if expert_bank is not None and ck.get('bank') is not None:
        expert_bank._cur, expert_bank.batched = ck['bank']['cur'], ck['bank']['batched']
        for x in expert_bank.experts:
            x._cur = expert_bank._cur
            x._batch_stale = True
    np.random.set_state(ck['numpy_rng'])
    random.setstate(ck['python_rng'])
    torch.set_rng_state(ck['torch_rng'])
    torch.cuda.synchronize()
    dist.barrier()
    return ck['epoch'], ck['loop']
