"""BASELINE.json configs[1] at its FULL sizes (MultiTaskFetchArm4-v5: five per-task buffers of 20 000 episodes = 10^6
transitions each -- 19 998 after the reference's rounding --, 256 parallel rollouts x T = 50, batch 256, 100 updates per cycle), checked through properties that do
not need an oracle run of that size: every gathered transition is an exact copy of the stored rows its own signature
names, HER goals come from a later step of the same episode, rewards recompute with the oracle's reward function, the
replay proportions are met draw by draw, a store into full buffers conserves everything it does not overwrite (later
episode wins a contested slot), and a whole cycle is deterministic."""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

CAP = 19998        # (10^6 - 10^6 % rollout_batch_size) // T episodes per buffer: config.py:190-199 at 256 rollouts


def _job(seed=0, env=None, b_r=256):
    params, dims, policy, worker = bench.build_job(use_graph=True, seed=seed, env=env, b_r=b_r)
    assert policy.batch_size == 256 and worker.rollout_batch_size == b_r
    assert policy.buffer[1].size == (CAP if b_r == 256 else (10 ** 6 - 10 ** 6 % b_r) // 50)
    return policy, worker


def _fill_signatures(policy):
    """Every slot of every buffer gets an episode whose rows say where they sit: o[..., 0:4] = (e // 128, e % 128, t,
    buffer) -- all below clip_obs = 200, which the sampler applies -- and values derived from (e, t, column) elsewhere."""
    lay, dev = policy._layout, policy.device
    T, O, AG = lay.T, lay.dims['o'], lay.dims['ag']
    CAP = policy.buffer[1].size
    e = torch.arange(CAP, device=dev, dtype=torch.float32)[:, None, None]
    t = torch.arange(T + 1, device=dev, dtype=torch.float32)[None, :, None]
    for i in range(1, min(policy.nb_tasks, 5) + 1):
        buf = policy.buffer[i]
        assert buf.pool_index == i                                   # (buffers 6.. of Arm8 are buffer 5: ddpg.py:106-110)
        rec = buf.records
        rec.zero_()
        v = lay.record_views(rec)
        k = torch.arange(O, device=dev, dtype=torch.float32)[None, None, :]
        o = 0.001 * torch.remainder(e * 31 + t * 17 + k * 7, 997.)
        o[:, :, 0] = torch.floor(e[:, :, 0] / 128)
        o[:, :, 1] = torch.remainder(e[:, :, 0], 128.)
        o[:, :, 2] = t[:, :, 0]
        o[:, :, 3] = float(i)
        v['o'].copy_(o)
        ka = torch.arange(AG, device=dev, dtype=torch.float32)[None, None, :]
        ag = 0.002 * torch.remainder(e * 13 + t * 29 + ka * 3, 499.)
        v['ag'].copy_(ag)
        v['g'].copy_(5.0 + 0.5 * ag[:, :T])
        ku = torch.arange(4, device=dev, dtype=torch.float32)[None, None, :]
        v['u'].copy_(torch.remainder(e * 7 + t[:, :T] * 3 + ku, 200.) / 100. - 1.)
        td = torch.zeros([CAP, T, policy.nb_tasks], device=dev)
        td[:, :, i - 1] = 1.0
        v['task_descr'].copy_(td)
        ch = torch.zeros([CAP, T, AG], device=dev)
        ch[:, :, 3 * (i - 1):3 * i] = 1.0
        v['change'].copy_(ch)
        buf.current_size = CAP
        buf.n_transitions_stored = CAP * T
    policy._pool.version += 1
    policy._tables_dirty = True


def test_her_gather_from_full_buffers_returns_exact_rows_and_future_goals():
    from oracle.reward import make_reward_fun
    policy, _ = _job()
    _fill_signatures(policy)
    policy.cp = np.array([0.3, 0.0, 0.2, 0.1])
    lay, T = policy._layout, policy._layout.T
    n_draws, B = 100, policy.batch_size
    rows = []
    for k in range(n_draws):
        rows.append(policy._sample_packed().clone())
        policy._step_ctr += 1                                        # what an update does: the next draw has its own stream
    assert policy.proportions.sum() == B and policy.proportions[0] == 0
    batch = torch.cat(rows)
    v = lay.batch_views(batch)
    o = v['o']
    e = (o[:, 0] * 128 + o[:, 1]).long()
    t = o[:, 2].long()
    b = o[:, 3].long()
    assert int(e.min()) >= 0 and int(e.max()) < CAP and int(t.min()) == 0 and int(t.max()) == T - 1
    assert int(e.max()) > 0.98 * CAP and int(e.min()) < 0.02 * CAP   # the whole capacity is addressed
    assert set(b.unique().tolist()) <= set(range(1, 5))
    # the replay proportions of ddpg.py:272-286, draw by draw
    per_draw = torch.stack([(b.view(n_draws, B) == i).sum(1) for i in range(5)], 1).cpu().numpy()
    assert (per_draw == policy.proportions[None, :5]).all()
    # exact copies of the rows the signature names, and of the NEXT row of the same episode
    st = policy._pool.storage
    src, nxt = st[b, e, t], st[b, e, t + 1]
    for key, rows_ in (('o', src), ('u', src), ('ag', src), ('o_2', nxt), ('ag_2', nxt)):
        off, dim = lay.off[key.replace('_2', '')], lay.dims[key.replace('_2', '')]
        assert torch.equal(v[key], rows_[:, off:off + dim]), key
    for key in lay.extra_keys:
        assert torch.equal(v[key], src[:, lay.off[key]:lay.off[key] + lay.dims[key]]), key
    # task descriptor: the buffer's task (her.py:131-155); goal: kept (non-HER) or the achieved goal of a LATER step
    task = b - 1
    assert torch.equal(v['task_descr'], torch.nn.functional.one_hot(task, policy.nb_tasks).float())
    g, g_stored = v['g'], src[:, lay.off['g']:lay.off['g'] + lay.dims['g']]
    her = ~(g == g_stored).all(1)
    frac = float(her.float().mean())
    assert 0.78 < frac < 0.82                                        # future_p = 1 - 1 / (1 + 4), 25 600 draws
    slots = (3 * task)[:, None] + torch.arange(3, device=g.device)[None, :]
    on_slots = torch.gather(g, 1, slots)
    outside = g.clone()
    outside.scatter_(1, slots, 0.0)
    assert float(outside[her].abs().sum()) == 0.0                    # her.py:148-149: zeroed except the task's slots
    fut = st[b, e][:, :, lay.off['ag']:lay.off['ag'] + lay.dims['ag']]          # [n, T + 1, AG]
    fut = torch.gather(fut, 2, slots[:, None, :].expand(-1, T + 1, -1))         # the task's slots at every step
    later = torch.arange(T + 1, device=g.device)[None, :] > t[:, None]
    match = ((fut == on_slots[:, None, :]).all(2) & later).any(1)
    assert bool(match[her].all())
    assert torch.equal(v['g_2'], g)
    # rewards recompute with the oracle's reward function on exactly these arrays
    ag_ids = [list(range(3 * j, 3 * j + 3)) for j in range(policy.nb_tasks)]
    r = make_reward_fun(ag_ids, ag_ids)(v['ag_2'].cpu().numpy(), g.cpu().numpy(), v['task_descr'].cpu().numpy(), None)
    assert np.array_equal(v['r'].cpu().numpy(), r)
    assert set(np.unique(r)) <= {0.0, -1.0} and (r == 0).any() and (r == -1).any()


def test_store_into_full_buffers_overwrites_random_slots_and_conserves_the_rest():
    policy, _ = _job()
    _fill_signatures(policy)
    lay, T, nb = policy._layout, policy._layout.T, policy.nb_tasks
    rng = np.random.RandomState(11)
    E = 256
    ep = dict(o=np.zeros([E, T + 1, lay.dims['o']], np.float32), u=rng.uniform(-1, 1, [E, T, 4]).astype(np.float32),
              g=rng.randn(E, T, 12).astype(np.float32), ag=rng.randn(E, T + 1, 12).astype(np.float32),
              task_descr=np.zeros([E, T, nb], np.float32), change=np.zeros([E, T, 12], np.float32),
              info_is_success=np.zeros([E, T, 1], np.float32))
    ep['o'][:, :, 0] = 1.0e6 + np.arange(E)[:, None]                 # ids no stored episode carries
    active = rng.rand(E, nb) < 0.45
    for j in range(nb):
        ep['change'][active[:, j], :, 3 * j:3 * j + 3] = 1.0
    ep['task_descr'][np.arange(E), :, rng.randint(0, nb, E)] = 1.0
    st = policy._pool.storage
    col = lay.off['o']
    before = st[:, :, 0, col].clone()                                # [buffers, CAP]: the id column of every slot
    whole_before = st.clone()
    stored0 = [policy.buffer[i].n_transitions_stored for i in range(nb + 1)]
    policy.store_episode(ep, np.zeros(nb), E)
    torch.cuda.synchronize()
    after = st[:, :, 0, col]
    changed = after != before
    assert not bool(changed[0].any())                                # buffer 0 is never written (ddpg.py:185)
    staged = torch.from_numpy(ep['o'][:, :, 0]).to(st.device)
    for j in range(nb):
        i = j + 1
        buf = policy.buffer[i]
        assert buf.current_size == CAP and buf.n_transitions_stored == stored0[i] + int(active[:, j].sum()) * T
        new_ids = after[i][changed[i]]
        ids = (new_ids - 1.0e6).long().cpu().numpy()
        assert len(set(ids.tolist())) == ids.size and set(ids.tolist()) <= set(np.nonzero(active[:, j])[0].tolist())
        # an active episode that is absent lost its slot to a LATER active episode of the same batch -- at most a handful
        lost = sorted(set(np.nonzero(active[:, j])[0].tolist()) - set(ids.tolist()))
        assert len(lost) <= 6 and all(x < ids.max() for x in lost)
        # whole records arrived, untouched slots are bit-identical to what they held
        slots = torch.nonzero(changed[i])[:, 0]
        got = st[i, slots]
        want = torch.from_numpy(np.ascontiguousarray(ep['o'][ids])).to(st.device)
        assert torch.equal(got[:, :, col:col + lay.dims['o']], want)
        keep = ~changed[i]
        assert torch.equal(st[i][keep], whole_before[i][keep])
    del staged


def test_a_full_size_cycle_is_deterministic_and_keeps_its_invariants():
    """Two jobs with the same seeds: rollout of 256 envs x 50 steps -> store -> 100 updates -> Polyak, three times; the
    parameters, moments, target networks and buffer contents agree bit for bit; counters and sizes are what the loop
    implies."""
    res = []
    for _ in range(2):
        np.random.seed(3)
        policy, worker = _job(seed=7)
        bench.prefill(policy, 2048, seed=0)
        for _ in range(3):
            bench.cycle(policy, worker)
        policy.settle()
        worker.settle()
        torch.cuda.synchronize()
        policy.check_faults()
        assert int(policy._step_ctr) == 300 == policy.Q_adam.t == policy.pi_adam.t
        assert all(torch.isfinite(x).all() for x in (policy.theta, policy.theta_target, policy._m, policy._v))
        assert not torch.equal(policy.theta, policy.theta_target)
        sizes = [policy.buffer[i].current_size for i in range(1, 5)]
        assert all(2048 <= s <= 2048 + 3 * 256 for s in sizes) and max(sizes) > 2048
        stored = [policy._pool.storage[i, :sizes[i - 1]].clone() for i in range(1, 5)]   # (the rest was never written)
        res.append((policy.theta.clone(), policy.theta_target.clone(), policy._m.clone(), policy._v.clone(), stored,
                    sizes, worker.n_episodes))
    a, b = res
    assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and a[5] == b[5] and a[6] == b[6] == 3 * 256
    assert all(torch.equal(x, y) for x, y in zip(a[4], b[4]))


def test_her_gather_from_full_aliased_buffers_arm8():
    """BASELINE configs[2] (MultiTaskFetchArm8-v5, 1 024 rollouts: buffers of 19 988 episodes): the logical buffers 6, 7, 8
    of the distractor tasks ARE buffer 5 (ddpg.py:106-110), only tasks < 5 are ever routed (ddpg.py:183), yet every logical
    buffer is sampled with its own task_to_replay (ddpg.py:326-336) -- rows from pool slot 5 therefore come relabelled to
    tasks 4..7, goals written on THAT task's slots from the episode's own later achieved goal."""
    policy, _ = _job(env='MultiTaskFetchArm8-v5', b_r=1024)
    assert policy.buffer[8] is policy.buffer[5] and policy.buffer[6].pool_index == 5
    _fill_signatures(policy)
    cap = policy.buffer[1].size
    assert cap == 19988
    policy.cp = np.array([0.3, 0.0, 0.2, 0.1, 0.05, 0.15, 0.0, 0.2])
    lay, T, nb = policy._layout, policy._layout.T, policy.nb_tasks
    n_draws, B = 60, policy.batch_size
    rows = []
    for k in range(n_draws):
        rows.append(policy._sample_packed().clone())
        policy._step_ctr += 1
    prop = policy.proportions
    assert prop.sum() == B and prop[0] == 0 and (prop[5:] > 0).sum() >= 3
    v = lay.batch_views(torch.cat(rows))
    o = v['o']
    e = (o[:, 0] * 128 + o[:, 1]).long()
    t = o[:, 2].long()
    b = o[:, 3].long()                                               # PHYSICAL buffer (pool slot) of the row
    assert int(e.max()) < cap and int(e.max()) > 0.98 * cap and set(b.unique().tolist()) <= set(range(1, 6))
    per_draw = torch.stack([(b.view(n_draws, B) == i).sum(1) for i in range(6)], 1).cpu().numpy()
    want = np.array([0, prop[1], prop[2], prop[3], prop[4], prop[5:].sum()])
    assert (per_draw == want[None, :]).all()                         # slot 5 serves logical buffers 5..8 together
    st = policy._pool.storage
    src, nxt = st[b, e, t], st[b, e, t + 1]
    for key, rows_ in (('o', src), ('u', src), ('ag', src), ('o_2', nxt), ('ag_2', nxt)):
        off, dim = lay.off[key.replace('_2', '')], lay.dims[key.replace('_2', '')]
        assert torch.equal(v[key], rows_[:, off:off + dim]), key
    g, g_stored = v['g'], src[:, lay.off['g']:lay.off['g'] + lay.dims['g']]
    her = ~(g == g_stored).all(1)
    assert 0.77 < float(her.float().mean()) < 0.83
    task = v['task_descr'].argmax(1)
    assert bool((v['task_descr'].sum(1) == 1).all())
    # rows kept as stored carry the stored descriptor (the pool slot's task); relabelled rows of slot i < 5 the task i - 1,
    # relabelled rows of slot 5 one of the tasks 4..7 -- each as often as its logical buffer's share of the batch says
    assert bool((task[~her] == b[~her] - 1).all())
    low = her & (b < 5)
    assert bool((task[low] == b[low] - 1).all())
    hi = her & (b == 5)
    assert set(task[hi].unique().tolist()) == {k for k in range(4, 8) if prop[k + 1] > 0}
    for k in range(4, 8):
        n_k = int((task[hi] == k).sum())
        expect = 0.8 * n_draws * prop[k + 1]
        assert abs(n_k - expect) <= 4 * np.sqrt(max(expect, 1.0)) + 1, (k, n_k, expect)
    slots = (3 * task)[:, None] + torch.arange(3, device=g.device)[None, :]
    on_slots = torch.gather(g, 1, slots)
    outside = g.clone()
    outside.scatter_(1, slots, 0.0)
    assert float(outside[her].abs().sum()) == 0.0
    fut = st[b, e][:, :, lay.off['ag']:lay.off['ag'] + lay.dims['ag']]
    fut = torch.gather(fut, 2, slots[:, None, :].expand(-1, T + 1, -1))
    later = torch.arange(T + 1, device=g.device)[None, :] > t[:, None]
    assert bool(((fut == on_slots[:, None, :]).all(2) & later).any(1)[her].all())
    from oracle.reward import make_reward_fun
    ids = [list(range(3 * j, 3 * j + 3)) for j in range(nb)]
    r = make_reward_fun(ids, ids)(v['ag_2'].cpu().numpy(), g.cpu().numpy(), v['task_descr'].cpu().numpy(), None)
    assert np.array_equal(v['r'].cpu().numpy(), r)


def test_full_size_batched_experts_cycles_are_deterministic():
    """BASELINE configs[4] at full size on one GPU: 4 experts on the shared per-task buffers, every cycle one expert's 256
    rollouts and 100 batched updates of ALL experts (400 agent-updates in 200 launches).  Two identically seeded jobs agree
    bit for bit after 4 cycles (each expert has acted once); the experts have moved apart; each one's batches are
    relabelled to ITS task (ddpg.py:302-318, her.py:135-136)."""
    res = []
    for _ in range(2):
        np.random.seed(5)
        params, dims, bank, workers = bench.build_experts_job(use_graph=True, seed=3)
        bench.prefill(bank[0], 2048, seed=1)
        for k in range(4):
            bench.experts_cycle(bank, workers, k)
        for x in bank:
            x.settle()
        torch.cuda.synchronize()
        bank.check_faults()
        assert all(int(x._step_ctr) == 400 == x.Q_adam.t for x in bank)
        lay = bank[0]._layout
        for i, x in enumerate(bank):
            td = lay.batch_views(x._staged)['task_descr']
            # relabelled rows (future_p = 0.8) carry expert i's task; rows kept as stored carry what the (synthetic,
            # randomly labelled) prefill wrote
            assert float((td.argmax(1) == i).float().mean()) > 0.7 and bool((td.sum(1) == 1).all())
        res.append([x.theta.clone() for x in bank] + [x.theta_target.clone() for x in bank] +
                   [workers[i].n_episodes for i in range(4)])
    a, b = res
    assert all(torch.equal(x, y) for x, y in zip(a[:8], b[:8])) and a[8:] == b[8:]
    assert all(torch.isfinite(x).all() for x in a[:8])
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[2], a[3])


def test_nineteen_virtual_ranks_at_full_buffer_sizes():
    """The reference's published regime (--num_cpu 19, readme.md:16) at ITS full sizes on one GPU: 19 x 5 private buffers of
    20 000 episodes (33 GB in one pool: element offsets beyond 2^32), all full.  The joint gather of 19 x 256 transitions:
    rank v's block holds exact copies of rows of rank v's OWN buffers (its signature names buffer, episode, step AND rank)
    and of the next row of the same episode, addresses the whole capacity of every rank, meets the replay proportions
    per rank, relabels with a later achieved goal of the same episode of the same rank."""
    V = 19
    params, dims, policy, worker = bench.build_job(use_graph=True, b_r=2, virtual_ranks=V)
    cap = policy.buffer[1].size
    assert cap == 20000 and len(policy._rank_buffers) == V and policy._Bt == V * 256
    lay, dev = policy._layout, policy.device
    T, O, AG = lay.T, lay.dims['o'], lay.dims['ag']
    assert policy._pool.storage.numel() > 2 ** 32
    e = torch.arange(cap, device=dev, dtype=torch.float32)[:, None, None]
    t = torch.arange(T + 1, device=dev, dtype=torch.float32)[None, :, None]
    k = torch.arange(O, device=dev, dtype=torch.float32)[None, None, :]
    ka = torch.arange(AG, device=dev, dtype=torch.float32)[None, None, :]
    for v, bufs in enumerate(policy._rank_buffers):
        for i in range(1, 5):
            buf = bufs[i]
            assert buf.pool_index == v * 5 + i
            rec = buf.records
            rec.zero_()
            w = lay.record_views(rec)
            o = 0.001 * torch.remainder(e * 31 + t * 17 + k * 7 + v * 101, 997.)
            o[:, :, 0] = torch.floor(e[:, :, 0] / 128)
            o[:, :, 1] = torch.remainder(e[:, :, 0], 128.)
            o[:, :, 2] = t[:, :, 0]
            o[:, :, 3] = float(i)
            o[:, :, 4] = float(v)
            w['o'].copy_(o)
            ag = 0.002 * torch.remainder(e * 13 + t * 29 + ka * 3 + v * 37, 499.)
            w['ag'].copy_(ag)
            w['g'].copy_(5.0 + 0.5 * ag[:, :T])
            td = torch.zeros([cap, T, policy.nb_tasks], device=dev)
            td[:, :, i - 1] = 1.0
            w['task_descr'].copy_(td)
            ch = torch.zeros([cap, T, AG], device=dev)
            ch[:, :, 3 * (i - 1):3 * i] = 1.0
            w['change'].copy_(ch)
            buf.current_size = cap
            buf.n_transitions_stored = cap * T
            del o, ag, td, ch
    policy._pool.version += 1
    policy._tables_dirty = True
    policy.cp = np.array([0.3, 0.0, 0.2, 0.1])
    n_draws, B = 8, 256
    rows = []
    for _ in range(n_draws):
        rows.append(policy._sample_packed().clone())
        policy._step_ctr += 1
    batch = torch.cat(rows)                                          # [n_draws x V x 256, 152]
    w = lay.batch_views(batch)
    o = w['o']
    e_, t_, b_, r_ = (o[:, 0] * 128 + o[:, 1]).long(), o[:, 2].long(), o[:, 3].long(), o[:, 4].long()
    want_rank = torch.arange(V, device=dev).repeat_interleave(B).repeat(n_draws)
    assert torch.equal(r_, want_rank)                                # every row of rank v's block comes from rank v's buffers
    assert int(e_.max()) > 0.97 * cap and int(e_.min()) < 0.03 * cap and int(t_.max()) == T - 1
    for v in (0, 7, 18):
        sel = r_ == v
        assert int(e_[sel].max()) > 0.9 * cap                        # ... over the whole capacity, rank by rank
        per_draw = torch.stack([(b_[sel].view(n_draws, B) == i).sum(1) for i in range(5)], 1).cpu().numpy()
        assert (per_draw == policy.proportions[None, :5]).all()
    st = policy._pool.storage                                        # [V x 5, cap, T + 1, row]
    pb = r_ * 5 + b_
    src, nxt = st[pb, e_, t_], st[pb, e_, t_ + 1]
    for key, rows_ in (('o', src), ('u', src), ('ag', src), ('o_2', nxt), ('ag_2', nxt)):
        off, dim = lay.off[key.replace('_2', '')], lay.dims[key.replace('_2', '')]
        assert torch.equal(w[key], rows_[:, off:off + dim]), key
    task = b_ - 1
    g, g_stored = w['g'], src[:, lay.off['g']:lay.off['g'] + lay.dims['g']]
    her = ~(g == g_stored).all(1)
    assert 0.77 < float(her.float().mean()) < 0.83
    slots = (3 * task)[:, None] + torch.arange(3, device=dev)[None, :]
    on_slots = torch.gather(g, 1, slots)
    fut = st[pb, e_][:, :, lay.off['ag']:lay.off['ag'] + lay.dims['ag']]
    fut = torch.gather(fut, 2, slots[:, None, :].expand(-1, T + 1, -1))
    later = torch.arange(T + 1, device=dev)[None, :] > t_[:, None]
    assert bool(((fut == on_slots[:, None, :]).all(2) & later).any(1)[her].all())
    # ... and a whole cycle (38 rollouts, routed store into the ranks' full buffers, 100 updates of 4 864 rows) runs clean
    for _ in range(2):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    policy.check_faults(wait=True)
    assert np.isfinite(policy._losses.cpu().numpy()).all()
