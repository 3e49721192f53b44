#!/usr/bin/env python
"""Benchmark of the CURIOUS rollout-and-update hot path on MI355X (BASELINE.json metric / configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one training cycle of train.py:148-155 on MultiTaskFetchArm4-v5 dims, per GPU:
    generate_rollouts (256 GPU-resident envs x T=50)  ->  store_episode (+ normaliser update)
    ->  n_batches=100 x train() (HER sample of 256 transitions, DDPG grads, all-reduce, Adam)  ->  update_target_net
`value` = HER-sampled gradient transitions per second over the whole job (all GPUs); `env_steps_per_sec` is reported
beside it from the same timed region.  Data is synthetic (synthetic arm env, random-init weights); arithmetic is f32.
For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL); if it is started
directly with --gpus N > 1 it re-launches itself that way before touching the GPU.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA
L2_PEAK_TBS = 34.5              # MI355X_MICROARCH.md: aggregate L2 bandwidth (8 XCDs; 135 GB/s per CU)
L2_PER_CU_GBS = 135.0           # the same figure per CU (64 bytes per clock)

ENV = 'MultiTaskFetchArm4-v5'
B_R = 256                      # parallel rollouts per GPU (configs[1])
N_BATCHES = 100
BATCH = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-regime-line', action='store_true',
                    help="the default run (N = 1, headline configuration, with the CPU baseline) also measures the reference's "
                         "published --num_cpu 19 regime as 19 virtual ranks in a child process: skip that")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--prefill', type=int, default=2048, help='synthetic episodes pre-loaded per buffer')
    ap.add_argument('--env', default=ENV, help='diagnostic: another synthetic env (the headline is %s)' % ENV)
    ap.add_argument('--phases', action='store_true', help='diagnostic: time rollout / store / updates separately '
                                                            '(adds device syncs; not the headline number)')
    ap.add_argument('--rollout-batch-size', type=int, default=None,
                    help='diagnostic: parallel rollouts per GPU (configs[2]: --env MultiTaskFetchArm8-v5 with 1024); per '
                         'virtual rank with --virtual-ranks (default then: 2, the reference\'s)')
    ap.add_argument('--virtual-ranks', type=int, default=1,
                    help="diagnostic: V of the reference's MPI ranks per GPU (readme.md:16: the published runs use 19): V "
                         'private buffer sets / seeds / rollout groups, every update = V minibatches of 256 in one launch '
                         'sequence, gradients summed over them (mpi_adam.py:26-28) -- the regime cpu_baseline.ranks times')
    ap.add_argument('--num-cpu', type=int, default=0,
                    help="the reference's --num_cpu R laid out over the --gpus processes exactly as experiment.train lays it "
                         'out (dist.virtual_layout: R // N ranks per process, one more on the first R %% N -- 19 on 8 GPUs = '
                         '3 3 3 2 2 2 2 2): `--gpus 8 --num-cpu 19` is the published job (readme.md:16) on one node')
    ap.add_argument('--structure', default='curious', choices=['curious', 'task_experts'],
                    help="diagnostic: 'task_experts' = BASELINE configs[4], every expert updated in one batched launch "
                         "sequence per update (curious_ddpg_update_experts)")
    ap.add_argument('--cpu-ranks', type=int, default=-1,
                    help='ranks of the multi-process CPU baseline (default min(19, host cores); 0 = skip)')
    ap.add_argument('--cpu-rank-worker', type=int, nargs=4, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--ipc-probe', type=int, default=0, help=argparse.SUPPRESS)     # child mode of ipc_probe()
    ap.add_argument('--no-ipc-probe', action='store_true',
                    help='several ranks: skip the side measurement of the fused IPC all-reduce + Adam path')
    args = ap.parse_args()
    if args.num_cpu and args.virtual_ranks > 1:
        ap.error('--num-cpu R names the ranks of the whole job, --virtual-ranks V the ranks per GPU: give one of them')
    if args.num_cpu and args.num_cpu < args.gpus:
        ap.error('--num-cpu %d ranks on --gpus %d processes' % (args.num_cpu, args.gpus))
    if args.rollout_batch_size is None:
        args.rollout_batch_size = 2 if (args.virtual_ranks > 1 or args.num_cpu > args.gpus) else B_R
    return args


def maybe_relaunch(args):
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(29500 + os.getpid() % 1000), __file__] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))


def build_job(use_graph, seed=0, env=None, b_r=B_R, virtual_ranks=1, rank_base=None, total_ranks=None):
    from curious_amd import dist, logger
    from curious_amd.experiment import config
    from curious_amd.rollout import RolloutWorker
    params = dict(config.MULTI_TASK_PARAMS)
    params.update(env_name=env or ENV, task_selection='active_competence_progress', goal_selection='random',
                  task_replay='replay_task_cp_buffer', goal_replay='her', structure='curious', normalize_obs=False,
                  num_cpu=dist.world_size(), clip_return=1, trial_id=0, seed=seed, rollout_batch_size=b_r,
                  n_batches=N_BATCHES, batch_size=BATCH, rng_mode='device', use_graph=use_graph,
                  async_store=os.environ.get('CURIOUS_ASYNC_STORE', '1') != '0', virtual_ranks=virtual_ranks)
    base = dist.rank() * virtual_ranks if rank_base is None else rank_base
    if rank_base is not None:                                     # an uneven layout (--num-cpu): experiment/train.py:268-270
        params['rank_base'], params['total_ranks'] = rank_base, total_ranks
    params = config.prepare_params(params)
    params['ddpg_params']['normalize_obs'] = False
    params['ddpg_params']['seed'] = seed
    dims = config.configure_dims(params)
    buffers = config.configure_buffer(dims=dims, params=params)
    policy = config.configure_ddpg(dims=dims, params=params, buffers=buffers, clip_return=True)
    worker = RolloutWorker(params['make_env'], policy, dims, logger, T=params['T'], rollout_batch_size=b_r,
                           exploit=False, use_target_net=False, compute_Q=False, noise_eps=params['noise_eps'],
                           random_eps=params['random_eps'], structure='curious',
                           task_selection='active_competence_progress', goal_selection='random',
                           queue_length=params['queue_length'], eval=False)
    worker.seed(seed + 1000000 * base)
    if virtual_ranks > 1:
        worker.seed_ranks([seed + 1000000 * (base + v) for v in range(virtual_ranks)])
    return params, dims, policy, worker


def build_experts_job(use_graph, seed=0, env=None, b_r=B_R, virtual_ranks=1):
    """BASELINE configs[4]: one expert per task on shared per-task buffers (train.py:285-291), all updated together."""
    from curious_amd import dist, logger
    from curious_amd.experiment import config
    from curious_amd.experts import ExpertBank
    from curious_amd.rollout import RolloutWorker
    params = dict(config.MULTI_TASK_PARAMS)
    params.update(env_name=env or ENV, task_selection='random', goal_selection='random',
                  task_replay='replay_current_task_buffer', goal_replay='her', structure='task_experts',
                  normalize_obs=False, num_cpu=dist.world_size(), clip_return=1, trial_id=0, seed=seed,
                  rollout_batch_size=b_r, n_batches=N_BATCHES, batch_size=BATCH, rng_mode='device', use_graph=use_graph,
                  virtual_ranks=virtual_ranks)
    params = config.prepare_params(params)
    params['ddpg_params']['normalize_obs'] = False
    params['ddpg_params']['seed'] = seed
    dims = config.configure_dims(params)
    buffers = config.configure_buffer(dims=dims, params=params)
    bank = ExpertBank(lambda i, **hooks: config.configure_ddpg(dims=dims, params=params, buffers=buffers,
                                                               clip_return=True, t_id=i, **hooks), params['nb_tasks'])
    workers = [RolloutWorker(params['make_env'], bank[i], dims, logger, T=params['T'], rollout_batch_size=b_r,
                             exploit=False, use_target_net=False, compute_Q=False, noise_eps=params['noise_eps'],
                             random_eps=params['random_eps'], structure='task_experts', task_selection='random',
                             goal_selection='random', queue_length=params['queue_length'], eval=False, unique_task=i)
               for i in range(params['nb_tasks'])]
    for i, w in enumerate(workers):
        w.seed(seed + 1000000 * dist.rank() * virtual_ranks + i)
        if virtual_ranks > 1:
            w.seed_ranks([seed + 1000000 * (dist.rank() * virtual_ranks + v) + i for v in range(virtual_ranks)])
    return params, dims, bank, workers


def experts_cycle(bank, workers, k):
    """One cycle of train.py:96-103 for the expert whose turn it is, with the update of ALL experts batched."""
    i = k % len(workers)
    episode, cp, n_ep = workers[i].generate_rollouts()
    bank[i].store_episode(episode, cp, n_ep)
    bank.train_batches(N_BATCHES)
    bank.update_target_net()


def prefill(policy, n_eps, seed):
    """Pre-load every per-task buffer with synthetic episodes (SURVEY 8d cfg 2): float32 random walks, ag = o[:AG],
    one-hot task constant per episode, goal on the task's slots, u ~ U(-1,1), change / success flags."""
    import torch
    lay = policy._layout
    T, O, AG, N = lay.T, lay.dims['o'], lay.dims['ag'], lay.dims['task_descr']
    gen = torch.Generator(device='cuda')
    gen.manual_seed(seed)
    seen = set()
    for buf in [bl[i] for bl in policy._rank_buffers for i in range(1, policy.nb_tasks + 1)]:   # (every virtual rank's)
        if id(buf) in seen:
            continue
        seen.add(id(buf))
        E = min(n_eps, buf.size)
        rec = buf.records[:E]
        v = lay.record_views(rec)
        o0 = torch.randn([E, 1, O], device='cuda', generator=gen)
        steps = 0.01 * torch.randn([E, T, O], device='cuda', generator=gen)
        o = torch.cat([o0, o0 + torch.cumsum(steps, dim=1)], dim=1)
        rec.zero_()
        v['o'].copy_(o)
        v['ag'].copy_(o[:, :, :AG])
        task = torch.randint(0, N, [E], device='cuda', generator=gen)
        td = torch.nn.functional.one_hot(task, N).float()
        v['task_descr'].copy_(td[:, None, :].expand(E, T, N))
        g = torch.zeros([E, T, AG], device='cuda')
        for j in range(N):
            sel = task == j
            if sel.any():
                g[sel, :, 3 * j:3 * j + 3] = (o[sel, T // 2, 3 * j:3 * j + 3]
                                              + 0.03 * torch.randn([int(sel.sum()), 3], device='cuda',
                                                                   generator=gen))[:, None, :]
        v['g'].copy_(g)
        v['u'].copy_(torch.rand([E, T, lay.dims['u']], device='cuda', generator=gen) * 2 - 1)
        v['change'].copy_(((o[:, :1, :AG] - o[:, 1:, :AG]).abs() > 1e-3).float())
        buf.current_size = E
        buf.n_transitions_stored = E * T
    policy._tables_dirty = True


def cycle(policy, worker):
    episode, cp, n_ep = worker.generate_rollouts()
    policy.store_episode(episode, cp, n_ep)
    policy.train_batches(N_BATCHES)                               # = for _ in range(N_BATCHES): policy.train()
    policy.update_target_net()


def kernel_flops_bytes(policy, lay, B_R=B_R, n_experts=1):
    """ALGORITHMIC work of one launch of each kernel class in one update / one rollout step."""
    c = policy
    B, H, nl, U = c._Bt, c.hidden, c.layers, c.dimu               # (virtual ranks: V minibatches of 256 per launch)
    O, G, N = c.dimo, c.dimg, c.dimtd
    Ka, Kc = O + N + G, O + N + U + G
    hid = nl - 1
    her_bytes_per_transition = ((2 * O + U + G + 2 * c.dimag + N + 0.8 * c.dimag)
                                + (c.dimag + G + O + N + U + O + G + 1)) * 4          # 1 034 B at Arm4
    adam_bytes = 28 * (c.P_Q + c.P_pi)
    Sa, Sc = O + N, O + N + U
    net = lambda S, D: (S + G) * H + hid * H * H + H * D          # multiply-adds of one forward pass per row
    rows_fwd = 2 * net(Sa, U) + 3 * net(Sc, 1)                    # target actor, actor; target critic, critic(u), critic(pi)
    rows_bwd = (hid * H * H + H) + (hid * H * H + H + U * H) + (hid * H * H + H * U)
    # bytes of weights the row-local update pulls L2 -> CU per launch: every workgroup (4 batch rows, one of three kinds)
    # streams every matrix of its chain once (THIS is what bounds a layer, not MFMA and not HBM)
    l0pi, l0q, hidw = (Sa + G) * H, (Sc + G) * H, hid * H * H
    w_actor = l0pi + hidw + H * U + l0q + hidw + H + hidw + U * H + H * U + hidw
    w_target = l0pi + hidw + H * U + l0q + hidw + H
    w_critic = l0q + hidw + 2 * H + hidw
    # (4 batch rows per workgroup; 8 from 768 rows on -- csrc/mlp_rows.h ROWS_R2: half the stream per row)
    rows_per_wg = 8 if (B * n_experts >= 768 and B % 32 == 0 and os.environ.get('CURIOUS_ROWS8', '1') != '0') else 4
    # (16 from `rows16` rows on, single agents: csrc/mlp_rows16.h -- the waves split the output columns)
    from curious_amd import ops as _ops
    r16 = _ops.get_option('rows16')
    if n_experts == 1 and r16 > 0 and B >= r16 and B % 64 == 0:
        rows_per_wg = 16
    rows_l2_bytes = 4 * (w_actor + w_target + w_critic) * (B // rows_per_wg)
    return dict(
        # the row-local routes (one launch per update / per env step; curious_amd/csrc/mlp_rows*.h)
        ddpg_rows_kernel=dict(bound='mfma', per_update=2 * B * (rows_fwd + rows_bwd), launches_update=1,
                              l2_stream_bytes=rows_l2_bytes, critical_cu_bytes=4 * w_actor, rows_per_workgroup=rows_per_wg),
        ddpg_rows_her_kernel=dict(bound='mfma', per_update=2 * B * (rows_fwd + rows_bwd), launches_update=1,
                                  l2_stream_bytes=rows_l2_bytes, critical_cu_bytes=4 * w_actor,
                                  rows_per_workgroup=rows_per_wg),
        policy_rows_kernel=dict(bound='mfma', per_update=0, launches_update=0, per_env_step=2 * B_R * net(Sa, U),
                                launches_env_step=1),
        # the weights-resident rollout: layer 0 is computed by all 4 members of a group (x 4), the rest once
        policy_resident_kernel=dict(bound='mfma', per_update=0, launches_update=0,
                                    per_env_step=2 * B_R * (net(Sa, U) + 3 * (Sa + G) * H), launches_env_step=1),
        # layers 0 + 1 in one launch: level A's 3 chains (+ the 2 action-free pre-activations of level B) per update;
        # the actor chain per env step
        fwd_l01_kernel=dict(bound='mfma', per_update=3 * 2 * B * H * H + 2 * B * H * (Kc + 4 * Ka), launches_update=1,
                            per_env_step=2 * B_R * H * H + 2 * B_R * H * Ka, launches_env_step=1),
        # remaining hidden layers (256^3 GEMMs): level A layers 2.. (3 chains), level B layers 2.. (2 chains)
        fwd_hot_kernel=dict(bound='mfma', per_update=5 * (hid - 1) * 2 * B * H * H, launches_update=2 * (hid - 1),
                            per_env_step=(hid - 1) * 2 * B_R * H * H, launches_env_step=hid - 1),
        # level B layer 1 with the actor heads and the action rows of layer 0 in its prologue
        fwd_pi_kernel=dict(bound='mfma', per_update=2 * (2 * B * H * H + 4 * B * H * U), launches_update=1),
        dx_hot_kernel=dict(bound='mfma', per_update=(3 * hid - 3) * 2 * B * H * H, launches_update=2 * hid - 2),
        dx_crit_kernel=dict(bound='mfma', per_update=2 * 2 * B * H * H + 3 * 2 * B * H, launches_update=1),
        dx_actor_kernel=dict(bound='mfma', per_update=2 * B * H * H + 4 * B * H * U, launches_update=1),
        # every weight gradient + Adam (28 B/param) + the next HER gather in one launch; priced on its MFMA work
        dw_adam_her_kernel=dict(bound='mfma', per_update=2 * hid * 2 * B * H * H + 2 * B * H * (Kc + Ka)
                                + 2 * B * H * (1 + U), launches_update=1,
                                hbm_bytes_per_update=adam_bytes + her_bytes_per_transition * B),
        her_sample_kernel=dict(bound='hbm', per_update=0, launches_update=0),
    )


def profile_pass(policy, worker, n_cycles=1, bank=None, step=None):
    """Per-kernel launch durations from HIP events recorded on the launch stream.

    Events cannot sit inside a replayed hipGraph, so the same cycle is replayed with eager launches, every kernel
    launch bracketed by an event pair (curious_prof_*).  A bracket costs a fixed amount of stream time itself; it is
    calibrated in the same run as  c = (sum of the bracketed durations of a 100-update block - one event pair around
    the identical block without brackets) / number of launches  and subtracted, so that the per-kernel averages add
    up to the real duration of the sequence (the quantity rocprofv3 --kernel-trace reports)."""
    from curious_amd import ops
    import torch
    # (batched experts: the bank's launches; its experts' rollouts are eager as well)
    graphed = [policy] if bank is None else [bank] + list(bank)
    saved = [g.use_graph for g in graphed]
    for g in graphed:
        g.use_graph = False

    def updates():
        (policy if bank is None else bank).train_batches(N_BATCHES)
    updates()                                                     # warm the eager path
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    updates()
    e1.record()
    torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1)                                 # ms, no brackets
    ops.prof_collect()
    ops.prof_enable(True)
    updates()
    ops.prof_enable(False)
    st = ops.prof_collect()
    n_launch = sum(v[0] for v in st.values())
    overhead_ms = max(0.0, (sum(v[1] for v in st.values()) - t_plain) / max(1, n_launch))
    ops.prof_enable(True)
    for _ in range(n_cycles):
        if step is not None:
            step()
        else:
            cycle(policy, worker)
    ops.prof_enable(False)
    stats = ops.prof_collect()
    for g, s in zip(graphed, saved):
        g.use_graph = s
    return stats, overhead_ms, t_plain / N_BATCHES


def pmc_traffic_file(virtual_ranks=1):
    """The committed PMC passes of this command: per round; a --virtual-ranks V line has passes of its own or none."""
    suffix = '' if virtual_ranks == 1 else '_virtual_ranks_%d' % virtual_ranks
    return next((p for p in (os.path.join(ROOT, 'profiles', 'r%02d_pmc_hbm_traffic%s.json' % (r, suffix))
                             for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p)), '')


def pmc_traffic(kernel, virtual_ranks=1):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r01_pmc_hbm_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs).  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half
    the bytes of 16-byte-per-lane streaming reads -> doubled; WRITE_SIZE is exact for 16-byte stores.  None if absent."""
    path = pmc_traffic_file(virtual_ranks)
    try:
        with open(path) as f:
            table = json.load(f)
        tot = {'FETCH_SIZE': [0, 0.0], 'WRITE_SIZE': [0, 0.0]}
        # (batches of >= 1 280 rows run the 16-row form of the row-local kernels under the same profile id: mlp_rows16.h)
        names = (kernel, kernel.replace('ddpg_rows_', 'ddpg_rows16_'))
        for name, counters in table.items():                      # template instances: fwd_hot_kernel<true>, <false>
            if any(name == k or name.startswith(k + '<') for k in names):
                for c in tot:
                    tot[c][0] += counters[c]['launches']
                    tot[c][1] += counters[c]['launches'] * counters[c]['avg_KB']
        if not tot['FETCH_SIZE'][0] or not tot['WRITE_SIZE'][0]:
            return None
        return int(1024 * (2 * tot['FETCH_SIZE'][1] / tot['FETCH_SIZE'][0] + tot['WRITE_SIZE'][1] / tot['WRITE_SIZE'][0]))
    except Exception:
        return None


def roofline(policy, worker, stats, n_cycles, overhead_ms, n_experts=1):
    """n_experts: batched task experts -- the update kernels carry the expert on grid.z / grid.y, one launch does the work of
    all of them (the acting kernels run one expert's rollout per cycle)."""
    work = kernel_flops_bytes(policy, policy._layout, worker.rollout_batch_size, n_experts)
    if n_experts > 1:
        for w_ in work.values():
            for key in ('per_update', 'l2_stream_bytes', 'hbm_bytes_per_update'):
                if key in w_:
                    w_[key] *= n_experts
    cal = {k: (v[0], max(v[1] - v[0] * overhead_ms, 0.0)) for k, v in stats.items() if v[0] > 0}
    if not cal:
        return None, {}
    dominant = max(cal, key=lambda k: cal[k][1])
    table = {k: dict(launches=v[0], total_ms=round(v[1], 4), avg_us=round(1e3 * v[1] / v[0], 3),
                     avg_us_bracketed=round(1e3 * stats[k][1] / v[0], 3)) for k, v in cal.items()}
    w = work.get(dominant)
    if w is None:
        return dict(kernel=dominant, bound='hbm', achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None,
                    traffic=None), table
    launches, ms = cal[dominant]
    units = n_cycles * N_BATCHES * w['per_update']
    if 'per_env_step' in w:
        units += n_cycles * worker.T * w['per_env_step']
    per_launch = units / launches
    avg_s = ms * 1e-3 / launches
    if w['bound'] == 'mfma':
        ach, peak, unit = per_launch / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, 'TFLOP/s'
    else:
        ach, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, 'GB/s'
    out = dict(kernel=dominant, bound=w['bound'], achieved=round(ach, 4), peak=peak, unit=unit,
               frac=round(ach / peak, 5), traffic=pmc_traffic(dominant, getattr(policy, 'V', 1)),
               # PMC counters cannot be read from inside the process: the figure comes from the committed rocprofv3 --pmc
               # passes of this same command (FETCH_SIZE x 2 + WRITE_SIZE per launch), not from this run
               traffic_source='%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, committed; not '
                              'measured in this run)' % os.path.relpath(pmc_traffic_file(getattr(policy, 'V', 1)) or 'none', ROOT),
               algorithmic_per_launch=round(per_launch, 1),
               avg_launch_us=round(avg_s * 1e6, 3), event_bracket_overhead_us=round(overhead_ms * 1e3, 3))
    if 'l2_stream_bytes' in w:
        # the resource that actually bounds this kernel: weight bytes streamed L2 -> CU (each CU's fill path), against the
        # aggregate L2 bandwidth of MI355X_MICROARCH.md (34.5 TB/s over 256 CUs; the launch occupies 192 of them)
        tb = w['l2_stream_bytes'] / avg_s / 1e12
        out['l2_stream'] = dict(achieved=round(tb, 3), peak=L2_PEAK_TBS, unit='TB/s', frac=round(tb / L2_PEAK_TBS, 4),
                                bytes_per_launch=int(w['l2_stream_bytes']), rows_per_workgroup=w.get('rows_per_workgroup', 4))
        # ... and the CU on the critical path: an actor-side workgroup pulls 12 network passes of weights through ITS fill
        # path (MI355X_MICROARCH.md: 64 B/clk = 135 GB/s per CU), one layer after the other, for the whole launch
        # (one chain per CU: a single rank's batch; with several ranks a CU runs many chains one after / beside the other)
        if getattr(policy, 'V', 1) == 1:
            gb = w['critical_cu_bytes'] / avg_s / 1e9
            out['l2_stream']['critical_cu'] = dict(achieved=round(gb, 1), peak=L2_PER_CU_GBS, unit='GB/s',
                                                   frac=round(gb / L2_PER_CU_GBS, 4),
                                                   bytes_per_launch=int(w['critical_cu_bytes']))
    return out, table


def cpu_baseline(seed=0, budget_s=20.0):
    """The oracle (NumPy port of the reference's CPU path) on ONE host core, same workload: cycles of
    256-env rollout + store + 100 updates.  A bounded sample (whole cycles until ~budget_s)."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.env import SyntheticMultiTaskArm
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    nb, dimo, T = 4, 40, 50
    G = 3 * nb
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb)]
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    np.random.seed(seed)
    sampler = oher.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', make_reward_fun(ids, ids),
                                                          tasks_ag_id=ids, tasks_g_id=ids)
    cap = 4096 * T        # smaller capacity than the GPU run: only memory, not arithmetic, depends on it
    bufs = [OBuf(shapes, cap, T, sampler) for _ in range(nb + 1)]
    agent = OracleDDPG(dims, T, bufs, sampler, ids, ids, batch_size=BATCH, weight_rng=np.random.RandomState(seed))
    envs = [SyntheticMultiTaskArm(nb, dimo, T, seed=seed, env_id=i) for i in range(B_R)]
    n_rollout_envs = B_R  # the whole 256-env rollout is timed (round 2 sampled 32 envs and scaled)

    def rollout():
        tasks = np.random.choice(range(nb), size=B_R)
        goals = np.random.uniform(-1, 1, (B_R, 3)).astype(np.float32)
        obs = []
        for i, e in enumerate(envs[:n_rollout_envs]):
            e.reset()
            obs.append(e.reset_task_goal(goals[i], int(tasks[i])))
        o = np.stack([x['observation'] for x in obs])
        g = np.stack([x['desired_goal'] for x in obs])
        td = np.stack([x['mask'] for x in obs])
        ep = dict(o=[o.copy()], ag=[o[:, :G].copy()], u=[], g=[], task_descr=[], change=[], info_is_success=[])
        ag0 = o[:, :G].copy()
        for t in range(T):
            u = agent.get_actions(o, o[:, :G], g, task_descr=td, noise_eps=0.2, random_eps=0.3)
            res = [e.step(u[i]) for i, e in enumerate(envs[:n_rollout_envs])]
            o = np.stack([r[0]['observation'] for r in res])
            ep['u'].append(u.astype(np.float32)); ep['g'].append(g.copy()); ep['task_descr'].append(td.copy())
            ep['change'].append(np.abs(ag0 - o[:, :G]) > 1e-3)
            ep['info_is_success'].append(np.array([[r[3]['is_success']] for r in res], np.float32))
            ep['o'].append(o.copy()); ep['ag'].append(o[:, :G].copy())
        return {k: np.array(v).swapaxes(0, 1) for k, v in ep.items()}

    with threadpool_limits(limits=1):
        t_roll = t_store = t_train = 0.0
        cycles = 0
        t_start = time.time()
        while time.time() - t_start < budget_s or cycles < 1:
            t0 = time.time()
            ep = rollout()
            t1 = time.time()
            # tile the 32 sampled episodes to the 256-episode batch of the workload before storing
            rep = B_R // n_rollout_envs
            ep = {k: np.concatenate([v] * rep, axis=0) for k, v in ep.items()}
            # make sure at least one task counts as active so that the buffers fill (synthetic env: Reach moves)
            agent.store_episode({k: v.astype(np.float64) for k, v in ep.items()}, np.zeros(nb), 0)
            t2 = time.time()
            for _ in range(N_BATCHES):
                agent.train()
            agent.update_target_net()
            t3 = time.time()
            t_roll += (t1 - t0) * rep
            t_store += t2 - t1
            t_train += t3 - t2
            cycles += 1
        total = t_roll + t_store + t_train
    return dict(value=round(cycles * N_BATCHES * BATCH / total, 1), unit='HER grad transitions/s',
                env_steps_per_sec=round(cycles * B_R * T / total, 1), cores=1, kind='port',
                sample='%d full cycles of the NumPy oracle (256-env x 50-step rollout + store + 100 updates, batch 256) '
                       'on 1 thread; %.1f s rollout, %.1f s store, %.1f s updates' % (cycles, t_roll, t_store, t_train))


def cpu_rank_worker(rank, world, port, budget_s):
    """One rank of the multi-process CPU baseline (SURVEY 8d ii): the reference's own regime -- one single-threaded
    process per rank (util.py:155-160), rollout_batch_size 2 and batch 256 per rank (config.py:70-73), gradients SUMMED
    over ranks before Adam (mpi_adam.py:26-28), normaliser sums averaged over ranks (normalizer.py:84-94) -- with gloo
    standing in for MPI.  Prints one JSON line (cycles, seconds)."""
    os.environ.update(OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    import torch
    import torch.distributed as td
    from oracle import her as oher
    from oracle.ddpg import OracleDDPG
    from oracle.env import SyntheticMultiTaskArm
    from oracle.replay_buffer import ReplayBuffer as OBuf
    from oracle.reward import make_reward_fun
    torch.set_num_threads(1)
    td.init_process_group('gloo', rank=rank, world_size=world)

    def allreduce_sum(x):
        t = torch.from_numpy(np.ascontiguousarray(x))
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t.numpy()

    nb, dimo, T, b_r = 4, 40, 50, 2
    G = 3 * nb
    ids = [[3 * j, 3 * j + 1, 3 * j + 2] for j in range(nb)]
    dims = dict(o=dimo, u=4, g=G, ag=G, task_descr=nb, info_is_success=1)
    shapes = dict(o=(T + 1, dimo), u=(T, 4), g=(T, G), ag=(T + 1, G), info_is_success=(T, 1), task_descr=(T, nb),
                  change=(T, G))
    np.random.seed(1000000 * rank)                                  # train.py:242
    sampler = oher.make_sample_multi_task_her_transitions('her', 4, 'replay_task_cp_buffer', make_reward_fun(ids, ids),
                                                          tasks_ag_id=ids, tasks_g_id=ids)
    bufs = [OBuf(shapes, 2048 * T, T, sampler) for _ in range(nb + 1)]
    agent = OracleDDPG(dims, T, bufs, sampler, ids, ids, batch_size=BATCH, weight_rng=np.random.RandomState(0),
                       allreduce_sum=allreduce_sum, comm_size=world)
    envs = [SyntheticMultiTaskArm(nb, dimo, T, seed=0, env_id=rank * b_r + i) for i in range(b_r)]

    def rollout():
        tasks = np.random.choice(range(nb), size=b_r)
        goals = np.random.uniform(-1, 1, (b_r, 3)).astype(np.float32)
        obs = []
        for i, e in enumerate(envs):
            e.reset()
            obs.append(e.reset_task_goal(goals[i], int(tasks[i])))
        o = np.stack([x['observation'] for x in obs])
        g = np.stack([x['desired_goal'] for x in obs])
        tdm = np.stack([x['mask'] for x in obs])
        ep = dict(o=[o.copy()], ag=[o[:, :G].copy()], u=[], g=[], task_descr=[], change=[], info_is_success=[])
        ag0 = o[:, :G].copy()
        for t in range(T):
            u = agent.get_actions(o, o[:, :G], g, task_descr=tdm, noise_eps=0.2, random_eps=0.3).reshape(b_r, 4)
            res = [e.step(u[i]) for i, e in enumerate(envs)]
            o = np.stack([r[0]['observation'] for r in res])
            ep['u'].append(u.astype(np.float32)); ep['g'].append(g.copy()); ep['task_descr'].append(tdm.copy())
            ep['change'].append(np.abs(ag0 - o[:, :G]) > 1e-3)
            ep['info_is_success'].append(np.array([[r[3]['is_success']] for r in res], np.float32))
            ep['o'].append(o.copy()); ep['ag'].append(o[:, :G].copy())
        return {k: np.array(v).swapaxes(0, 1) for k, v in ep.items()}

    # a few stored episodes per rank before the clock starts, so that every per-task buffer can be sampled
    for _ in range(3):
        agent.store_episode({k: v.astype(np.float64) for k, v in rollout().items()}, np.zeros(nb), 0)
    td.barrier()
    t0 = time.time()
    cycles = 0
    go = torch.ones(1)
    while True:
        ep = rollout()
        agent.store_episode({k: v.astype(np.float64) for k, v in ep.items()}, np.zeros(nb), 0)
        for _ in range(N_BATCHES):
            agent.train()
        agent.update_target_net()
        cycles += 1
        go[0] = 1.0 if time.time() - t0 < budget_s else 0.0       # rank 0 decides, everybody stops together
        td.broadcast(go, src=0)
        if go[0] == 0:
            break
    td.barrier()
    elapsed = time.time() - t0
    if rank == 0:
        print(json.dumps(dict(cycles=cycles, seconds=elapsed)), flush=True)
    td.destroy_process_group()


def cpu_baseline_ranks(n_ranks, budget_s=12.0):
    """R single-threaded oracle processes with summed gradients (the reference's 19-rank MPI regime, readme.md:16), as
    child processes of this script; returns the aggregate rates."""
    port = 29700 + os.getpid() % 200
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-rank-worker', str(r), str(n_ranks),
                               str(port), str(int(budget_s))], stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1',
                                                                  CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES=''))
             for r in range(n_ranks)]
    try:
        out, _ = procs[0].communicate(timeout=budget_s * 6 + 120)
        for p in procs[1:]:
            p.wait(timeout=60)
        rec = json.loads(out.decode().strip().splitlines()[-1])
    except Exception as err:                                        # a baseline leg must never sink the bench line
        for p in procs:
            if p.poll() is None:
                p.kill()
        return dict(error='%s: %s' % (type(err).__name__, err), cores=n_ranks)
    total = rec['seconds']
    return dict(value=round(n_ranks * rec['cycles'] * N_BATCHES * BATCH / total, 1), unit='HER grad transitions/s',
                env_steps_per_sec=round(n_ranks * rec['cycles'] * 2 * 50 / total, 1), cores=n_ranks, kind='port',
                sample='%d cycles per rank of the NumPy oracle on %d single-threaded processes (per rank: 2 rollouts x '
                       'T=50, 100 updates of batch 256; gradients summed and normaliser sums averaged over ranks with gloo '
                       'in place of MPI), %.1f s' % (rec['cycles'], n_ranks, total))


def collective_report(policy, policies):
    """What a several-rank line needs to be read on its own (SCALE runs): the gradient all-reduce timed alone (a chain
    of 100 all-reduces of the fused [P] float32 vector, replayed from one hipGraph when the update graphs capture their
    collective too, eager otherwise), the communicator's settings as this process saw them, and the agreement of the
    replicas (128-bit parameter checksums of all ranks, mpi_adam.py:42-50)."""
    import torch
    import torch.distributed as td
    from curious_amd import dist
    world, backend = dist.world_size(), td.get_backend()
    captured = bool(dist.captured_allreduce_ok())
    n_chain = 100
    buf = torch.zeros_like(policy.grad)
    torch.cuda.synchronize()
    dist.barrier()

    def chain():
        for _ in range(n_chain):
            td.all_reduce(buf)
    graph = None
    if captured:
        chain()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            chain()
    run = graph.replay if graph is not None else chain
    run()                                                          # warm
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_us = e0.elapsed_time(e1) * 1e3 / (reps * n_chain)
    t = torch.tensor([dev_us, wall * 1e6 / (reps * n_chain)], dtype=torch.float64, device='cuda')
    if world > 1:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    del graph
    # replicas: every rank's checksum of every policy, gathered
    sums = []
    for pol in policies:
        pol._check_synced(wait=True)                               # raises RankDivergence on a mismatch with rank 0
        sums.append(pol._sync_buf[0].clone())
    mine = torch.stack(sums).reshape(1, -1)
    allsums = dist.allgather(mine) if world > 1 else mine
    identical = bool((allsums == allsums[:1]).all())
    try:
        ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return dict(allreduce_us=round(float(t[0]), 3), allreduce_host_us=round(float(t[1]), 3),
                allreduce_bytes=int(buf.numel() * 4), allreduce_chain=n_chain,
                rccl=dict(backend=backend, world=world, captured=captured, version=ver,
                          env={k: os.environ.get(k) for k in ('NCCL_PROTO', 'NCCL_ALGO', 'NCCL_MIN_NCHANNELS',
                                                              'NCCL_MAX_NCHANNELS', 'RCCL_MSCCL_ENABLE',
                                                              'HSA_ENABLE_IPC_MODE_LEGACY', 'CURIOUS_GRAPH_ALLREDUCE',
                                                              'CURIOUS_ALLREDUCE')}),
                replicas_identical=identical, replica_checksums=[[int(x) for x in row] for row in allsums.cpu()])


def ipc_probe_child(args):
    """Child mode (bench.py --ipc-probe K, one child per rank, its own process group on MASTER_PORT): K timed cycles of the
    same workload with DDPG(_allreduce='ipc') -- the fused reduce-scatter + Adam + all-gather kernel over peer-mapped
    buffers (csrc/ipc.hip) in place of the RCCL all-reduce + optimiser launch.  Rank 0 prints one JSON line."""
    import numpy as np
    import torch
    from curious_amd import dist
    os.environ['CURIOUS_ALLREDUCE'] = 'ipc'
    dist.init_from_env()
    rank, world = dist.rank(), dist.world_size()
    torch.cuda.set_device(dist.local_device_index())
    np.random.seed(1234 + 1000000 * rank)
    params, dims, policy, worker = build_job(use_graph=not args.no_graph, env=args.env, b_r=args.rollout_batch_size)
    prefill(policy, args.prefill, seed=rank)
    for _ in range(2):
        cycle(policy, worker)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.ipc_probe):
        cycle(policy, worker)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    err = int(policy._ipc['words'][1]) if getattr(policy, '_ipc', None) else -1
    policy._check_synced(wait=True)
    if rank == 0:
        print(json.dumps(dict(ms_per_step=round(1e3 * elapsed / args.ipc_probe, 4), steps=args.ipc_probe, world=world,
                              wait_gave_up=err, replicas_identical=True)), flush=True)
    teardown([policy])


def ipc_probe(args, steps=10, timeout=100.0):
    """Several ranks: how the same cycle runs with the fused IPC all-reduce + Adam kernel instead of RCCL + the optimiser
    launch -- the only place this path can meet real xGMI links is the driver's multi-GPU run.  Measured in CHILD
    processes (one per rank, on the rank's GPU, while the parents idle), so that whatever goes wrong there -- a mapping
    that fails, a wait that never ends -- cannot take the bench line with it: a failure is reported as {"error": ...}."""
    port = int(os.environ.get('MASTER_PORT', '29500')) + 23
    cmd = [sys.executable, os.path.abspath(__file__), '--ipc-probe', str(steps), '--gpus', str(args.gpus),
           '--env', args.env, '--rollout-batch-size', str(args.rollout_batch_size), '--prefill', str(args.prefill)]
    if args.no_graph:
        cmd.append('--no-graph')
    env = dict(os.environ, MASTER_PORT=str(port), CURIOUS_ALLREDUCE='ipc')
    for k in list(env):                                            # the children rendezvous among themselves (rank 0's child
        if k.startswith('TORCHELASTIC_'):                          # hosts the store), not through the launcher's agent store
            del env[k]
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        try:
            out, errtxt = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            return dict(error='timed out after %.0f s' % timeout)
        if p.returncode != 0:
            return dict(error='child exited with %d: %s' % (p.returncode, errtxt.decode(errors='replace')[-300:]))
        lines = [ln for ln in out.decode().splitlines() if ln.startswith('{')]
        return json.loads(lines[-1]) if lines else dict(ok=True)
    except Exception as err:                                       # a side measurement must never sink the bench line
        return dict(error='%s: %s' % (type(err).__name__, err))


def regime_line(args, ranks=19, steps=10, timeout=150.0):
    """The default run's diagnostic: the reference's PUBLISHED regime (readme.md:16, `--num_cpu 19`) as 19 virtual ranks on
    this GPU -- the one configuration in which the chip is loaded -- measured in a child process behind the timed region
    (`bench.py --virtual-ranks 19`, a few cycles), so that the driver's own record carries it next to the headline.  A
    failure there is reported, it does not take the line with it."""
    cmd = [sys.executable, os.path.abspath(__file__), '--virtual-ranks', str(ranks), '--steps', str(steps), '--warmup', '3',
           '--no-cpu-baseline', '--no-regime-line']
    if args.no_graph:
        cmd.append('--no-graph')
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC_')}
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        try:
            out, errtxt = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            return dict(error='timed out after %.0f s' % timeout)
        if p.returncode != 0:
            return dict(error='child exited with %d: %s' % (p.returncode, errtxt.decode(errors='replace')[-300:]))
        d = json.loads([ln for ln in out.decode().splitlines() if ln.startswith('{')][-1])
        k = d.get('kernels', {})
        return {'command': 'python bench.py --virtual-ranks %d --steps %d --warmup 3 --no-cpu-baseline' % (ranks, steps),
                'what': "the reference's --num_cpu %d job on ONE GPU: %d private buffer sets / seeds / rollout groups, every "
                        "update = %d minibatches of 256 in one launch sequence, gradients summed (diagnostic, not the headline)"
                        % (ranks, ranks, ranks),
                'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'ranks': ranks,
                'roofline': {key: d['roofline'].get(key) for key in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                                     'avg_launch_us')},
                'kernels_avg_us': {name: round(v['avg_us'], 2) for name, v in k.items()
                                   if name in ('ddpg_rows_kernel', 'dw_adam_her_kernel', 'policy_resident_kernel')}}
    except Exception as err:                                       # a side measurement must never sink the bench line
        return dict(error='%s: %s' % (type(err).__name__, err))


def teardown(policies, bank=None):
    """Captured graphs hold the communicator's streams: drop them before the process group goes, then leave through the
    normal interpreter exit (curious_amd.experiment.train.shutdown)."""
    from curious_amd.experiment.train import shutdown
    shutdown(policies, bank)


def main():
    args = parse()
    if args.cpu_rank_worker is not None:
        cpu_rank_worker(*args.cpu_rank_worker)
        return
    if args.ipc_probe:
        ipc_probe_child(args)
        return
    maybe_relaunch(args)
    # the CPU legs run first, before this process touches the GPU (rank 0 of a one-GPU run only)
    cpu = None
    V = args.virtual_ranks if not (args.num_cpu and args.gpus == 1) else args.num_cpu
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        n_ranks = min(19, os.cpu_count() or 1) if args.cpu_ranks < 0 else args.cpu_ranks
        if V > 1:
            # the SAME job on the host cores: R single-threaded oracle processes, 2 rollouts and batch 256 per rank,
            # gradients summed over ranks -- as many ranks as the box has cores for (stated in `cores`)
            cpu = cpu_baseline_ranks(max(2, n_ranks), budget_s=20.0)
        else:
            cpu = cpu_baseline()
            if n_ranks >= 2:
                cpu['ranks'] = cpu_baseline_ranks(n_ranks)
    import numpy as np
    import torch
    from curious_amd import dist, ops
    from curious_amd.util import freeze_setup_objects
    dist.init_from_env()
    rank, world = dist.rank(), dist.world_size()
    assert world == args.gpus, 'WORLD_SIZE %d != --gpus %d' % (world, args.gpus)
    torch.cuda.set_device(dist.local_device_index())
    np.random.seed(1234 + 1000000 * rank)                        # train.py:242
    b_r = args.rollout_batch_size
    experts = args.structure == 'task_experts'
    # ranks of the job and of this process: V per process (--virtual-ranks), or --num-cpu R laid out like experiment.train does
    R_total, layout, base = V * world, [V] * world, None
    if args.num_cpu:
        assert not experts, '--num-cpu: the curious structure (task_experts: --virtual-ranks)'
        V, base, R_total = dist.virtual_layout(args.num_cpu)
        layout = [int(x) for x in dist.allgather_object(V)]
    if experts:
        params, dims, bank, workers = build_experts_job(use_graph=not args.no_graph, env=args.env, b_r=b_r, virtual_ranks=V)
        prefill(bank[0], args.prefill, seed=rank)                 # the buffers are shared by the experts
        counter = [0]

        def step():
            experts_cycle(bank, workers, counter[0])
            counter[0] += 1
        policy, worker = bank[0], workers[0]
    else:
        params, dims, policy, worker = build_job(use_graph=not args.no_graph, env=args.env, b_r=b_r, virtual_ranks=V,
                                                 rank_base=base, total_ranks=R_total if base is not None else None)
        prefill(policy, args.prefill, seed=rank)

        def step():
            cycle(policy, worker)

    for _ in range(args.warmup):
        step()
    freeze_setup_objects()                                        # as experiment/train.py does before its epoch loop
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)

    phases = None
    if args.phases and not experts:
        tr = ts = tu = 0.0
        for _ in range(5):
            torch.cuda.synchronize(); a = time.perf_counter()
            episode, cp, n_ep = worker.generate_rollouts()
            torch.cuda.synchronize(); b = time.perf_counter()
            policy.store_episode(episode, cp, n_ep)
            torch.cuda.synchronize(); c = time.perf_counter()
            policy.train_batches(N_BATCHES)
            policy.update_target_net()
            torch.cuda.synchronize(); d = time.perf_counter()
            tr += b - a; ts += c - b; tu += d - c
        phases = dict(rollout_ms=round(tr / 5 * 1e3, 3), store_ms=round(ts / 5 * 1e3, 3),
                      updates_ms=round(tu / 5 * 1e3, 3))

    # per-kernel HIP-event timing of the same cycle (eager launches; events cannot sit inside a replayed hipGraph)
    prof_cycles = 1
    if experts:
        stats, overhead_ms, eager_update_ms = profile_pass(policy, worker, prof_cycles, bank=bank, step=step)
        roof, table = roofline(policy, worker, stats, prof_cycles, overhead_ms, n_experts=len(bank))
    else:
        stats, overhead_ms, eager_update_ms = profile_pass(policy, worker, prof_cycles)
        roof, table = roofline(policy, worker, stats, prof_cycles, overhead_ms)

    coll = None
    if dist.is_distributed():
        coll = collective_report(policy, list(bank) if experts else [policy])
        if world > 1 and not experts and not args.no_ipc_probe and os.environ.get('CURIOUS_ALLREDUCE', 'rccl') != 'ipc':
            torch.cuda.synchronize()
            dist.barrier()
            probe = ipc_probe(args)                                 # every rank runs its child; rank 0's child reports
            dist.barrier()
            coll['ipc_probe'] = probe

    if rank == 0:
        T = params['T']
        # minibatches of BATCH rows one update of the WHOLE job consumes: one per rank (and expert)
        n_exp = len(bank) if experts else 1
        mb_job = n_exp * R_total
        n_pol = mb_job / world                                    # ... per process on average (an uneven --num-cpu layout)
        Vmax = max(layout)
        headline = (args.env == ENV and b_r == B_R and not experts and R_total == world)
        workload = ('%s, %d parallel rollouts x T=%d per GPU, HER future k=4, batch %d, %d updates per cycle, %d per-task '
                    'buffers' % (args.env, b_r * V, T, BATCH, N_BATCHES, policy.nb_tasks + 1))
        if headline:
            workload += ' (configs[1])'
        elif R_total > world and not experts:
            per = ('%d VIRTUAL RANKS per GPU' % Vmax) if min(layout) == Vmax else \
                ('virtual ranks, %s per GPU (dist.virtual_layout)' % ' '.join(str(x) for x in layout))
            workload = ('%s in the reference\'s published regime (readme.md:16, --num_cpu %d) as %s: '
                        'per rank %d rollouts x T=%d, private per-task buffers and seeds, a minibatch of %d per update; '
                        'every update consumes the minibatches of a GPU\'s ranks in one launch sequence, gradients summed '
                        'over all %d (mpi_adam.py:26-28), normaliser sums averaged (normalizer.py:84-94); %d updates per cycle '
                        '(diagnostic, not the headline configuration)' %
                        (args.env, R_total, per, b_r, T, BATCH, R_total, N_BATCHES))
        elif experts:
            workload += ('; structure=task_experts: %d experts, every update applies to all of them in one batched launch '
                         'sequence (configs[4], diagnostic)' % n_exp)
            if V > 1:
                workload += ('; %d virtual ranks per GPU: every expert\'s update sums the gradients of %d minibatches, one '
                             'from each rank\'s buffer of its task (train.py:65-121, mpi_adam.py:26)' % (V, V))
        else:
            workload += ' (diagnostic, not the headline configuration)'
        # the parameter side of an update (40 B / param) is spread over the minibatches ONE process puts through it
        bpt = 47213.0 if Vmax == 1 else 1034.0 + (47213.0 - 1034.0) / (R_total / world)
        out = {
            'metric': 'HER-sampled gradient transitions/sec (+ env_steps_per_sec), %s cycle' % args.env,
            'value': round(args.steps * N_BATCHES * BATCH * mb_job / elapsed, 1),
            'unit': 'transitions/s',
            'env_steps_per_sec': round(args.steps * b_r * R_total * T / elapsed, 1),
            'updates_per_sec_per_gpu': round(args.steps * N_BATCHES * n_exp / elapsed, 1),
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload,
                       'step': 'one cycle: rollout + store_episode + 100 x train() + update_target_net',
                       'rollout_batch_size': b_r, 'batch_size': BATCH, 'n_batches': N_BATCHES, 'virtual_ranks': Vmax,
                       'ranks': R_total, 'ranks_per_process': layout,
                       'hipgraph': not args.no_graph, 'rng': 'device (Philox)',
                       'parallelism': 'dp%d' % world},
            'roofline': roof,
            # SURVEY 8d whole-update figure: 47 213 algorithmic bytes per gradient transition (1 034 B of HER rows + the
            # 40 B/param of an update spread over its 256 transitions -- over V x 256 with virtual ranks)
            'step_hbm': {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS * world,
                         'bytes_per_transition': round(bpt, 1),
                         'achieved': round(args.steps * N_BATCHES * BATCH * mb_job / elapsed * bpt / 1e9, 2),
                         'frac': round(args.steps * N_BATCHES * BATCH * n_pol / elapsed * bpt / 1e9 / HBM_PEAK_GBS, 5)},
            'kernels': table,
        }
        if coll is not None:
            out['collectives'] = coll
        if phases:
            out['phases'] = phases
        if headline and world == 1 and cpu is not None and not args.no_regime_line:
            out['reference_regime'] = regime_line(args)
        if cpu is not None:
            out['cpu_baseline'] = cpu
        elif world > 1 or args.gpus > 1:
            # (rank 0 at N = 1 only, as the contract says: timing the oracle here would hold N - 1 GPUs idle)
            out['cpu_baseline'] = None
            out['cpu_baseline_note'] = 'measured on rank 0 of the N = 1 run only (python bench.py)'
        print(json.dumps(out), flush=True)
    teardown(list(bank) if experts else [policy], bank if experts else None)


if __name__ == '__main__':
    main()
