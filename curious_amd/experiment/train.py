"""Training entry point.  Mirrors baselines/her/experiment/train.py (train(), logs(), launch(), CLI flags).

Launch one process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N -m curious_amd.experiment.train --env ... --num_cpu N
(the reference re-executes itself under `mpirun -np N`, util.py:148-171; here the ranks are created by
torch.distributed.run and `--num_cpu` only has to match WORLD_SIZE).
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

from curious_amd import dist, logger
from curious_amd.experiment import config
from curious_amd.rollout import RolloutWorker
from curious_amd.util import find_save_path, freeze_setup_objects, mpi_average, thaw_setup_objects

ENV = 'MultiTaskFetchArm4-v5'
NUM_CPU = 1
STRUCTURE = 'curious'                       # 'curious' | 'flat' | 'task_experts'
TASK_SELECTION = 'active_competence_progress'
GOAL_SELECTION = 'random'
GOAL_REPLAY = 'her'
TASK_REPLAY = 'replay_task_cp_buffer'
t0 = time.time()


def train(*args, **kwargs):
    """train.py:49-166 (see _train); the objects that exist when the loop starts are kept out of the cycle collector's
    way for its duration (util.freeze_setup_objects)."""
    from curious_amd.util import BackgroundWriter
    evaluator = kwargs.get('evaluator', args[2] if len(args) > 2 else None)
    writer = BackgroundWriter() if evaluator is not None and not hasattr(evaluator, 'writer') else None
    if writer is not None:
        evaluator.writer = writer                                    # policy files are written behind the training loop
    freeze_setup_objects()
    try:
        return _train(*args, **kwargs)
    finally:
        thaw_setup_objects()
        if writer is not None:
            del evaluator.writer
            writer.close()                                           # every file is on disk when train() returns


class FaultTolerance:
    """What the training loop does with a HandoffFault (curious_amd.ddpg): the optimiser froze the parameters at the
    last good update, so the job can simply go on -- the check that raised has already cleared the word.  An isolated
    fault is logged and survived; a second one within `window` cycles of the first means the hand-off does not work on
    this device / partition at all and is raised.  (fault_check='sync': DDPG.train_batches_guarded replays the faulted
    run of updates instead, bit-identical to a run without the fault.)"""

    def __init__(self, window=64):
        self.window, self.cycle, self.last, self.count = window, 0, None, 0

    def tick(self):
        self.cycle += 1

    def call(self, fn, *args):
        from curious_amd.ddpg import HandoffFault
        try:
            return fn(*args)
        except HandoffFault as err:
            self.count += 1
            if self.last is not None and self.cycle - self.last <= self.window:
                raise
            self.last = self.cycle
            logger.warn('%s -- training goes on from the last good parameters' % err)
            return None


class Checkpointer:
    """Resumable training state on the reference's save cadence (train.py:195-205: every `policy_save_interval` epochs;
    `checkpoint_interval` overrides it, 0 = never) and behind the job's last epoch: curious_amd.checkpoint.save_job_state
    -- a collective, every process writes its own file into `<logdir>/training_state/`."""

    def __init__(self, dirpath, interval, n_epochs, policy, workers, expert_bank=None):
        self.dir, self.interval, self.n_epochs = dirpath, int(interval or 0), int(n_epochs)
        self.policy, self.workers, self.bank = policy, workers, expert_bank

    def due(self, epoch):
        return self.dir is not None and self.interval > 0 and (epoch % self.interval == 0 or epoch == self.n_epochs - 1)

    def save(self, epoch, best_success_rate, ft, extra=None):
        from curious_amd.checkpoint import save_job_state
        loop = dict(best_success_rate=best_success_rate, ft=(ft.cycle, ft.last, ft.count), elapsed=time.time() - t0)
        loop.update(extra or {})
        # (the evaluator carries the job's BackgroundWriter while train() runs: the file is written behind the loop)
        save_job_state(self.dir, epoch, self.policy, self.workers, self.bank, loop,
                       writer=getattr(self.workers[-1], 'writer', None))


def _train(policy, rollout_worker, evaluator, n_epochs, n_test_rollouts, n_cycles, n_batches, policy_save_interval,
           save_policies, structure, task_selection, params, perturbation_study=False, expert_bank=None,
           checkpointer=None, resumed=None, **kwargs):
    """train.py:49-166.  expert_bank (task_experts only): update ALL experts in one batched launch sequence after
    every rollout (BASELINE configs[4]) instead of only the expert that collected it.
    checkpointer: writes the resumable state (Checkpointer); resumed = (epoch, loop dict) of the checkpoint this job was
    restored from (curious_amd.checkpoint.load_job_state): the loop goes on behind that epoch."""
    rank = dist.rank()
    if rank == 0 and logger.get_dir() is not None:
        latest_policy_path = os.path.join(logger.get_dir(), 'policy_latest.pkl')
        best_policy_path = os.path.join(logger.get_dir(), 'policy_best.pkl')
        periodic_policy_path = os.path.join(logger.get_dir(), 'policy_{}.pkl')
        logger.info("Training...")
    else:
        latest_policy_path = best_policy_path = periodic_policy_path = None
    best_success_rate = -1
    nb_tasks = params['nb_tasks']
    ft = FaultTolerance()
    first_epoch = 0
    if resumed is not None:
        first_epoch = resumed[0] + 1
        best_success_rate = resumed[1].get('best_success_rate', -1)
        ft.cycle, ft.last, ft.count = resumed[1].get('ft', (0, None, 0))
    sync_faults = params.get('fault_check', 'async') == 'sync'

    def updates(pol, n):
        if sync_faults and hasattr(pol, 'train_batches_guarded'):
            return pol.train_batches_guarded(n)
        return ft.call(pol.train_batches, n)

    if structure == 'task_experts':
        p = 1 / nb_tasks * np.ones([nb_tasks])
        epoch, i_policy = -1, -1
        if resumed is None:
            evaluator.clear_history()
            evaluator.clear_competence_queue()
            evaluator.generate_eval_rollouts(n_test_rollouts)
            best_success_rate = logs(rollout_worker[i_policy], evaluator, epoch, best_success_rate, best_policy_path,
                                     periodic_policy_path, policy_save_interval, save_policies, latest_policy_path,
                                     policy[i_policy], rank, structure, i_policy=i_policy, task_experts_cp=p)
        for epoch in range(first_epoch, n_epochs):
            if task_selection == 'random':
                i_policy = epoch % nb_tasks                           # train.py:79-81
            else:
                # train.py:82-104: the reference computes `proba` from CP but draws from the stale uniform `p`
                # (SURVEY 7 quirk list); rank 0 draws, everyone follows (C12)
                i_policy = int(np.random.choice(range(nb_tasks), p=p)) if rank == 0 else 0
                i_policy = dist.broadcast_object(i_policy, 0)
            rollout_worker[i_policy].clear_history()
            for _ in range(n_cycles):
                ft.tick()
                episode, cp, n_ep = rollout_worker[i_policy].generate_rollouts()
                ft.call(policy[i_policy].store_episode, episode, cp, n_ep)
                if expert_bank is not None:
                    if expert_bank.trainable():
                        updates(expert_bank, n_batches)               # every expert, one launch sequence per update
                        ft.call(expert_bank.update_target_net)
                    continue
                updates(policy[i_policy], n_batches)                  # = n_batches x train() (train.py:101-102)
                ft.call(policy[i_policy].update_target_net)
            evaluator.clear_history()
            evaluator.generate_eval_rollouts(n_test_rollouts)
            best_success_rate = logs(rollout_worker[i_policy], evaluator, epoch, best_success_rate,
                                     best_policy_path, periodic_policy_path, policy_save_interval, save_policies,
                                     latest_policy_path, policy[i_policy], rank, structure, i_policy=i_policy,
                                     task_experts_cp=p)
            if checkpointer is not None and checkpointer.due(epoch):
                checkpointer.save(epoch, best_success_rate, ft)
    else:
        epoch = -1
        if resumed is None:
            evaluator.clear_history()
            evaluator.generate_eval_rollouts(n_test_rollouts)
            best_success_rate = logs(rollout_worker, evaluator, epoch, best_success_rate, best_policy_path,
                                     periodic_policy_path, policy_save_interval, save_policies, latest_policy_path,
                                     policy, rank, structure)
        for epoch in range(first_epoch, n_epochs):
            logger.info('Starting new epoch ', epoch, 'at time', time.time() - t0)
            t_ep = time.time()
            rollout_worker.clear_history()
            if perturbation_study and (epoch == 250 or (epoch > 250 and epoch == first_epoch)):   # (a job resumed past it)
                perturb_envs(rollout_worker, evaluator)              # train.py:142-146
            for cyc in range(n_cycles):                               # train.py:148-155 -- the hot loop
                ft.tick()
                episode, cp, n_ep = rollout_worker.generate_rollouts()
                ft.call(policy.store_episode, episode, cp, n_ep)
                updates(policy, n_batches)                           # = n_batches x train() (train.py:152-153)
                ft.call(policy.update_target_net)
            evaluator.clear_history()
            evaluator.generate_eval_rollouts(n_test_rollouts)
            torch.cuda.synchronize()
            logger.info('Epoch', epoch, 'over in ', time.time() - t_ep, 's.')
            best_success_rate = logs(rollout_worker, evaluator, epoch, best_success_rate, best_policy_path,
                                     periodic_policy_path, policy_save_interval, save_policies, latest_policy_path,
                                     policy, rank, structure)
            if checkpointer is not None and checkpointer.due(epoch):
                checkpointer.save(epoch, best_success_rate, ft)
    return best_success_rate


def perturb_envs(rollout_worker, evaluator, n=2):
    """The perturbation study's switch (train.py:142-146): from epoch 250 on the first two envs of the training worker and
    of the evaluator return biased observations (`env.unwrapped.bias = True`; what the bias does is the env's business --
    gym_flowers upstream).  An env object without a `bias` attribute cannot honour it: refused, not ignored."""
    for worker in (rollout_worker, evaluator):
        envs = list(worker.envs)
        if len(envs) < n:
            raise NotImplementedError('--perturb needs a Python list of at least %d envs per worker (host envs); the '
                                      'GPU-resident batched env of this build has no observation-bias model' % n)
        for i in range(n):
            env = envs[i].unwrapped
            if not hasattr(env, 'bias'):
                raise NotImplementedError('--perturb: %s has no `bias` switch (train.py:145-146 sets '
                                          'env.unwrapped.bias = True)' % type(env).__name__)
            env.bias = True


def logs(rollout_worker, evaluator, epoch, best_success_rate, best_policy_path, periodic_policy_path,
         policy_save_interval, save_policies, latest_policy_path, policy, rank, structure, i_policy=None,
         task_experts_cp=None):
    """train.py:170-214: same keys in progress.csv.  (Virtual ranks: a process's values are means over ITS ranks and enter
    the cross-process mean with that many votes -- the mean over the job's ranks whatever the layout.)"""
    w = int(getattr(policy, 'V', 1) or 1)
    logger.record_tabular('epoch', epoch)
    for key, val in evaluator.logs('test'):
        logger.record_tabular(key, "%.3g" % mpi_average(val, w))
    for key, val in rollout_worker.logs('train'):
        logger.record_tabular(key, "%.3g" % mpi_average(val, w))
    for key, val in policy.logs():
        logger.record_tabular(key, "%.3g" % mpi_average(val, w))
    if rank == 0:
        if i_policy is not None:
            logger.record_tabular('IND_TASK_rollout', i_policy)
        for key, val in rollout_worker.additional_logs('train'):
            logger.record_tabular(key, val)
        for key, val in evaluator.additional_logs('test'):
            logger.record_tabular(key, val)
        logger.record_tabular('Time', time.time() - t0)
        logger.dump_tabular()
    else:
        logger._state['kv'].clear()
    success_rate = mpi_average(evaluator.current_success_rate(), w)
    snap = []                                                        # one host copy of the policy for all saves below

    def snapshot():
        if not snap:
            snap.append(evaluator.snapshot_policy())
        return snap[0]
    if rank == 0 and success_rate >= best_success_rate and save_policies and best_policy_path:
        best_success_rate = success_rate
        logger.info('New best success rate: {}. Saving policy to {} ...'.format(best_success_rate, best_policy_path))
        evaluator.save_policy(best_policy_path, snapshot())
    if rank == 0 and policy_save_interval > 0 and epoch % policy_save_interval == 0 and save_policies \
            and periodic_policy_path:
        policy_path = periodic_policy_path.format(epoch)
        logger.info('Saving periodic policy to {} ...'.format(policy_path))
        evaluator.save_policy(policy_path, snapshot())
        evaluator.save_policy(latest_policy_path, snapshot())
    # ranks must hold different RNG streams (train.py:207-212, C13)
    local_uniform = np.random.uniform(size=(1,))
    root_uniform = dist.broadcast_object(float(local_uniform[0]), 0)
    if rank != 0 and local_uniform[0] == root_uniform:               # train.py:211-212 (an exception: survives python -O)
        raise dist.RankDivergence('rank %d draws from the same NumPy stream as rank 0' % rank)
    return best_success_rate


def launch(env, trial_id, n_epochs, num_cpu, seed, policy_save_interval, clip_return, normalize_obs, structure,
           task_selection, goal_selection, goal_replay, task_replay, perturb=False, save_policies=True,
           override_params=None, save_root='./save/', resume=None, checkpoint_interval=None):
    """train.py:217-339.
    resume: the log directory of an earlier job of the SAME configuration (its save_root/env/trial directory): the job goes
    on behind that job's last complete checkpoint (curious_amd.checkpoint) -- same directory, progress.csv continued -- up
    to epoch n_epochs - 1.  checkpoint_interval: epochs between checkpoints (default: policy_save_interval, the
    reference's save cadence train.py:195-205; 0: none)."""
    global t0
    dist.init_from_env()
    rank = dist.rank()
    if torch.cuda.is_available():
        torch.cuda.set_device(dist.local_device_index())
    resume_meta = None
    if resume is not None:
        from curious_amd.checkpoint import latest_epoch
        resume = os.path.join(os.path.abspath(resume), '')
        resume_meta = latest_epoch(resume)
        if resume_meta is None:
            raise FileNotFoundError('--resume %s: no complete checkpoint there (training_state/LATEST.json)' % resume)
    if rank == 0:
        save_dir = resume if resume is not None else find_save_path(save_root + env + "/", trial_id)
        # (a job that went on past its last checkpoint before it died logs those epochs again)
        logger.configure(dir=save_dir, resume_epoch=None if resume_meta is None else resume_meta['epoch'])
    else:
        save_dir = None
    # where every process writes its part of a checkpoint: rank 0's log directory
    state_dir = dist.broadcast_object(os.path.abspath(save_dir) if rank == 0 else None, 0)
    if perturb and structure == 'task_experts':
        raise NotImplementedError('--perturb applies to the curious / flat loop only (train.py:142-146)')
    # Virtual ranks: --num_cpu R with R > WORLD_SIZE runs the reference's R ranks on the WORLD_SIZE processes -- R // W per
    # process, one more on the first R % W (readme.md:16: the published runs use 19 ranks, and "fewer cpus for a longer
    # time is NOT equivalent"; train.py:272-281): exactly R ranks whatever the GPU count (dist.virtual_layout).  Process r
    # stands for the global ranks base .. base + V - 1.
    world = dist.world_size()
    V, base, total = 1, rank, world
    if num_cpu > world:
        # what virtual ranks need (DDPG.__init__, RolloutWorker.__init__).  The EFFECTIVE rng mode: DDPG's own default is
        # 'numpy' -- 'device' is the default of the command line only (main() below), a programmatic launch has to name it.
        # Not met: an error.  Fewer ranks than asked for is another job (readme.md:16: "fewer cpus for a longer time is NOT
        # equivalent"), and round 5 ran it behind a warning.
        over = dict(override_params or {})
        missing = []
        if structure not in ('curious', 'task_experts'):
            missing.append("structure 'curious' or 'task_experts' (is %r)" % structure)
        if structure == 'task_experts' and over.get('experts_update', 'sequential') != 'batched':
            missing.append("experts_update 'batched' for task_experts")
        if over.get('rng_mode', 'numpy') != 'device':
            missing.append("override_params['rng_mode'] = 'device' (is %r)" % over.get('rng_mode', 'numpy'))
        if 'buffer' not in task_replay:
            missing.append("per-task buffers (task_replay %r)" % task_replay)
        if missing:
            raise ValueError('--num_cpu %d on %d process(es) means %d ranks per process (virtual ranks), which need: %s.  '
                             'Start %d processes (torch.distributed.run) or pass --num_cpu %d'
                             % (num_cpu, world, -(-num_cpu // world), '; '.join(missing), num_cpu, world))
        V, base, total = dist.virtual_layout(num_cpu)
    # (a process that stands for ONE rank of an uneven layout runs the single-rank agent: nothing of its own needs V)
    rank_seed = seed + 1000000 * base                                 # train.py:242-243 (of this process's first rank)
    np.random.seed(rank_seed)
    import random
    random.seed(rank_seed)
    torch.manual_seed(rank_seed)

    params = dict(config.MULTI_TASK_PARAMS if structure in ('curious', 'task_experts') else config.DEFAULT_PARAMS)
    params.setdefault('eps_task', 0.4)
    params['time'] = str(datetime.datetime.now())
    params.update(env_name=env, task_selection=task_selection, goal_selection=goal_selection, task_replay=task_replay,
                  goal_replay=goal_replay, structure=structure, normalize_obs=normalize_obs, num_cpu=num_cpu,
                  clip_return=clip_return, trial_id=trial_id, seed=seed)
    if override_params:
        params.update(override_params)
    params['virtual_ranks'] = V
    if total != world * V or base != rank * V:
        params['rank_base'], params['total_ranks'] = base, total
    if rank == 0:
        plain = {k: v for k, v in params.items() if isinstance(v, (int, float, str, bool, type(None)))}
        if resume is not None and os.path.exists(os.path.join(resume, 'params.json')):
            with open(os.path.join(resume, 'params.json')) as f:
                was = json.load(f)
            diff = {k: (was[k], plain.get(k)) for k in was if k not in ('time', 'resumed_at_epoch') and was[k] != plain.get(k)}
            if diff:
                raise ValueError('--resume %s: the job there was configured differently (was, now): %s' % (resume, diff))
            plain['resumed_at_epoch'] = resume_meta['epoch']
        with open(os.path.join(logger.get_dir(), 'params.json'), 'w') as f:
            json.dump(plain, f)
    params = config.prepare_params(params)
    params['ddpg_params']['normalize_obs'] = normalize_obs
    params['ddpg_params'].setdefault('seed', seed)                    # identical initial weights on every rank
    if rank == 0:
        config.log_params(params, logger=logger)
    if V > 1:
        logger.info('--num_cpu %d on %d process(es): %d virtual ranks on this GPU, global ranks %d..%d of %d' %
                    (num_cpu, world, V, base, base + V - 1, total))
    elif num_cpu != world:
        logger.warn('--num_cpu %d differs from WORLD_SIZE %d: the ranks are the %d processes torch.distributed.run created'
                    % (num_cpu, world, world))

    dims = config.configure_dims(params)
    buffers = config.configure_buffer(dims=dims, params=params)
    expert_bank = None
    if structure == 'task_experts' and params.get('experts_update', 'sequential') == 'batched':
        from curious_amd.experts import ExpertBank
        expert_bank = ExpertBank(lambda i, **hooks: config.configure_ddpg(dims=dims, params=params, buffers=buffers,
                                                                           clip_return=clip_return, t_id=i, **hooks),
                                 params['nb_tasks'])
        policy = list(expert_bank)
    elif structure == 'task_experts':
        policy = [config.configure_ddpg(dims=dims, params=params, buffers=buffers, clip_return=clip_return, t_id=i)
                  for i in range(params['nb_tasks'])]
    else:
        policy = config.configure_ddpg(dims=dims, params=params, buffers=buffers, clip_return=clip_return)

    rollout_params = {'exploit': False, 'use_target_net': False, 'use_demo_states': True, 'compute_Q': False,
                      'T': params['T'], 'structure': structure, 'task_selection': task_selection,
                      'goal_selection': goal_selection, 'queue_length': params['queue_length'], 'eval': False,
                      'eps_task': params['eps_task']}
    eval_params = {'exploit': True, 'use_target_net': params['test_with_polyak'], 'use_demo_states': False,
                   'compute_Q': True, 'T': params['T'], 'structure': structure, 'task_selection': task_selection,
                   'goal_selection': goal_selection, 'queue_length': params['queue_length'], 'eval': True}
    for name in ['T', 'rollout_batch_size', 'gamma', 'noise_eps', 'random_eps']:
        rollout_params[name] = params[name]
        eval_params[name] = params[name]
    if structure == 'task_experts':
        rollout_worker = [RolloutWorker(params['make_env'], policy[i], dims, logger, unique_task=i, **rollout_params)
                          for i in range(params['nb_tasks'])]
        for i in range(params['nb_tasks']):
            rollout_worker[i].seed(rank_seed + i)
    else:
        rollout_worker = RolloutWorker(params['make_env'], policy, dims, logger, **rollout_params)
        rollout_worker.seed(rank_seed)
    evaluator = RolloutWorker(params['make_env'], policy, dims, logger, **eval_params)
    evaluator.seed(rank_seed + 100)
    if V > 1:                                                         # every virtual rank's own host streams
        if structure == 'task_experts':                               # (+ i: train.py:296-297 seeds worker i with rank_seed + i)
            for i, w in enumerate(rollout_worker):
                w.seed_ranks([seed + 1000000 * (base + v) + i for v in range(V)])
        else:
            rollout_worker.seed_ranks([seed + 1000000 * (base + v) for v in range(V)])
        evaluator.seed_ranks([seed + 1000000 * (base + v) + 100 for v in range(V)])

    workers = (list(rollout_worker) if isinstance(rollout_worker, list) else [rollout_worker]) + [evaluator]
    interval = policy_save_interval if checkpoint_interval is None else checkpoint_interval
    checkpointer = Checkpointer(state_dir, interval if save_policies else 0, n_epochs, policy, workers, expert_bank)
    resumed = None
    if resume is not None:
        from curious_amd.checkpoint import load_job_state
        resumed = load_job_state(state_dir, policy, workers, expert_bank)
        t0 = time.time() - resumed[1].get('elapsed', 0.0)             # the 'Time' column goes on where it stopped
        logger.info('Resumed from %s: epoch %d done, going on to epoch %d' % (state_dir, resumed[0], n_epochs - 1))

    best = train(logdir=save_dir, policy=policy, rollout_worker=rollout_worker, evaluator=evaluator,
                 n_epochs=n_epochs, n_test_rollouts=params['n_test_rollouts'], n_cycles=params['n_cycles'],
                 n_batches=params['n_batches'], perturbation_study=perturb, policy_save_interval=policy_save_interval,
                 save_policies=save_policies, structure=structure, task_selection=task_selection, params=params,
                 expert_bank=expert_bank, checkpointer=checkpointer, resumed=resumed)
    shutdown(policy if isinstance(policy, list) else [policy], expert_bank)
    return best


def shutdown(policies, expert_bank=None):
    """End of a job: last divergence check, then drop every captured graph BEFORE the process group goes away (graphs
    reference the communicator's streams), so that the interpreter exits normally."""
    import gc
    for p in policies:
        p.finish_sync_checks()
        p._graph_a = p._graph_b = p._graph_ba = p._graph_chain = None
        p._chains = None
        p._graphs = [None, None]
        p._roll_graphs = {}
        if getattr(p, '_ipc', None) is not None:
            p._ipc = None
            p._ipc_block.disconnect()                                # (collective: unmap the peers' blocks)
    if expert_bank is not None:
        expert_bank._graphs = {}
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_distributed():
        dist.barrier()
        torch.distributed.destroy_process_group()


def main(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--env', type=str, default=ENV)
    parser.add_argument('--trial_id', type=int, default=0)
    parser.add_argument('--n_epochs', type=int, default=316)
    parser.add_argument('--num_cpu', type=int, default=NUM_CPU)
    parser.add_argument('--seed', type=int, default=int(np.random.randint(1e6)))
    parser.add_argument('--policy_save_interval', type=int, default=50)
    parser.add_argument('--clip_return', type=int, default=1)
    parser.add_argument('--normalize_obs', type=lambda s: s.lower() in ('1', 'true', 'yes'), default=False)
    parser.add_argument('--structure', type=str, default=STRUCTURE)
    parser.add_argument('--task_selection', type=str, default=TASK_SELECTION)
    parser.add_argument('--goal_selection', type=str, default=GOAL_SELECTION)
    parser.add_argument('--goal_replay', type=str, default=GOAL_REPLAY)
    parser.add_argument('--task_replay', type=str, default=TASK_REPLAY)
    parser.add_argument('--perturb', type=lambda s: s.lower() in ('1', 'true', 'yes'), default=False)
    parser.add_argument('--resume', type=str, default=None,
                        help='log directory of an earlier job with the same flags (save/<env>/<trial>/): go on behind its last '
                             'complete checkpoint, bit for bit what the uninterrupted job would have computed')
    parser.add_argument('--checkpoint_interval', type=int, default=None,
                        help='epochs between resumable checkpoints (default: --policy_save_interval, the cadence of '
                             'train.py:195-205; 0: none); one is always written behind the last epoch')
    # MI355X-side knobs
    parser.add_argument('--rollout_batch_size', type=int, default=None)
    parser.add_argument('--rng_mode', type=str, default='device', choices=['numpy', 'device'])
    parser.add_argument('--use_graph', type=int, default=1)
    parser.add_argument('--async_store', type=int, default=1,
                        help='single rank, device RNG: let the device route the episodes of a rollout (no host wait '
                             'between a rollout and its updates); 0 = always wait for the rollout flags first')
    parser.add_argument('--n_cycles', type=int, default=None)
    parser.add_argument('--n_batches', type=int, default=None)
    parser.add_argument('--experts_update', type=str, default='sequential', choices=['sequential', 'batched'],
                        help="task_experts: 'batched' updates all experts in one launch sequence per update")
    parser.add_argument('--fault_check', type=str, default='async', choices=['async', 'sync'],
                        help="guard of the in-kernel Q' hand-off: 'async' reads the verdict cycles later, survives an "
                             "isolated fault on the last good parameters; 'sync' waits for every run of updates and replays "
                             'a faulted one bit-identically (DDPG.train_batches_guarded)')
    args = vars(parser.parse_args(argv))
    over = {'rng_mode': args.pop('rng_mode'), 'use_graph': bool(args.pop('use_graph')),
            'async_store': bool(args.pop('async_store')),
            'experts_update': args.pop('experts_update'), 'fault_check': args.pop('fault_check')}
    for k in ('rollout_batch_size', 'n_cycles', 'n_batches'):
        v = args.pop(k)
        if v is not None:
            over[k] = v
    launch(override_params=over, **args)


if __name__ == '__main__':
    main(sys.argv[1:])
