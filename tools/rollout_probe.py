"""Probe: where a batched rollout's time goes (reset / T acting steps eager vs hipGraph / epilogue)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from curious_amd import dist
    dist.init_from_env()
    torch.cuda.set_device(0)
    params, dims, policy, worker = bench.build_job(use_graph=True)
    bench.prefill(policy, 2048, seed=0)
    for _ in range(3):
        bench.cycle(policy, worker)
    env = worker.benv
    B = worker.rollout_batch_size
    tasks = np.random.choice(range(4), size=B)
    goals = np.random.uniform(-1, 1, (B, 3)).astype(np.float32)
    print('generate_rollouts        : %.3f ms' % timed(worker.generate_rollouts))
    print('reset_all                : %.3f ms' % timed(lambda: env.reset_all(tasks, goals)))
    print('act_rollout (graph)      : %.3f ms' % timed(lambda: policy.act_rollout(env, worker.T, 0.2, 0.3)))
    policy.use_graph = False
    print('act_rollout (eager)      : %.3f ms' % timed(lambda: policy.act_rollout(env, worker.T, 0.2, 0.3)))
    policy.use_graph = True
    print('last_success().cpu()     : %.3f ms' % timed(lambda: env.last_success().cpu().numpy()))
    print('isnan(o).any()           : %.3f ms' % timed(lambda: bool(torch.isnan(env.o).any())))
    succ = env.last_success().cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    for _ in range(20):
        worker._finish_rollout(succ, succ - 1.0, None, [int(x) for x in tasks])
    print('_finish_rollout          : %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
    ep, cp, n_ep = worker.generate_rollouts()
    print('store_episode            : %.3f ms' % timed(lambda: policy.store_episode(ep, cp, n_ep)))
    print('update_target_net        : %.3f ms' % timed(policy.update_target_net))
    print('train_batches(100)       : %.3f ms' % timed(lambda: policy.train_batches(100)))


if __name__ == '__main__':
    main()
