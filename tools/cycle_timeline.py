"""Probe: where one bench cycle spends its wall-clock time -- host stamps at the phase boundaries of bench.cycle and GPU
event stamps around the rollout, the store and the updates (is the GPU waiting for Python between the phases?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    np.random.seed(1)
    params, dims, policy, worker = bench.build_job(use_graph=True)
    bench.prefill(policy, 2048, seed=0)
    for _ in range(5):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    n = 30
    # finer host stamps on the path on which the GPU waits for Python: flag sync returns -> first update graph replays
    marks = {'sync': [], 'replay': []}
    env = worker.benv
    orig_wait = env.wait_flags

    def wait():
        out = orig_wait()
        marks['sync'].append(time.perf_counter())
        return out
    env.wait_flags = wait
    orig_replay = torch.cuda.CUDAGraph.replay

    def replay(self):
        marks['replay'].append(time.perf_counter())
        return orig_replay(self)
    torch.cuda.CUDAGraph.replay = replay
    host = np.zeros([n, 5])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    t00 = time.perf_counter()
    for c in range(n):
        ev[c][0].record()
        host[c, 0] = time.perf_counter()
        episode, cp, n_ep = worker.generate_rollouts()
        ev[c][1].record()
        host[c, 1] = time.perf_counter()
        policy.store_episode(episode, cp, n_ep)
        ev[c][2].record()
        host[c, 2] = time.perf_counter()
        policy.train_batches(bench.N_BATCHES)
        ev[c][3].record()
        host[c, 3] = time.perf_counter()
        policy.update_target_net()
        ev[c][4].record()
        host[c, 4] = time.perf_counter()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t00) / n * 1e3
    h = np.diff(host, axis=1).mean(axis=0) * 1e3
    nxt = (host[1:, 0] - host[:-1, 4]).mean() * 1e3
    g = np.array([[ev[c][i].elapsed_time(ev[c][i + 1]) for i in range(4)] for c in range(n)]).mean(axis=0)
    gap = np.mean([ev[c][4].elapsed_time(ev[c + 1][0]) for c in range(n - 1)])
    print('cycle %.3f ms (async_store %s)' % (total, 'on' if policy.async_store else 'off'))
    sync = np.array(marks['sync'][-n:])
    rep = np.array(marks['replay']).reshape(-1, 3)[-n:]           # per cycle: rollout graph, two update chains
    if not policy.async_store:
        print('host  ms after the flag sync returned: -> generate_rollouts returns %.3f | -> store_episode returns %.3f | '
              '-> first update graph launched %.3f | -> second %.3f'
              % (((host[:, 1] - sync).mean()) * 1e3, ((host[:, 2] - sync).mean()) * 1e3,
                 ((rep[:, 1] - sync).mean()) * 1e3, ((rep[:, 2] - sync).mean()) * 1e3))
    else:
        print('host  ms: the flags of a rollout are read at the start of the next cycle (the wait returns %.3f ms after the '
              'cycle started on the host); rollout graph -> first update graph launched %.3f'
              % (((sync - host[:, 0]).mean()) * 1e3, ((rep[:, 1] - rep[:, 0]).mean()) * 1e3))
    print('host  ms: rollout (incl. the flag sync) %.3f | store %.3f | train_batches (enqueue) %.3f | target %.3f | loop %.3f'
          % (h[0], h[1], h[2], h[3], nxt))
    print('GPU   ms between event stamps: rollout %.3f | store %.3f | updates %.3f | target %.3f | to next cycle %.3f'
          % (g[0], g[1], g[2], g[3], gap))


if __name__ == '__main__':
    main()
