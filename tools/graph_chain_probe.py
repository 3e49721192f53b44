"""Probe: cost of hipGraph launch boundaries.  Times 100 updates replayed as 100 one-update graphs, as 10 ten-update
graphs and as one 100-update graph (same kernels, same order).  Run on a GPU box:  python tools/graph_chain_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import dist
    dist.init_from_env()
    torch.cuda.set_device(0)
    params, dims, policy, worker = bench.build_job(use_graph=True)
    bench.prefill(policy, 2048, seed=0)
    for _ in range(3):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t1 = timed(lambda: [policy.train() for _ in range(100)])
    print('100 x 1-update graph : %.1f us / update' % (t1 * 1e4))
    for K in (2, 5, 10, 25, 100):
        g = policy._capture(lambda: [policy._update_fused(i & 1) for i in range(K)])
        tk = timed(lambda: [g.replay() for _ in range(100 // K)])
        print('%3d x %3d-update graph: %.1f us / update' % (100 // K, K, tk * 1e4))


if __name__ == '__main__':
    main()
