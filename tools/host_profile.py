"""cProfile of the host side of the bench cycle (where does the GPU wait for Python?)."""
import cProfile
import os
import pstats
import sys

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
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(40):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(50)


if __name__ == '__main__':
    main()
