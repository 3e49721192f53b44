"""Lab: where a step of the weights-resident rollout (policy_resident_kernel) spends its cycles -- per-phase s_memtime
stamps of workgroup 0 (option lab_res_stamps), averaged over the 50 steps of a bench rollout of 256 envs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import ops
    torch.cuda.set_device(0)
    np.random.seed(1)
    params, dims, policy, worker = bench.build_job(use_graph=False)
    bench.prefill(policy, 256, seed=0)
    for _ in range(3):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    n = worker.rollout_batch_size
    ws = policy._act_ws[n]
    off = (n // 4) * 2 * 4 * 256 * 2                              # res_xbuf_floats(n)
    st = ws[off:off + 16].view(torch.int64)
    st.zero_()
    with ops.option('lab_res_stamps', 1):
        for _ in range(4):
            worker.generate_rollouts()
            worker.settle()
    torch.cuda.synchronize()
    v = st.cpu().numpy().astype(np.float64)
    steps = v[7]
    names = ['layer 0', 'hidden 1 slice', 'x2 all-gather', 'hidden 2 slice + h2s', 'output partials', 'x3 + tanh + noise',
             'env step']
    print('policy_resident_kernel, workgroup 0, %d steps: cycles per step' % int(steps))
    for name, c in zip(names, v[:7]):
        print('  %-24s %7.0f' % (name, c / steps))
    print('  %-24s %7.0f' % ('sum', v[:7].sum() / steps))


if __name__ == '__main__':
    main()
