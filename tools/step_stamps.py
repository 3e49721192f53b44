"""Lab: timeline of ONE one-launch update (ddpg_step_kernel) from cycle stamps (option lab_step bit 3): row group 0 of
each kind (actor side, target, main critic) and the tiles those workgroups (and spare block 0) work on afterwards.
The cycle counters of different XCDs are not aligned: only differences inside one workgroup mean something.

    python tools/step_stamps.py [lab_step bits to add, e.g. 1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import ops
    extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    torch.cuda.set_device(0)
    np.random.seed(1)
    params, dims, policy, worker = bench.build_job(use_graph=False)
    bench.prefill(policy, 256, seed=0)
    for _ in range(2):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    B, H, nl = policy.batch_size, policy.hidden, policy.layers
    fault = ops.fault_word(policy.net_cfg, B, policy._workspace)
    fault_off = (fault.data_ptr() - policy._workspace.data_ptr()) // 4
    part0 = fault_off - 2 * (nl - 1) * H * H - 2 * B - 6 * 16 * B  # mlp.hip carve(): part[6], qt, wT, fault
    st = policy._workspace[part0:part0 + 2 * 256].view(torch.int64)
    with ops.option('lab_step', 8 | extra):
        policy.train_batches(6)
        torch.cuda.synchronize()
        st.zero_()
        policy.train_batches(2)                                   # the stamps of the second one stay
        torch.cuda.synchronize()
    v = st.cpu().numpy().astype(np.int64)
    t0 = v[v > 0].min()
    rel = lambda x: (x - t0) if x > 0 else -1
    for kind, name, n in ((0, 'actor side ', 10), (1, 'target     ', 7), (2, 'main critic', 7)):
        print('row group 0, %s stamps: %s' % (name, [int(v[kind * 32 + k] - v[kind * 32]) for k in range(n)]))
    print('tiles of four workers (cycles; a worker\'s own counter): start -> wait begins, wait over | wait over -> operands in + MFMA, end')
    for wi, wname in enumerate(('spare block 0', 'target group 0', 'main-critic group 0', 'actor-side group 0')):
        for k in range(4):
            s = v[128 + 8 * (4 * wi + k): 128 + 8 * (4 * wi + k) + 8]
            if s[0] == 0:
                continue
            print('  %-20s tile %d: %7d %7d | %7d %7d      (start at %d after the group\'s first stamp)' %
                  (wname, k, s[1] - s[0], s[2] - s[0], s[3] - s[2] if s[3] else -1, s[4] - s[2],
                   s[0] - v[[0, 32, 64, 0][wi]] if wi in (1, 2, 3) else 0))
    print('span of all stamps: %d cycles' % (v.max() - t0))


if __name__ == '__main__':
    main()
