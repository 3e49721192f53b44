"""Lab: per-block cycle stamps of dw_adam_her_kernel (option lab_dw_stamps) on the bench job -- where a gather block, a
hidden-layer tile and a small-problem tile spend their time, and when (relative to the launch's first block) they start.

    python tools/dw_stamps.py            # plain block order
    CURIOUS_DW_XCD=1 python tools/dw_stamps.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import dist, ops
    dist.init_from_env()
    torch.cuda.set_device(0)
    params, dims, p, worker = bench.build_job(use_graph=False)
    bench.prefill(p, 2048, seed=0)
    for _ in range(2):
        bench.cycle(p, worker)
    S = p.sample_transitions

    def upd():
        ops.ddpg_update(p.net_cfg, p.theta, p.theta_target, p._pp[0], p._layout, p.batch_size, p._workspace, p.grad,
                        p._losses, p._Q_pi, p._m, p._v, step_ctr=p._step_ctr, alpha_tab=p._alpha_tab,
                        next_batch=p._pp[1], storage=p._pool.storage, buf_stride=p._pool.buf_stride, tasks=S.tasks,
                        params=S.params(p.clip_obs, p.relative_goals), rng=p._rng_desc)
    for _ in range(20):
        upd()
    nb = 512
    st = ops.dw_stamps(p.net_cfg, p.batch_size, p._workspace, nb)
    names = {1: 'gather', 2: 'hidden tile', 3: 'small tile / fin'}
    print('dw_xcd =', ops.get_option('dw_xcd'))
    with ops.option('lab_dw_stamps', 1):
        acc = {}
        for rep in range(10):
            st.zero_()
            upd()
            torch.cuda.synchronize()
            a = st.cpu().numpy().astype(np.int64)
            live = a[:, 4] > 0
            t0 = a[live, 5].min()
            for kind in (1, 2, 3):
                sel = live & (a[:, 4] == kind) & (a[:, 3] > a[:, 0])
                if not sel.any():
                    continue
                rows = a[sel]
                acc.setdefault(kind, []).append([
                    (rows[:, 3] - rows[:, 0]).mean(), (rows[:, 3] - rows[:, 0]).max(),
                    (rows[:, 1] - rows[:, 0])[rows[:, 1] > 0].mean() if (rows[:, 1] > 0).any() else 0,
                    (rows[:, 2] - rows[:, 1])[rows[:, 2] > 0].mean() if (rows[:, 2] > 0).any() else 0,
                    (rows[:, 3] - rows[:, 2])[rows[:, 2] > 0].mean() if (rows[:, 2] > 0).any() else 0,
                    (rows[:, 5] - t0).mean() * 10.0, (rows[:, 5] - t0).max() * 10.0, sel.sum(),
                    (rows[:, 6] - rows[:, 0])[rows[:, 6] > 0].mean() if (rows[:, 6] > 0).any() else 0])
        # the slowest blocks of the last launch
        order = np.argsort(-(a[:, 3] - a[:, 0]) * live)[:12]
        print('slowest blocks of one launch: block id, kind, pi, idx | total = prologue + loads/mfma + epilogue | start ns')
        for b in order:
            r = a[b]
            print('  %4d %-16s %2d %4d | %6d = %6d + %6d + %6d | %5d' %
                  (b, names[int(r[4])], 0, r[7], r[3] - r[0], max(r[1] - r[0], 0), max(r[2] - r[1], 0),
                   r[3] - max(r[2], r[0]), (r[5] - t0) * 10))
        if os.environ.get('DW_STAMPS_ALL'):
            print('all small-tile blocks of one launch by (problem, tile): total = prologue + loads/mfma + epilogue')
            sm = [(int(r[7]), int(b)) for b, r in enumerate(a) if live[b] and r[4] == 3 and r[3] > r[0]]
            for idx, b in sorted(sm):
                r = a[b]
                print('  p %2d t %2d xcd %d | %6d = %6d + %6d + %6d' % (idx // 16, idx % 16, b & 7, r[3] - r[0],
                      max(r[1] - r[0], 0), max(r[2] - r[1], 0), r[3] - max(r[2], r[0])))
        print('%-18s %6s | %9s %9s | %9s %9s %9s | start after the first block (ns): mean, max' %
              ('kind', 'blocks', 'mean cyc', 'max cyc', 'prologue', 'loads+mfma', 'epilogue'))
        for kind, rows in acc.items():
            m = np.median(np.array(rows), axis=0)
            print('%-18s %6d | %9.0f %9.0f | %9.0f %9.0f %9.0f | %7.0f %7.0f | arguments in after %5.0f' %
                  (names[kind], m[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[8]))
    # duration of the launch by events, eager
    ops.prof_collect()
    ops.prof_enable(True)
    for _ in range(200):
        upd()
    ops.prof_enable(False)
    stt = ops.prof_collect()
    print({k: round(1e3 * v[1] / v[0], 2) for k, v in stt.items() if v[0]})


if __name__ == '__main__':
    main()
