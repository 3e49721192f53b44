"""Lab: when the blocks of dw_adam_her_kernel start and end, by kind (a build with -DDW_STAMPS + option lab_dw_stamps), on the
bench job with V virtual ranks -- which kind of block the launch ends with.

    python tools/build_variant.py dwst -DDW_STAMPS
    CURIOUS_LIB=abtest/dwst.so python tools/dw_timeline.py [V]          # CURIOUS_DW64=0: the 16 x 64 hidden tiles
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import dist, ops
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    dist.init_from_env()
    torch.cuda.set_device(0)
    params, dims, p, worker = bench.build_job(use_graph=False, b_r=2 if V > 1 else 256, virtual_ranks=V)
    bench.prefill(p, 2048, seed=0)
    for _ in range(2):
        bench.cycle(p, worker)
    B = p._Bt
    S = p.sample_transitions

    def upd():
        ops.ddpg_update(p.net_cfg, p.theta, p.theta_target, p._pp[0], p._layout, B, p._workspace, p.grad,
                        p._losses, p._Q_pi, p._m, p._v, step_ctr=p._step_ctr, alpha_tab=p._alpha_tab,
                        next_batch=p._pp[1], storage=p._pool.storage, buf_stride=p._pool.buf_stride, tasks=S.tasks,
                        params=S.params(p.clip_obs, p.relative_goals), rng=p._rng_desc)
    for _ in range(20):
        upd()
    nb = 6 * 16 * B // 16
    st = ops.dw_stamps(p.net_cfg, B, p._workspace, nb)
    names = {1: 'gather', 2: 'hidden tile', 3: 'small tile / fin'}
    print('V = %d, %d rows; dw64 = %d, dw_xcd = %d' % (V, B, ops.get_option('dw64'), ops.get_option('dw_xcd')))
    with ops.option('lab_dw_stamps', 1):
        for rep in range(3):
            st.zero_()
            upd()
            torch.cuda.synchronize()
        a = st.cpu().numpy().astype(np.int64)
    live = a[:, 4] > 0
    t0 = a[live, 5].min()
    print('%-18s %6s | life k cycles: mean  max | starts (us after the first block): min  mean  max | ends ~ (us): mean  max'
          % ('kind', 'blocks'))
    for kind in (1, 2, 3):
        sel = live & (a[:, 4] == kind) & (a[:, 3] > a[:, 0])
        if not sel.any():
            continue
        r = a[sel]
        life = (r[:, 3] - r[:, 0])
        start = (r[:, 5] - t0) / 100.0
        end = start + life / 2400.0                                  # (~2.4 GHz)
        print('%-18s %6d | %19.1f %6.1f | %38.1f %6.1f %6.1f | %18.1f %6.1f' %
              (names[kind], sel.sum(), life.mean() / 1e3, life.max() / 1e3, start.min(), start.mean(), start.max(),
               end.mean(), end.max()))
    ops.prof_collect()
    ops.prof_enable(True)
    for _ in range(100):
        upd()
    ops.prof_enable(False)
    stt = ops.prof_collect()
    print({k: round(1e3 * v[1] / v[0], 2) for k, v in stt.items() if v[0]})


if __name__ == '__main__':
    main()
