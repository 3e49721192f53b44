"""Lab: phase stamps of EVERY row group of ddpg_rows_kernel (option lab_rows_stamps) on the bench job with V virtual ranks --
how long each kind of workgroup lives, where inside its chain the time goes, when (relative to the launch's first
workgroup) it starts and ends, and the shader clock the launch holds (cycle counter against the 100 MHz real-time counter).

    python tools/rows_stamps.py [V] [rows16 threshold]        # e.g. 19, 19 0 (the 8-row form), 8
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

KINDS = ['actor side', 'target', 'main critic']
PHASES = {
    0: ['inputs', 'l0 pi', 'hidden pi', 'head pi', 'l0 + hidden Q', 'head + seed', 'bwd Q', 'dz + seed', 'bwd pi'],
    1: ['inputs', 'l0 pi', 'hidden pi', 'head pi', 'l0 + hidden Q', 'head + publish'],
    2: ['inputs', 'l0 Q', 'hidden Q', "head + wait Q'", 'loss + seed', 'bwd Q'],
}


def main():
    from curious_amd import dist, ops
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    if len(sys.argv) > 2:
        ops.set_option('rows16', int(sys.argv[2]))
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
    R = 16 if (ops.get_option('rows16') > 0 and B >= ops.get_option('rows16')) else (8 if B >= 768 else 4)
    nrg = B // R
    st = ops.dw_stamps(p.net_cfg, B, p._workspace, nrg * 3 * 2).view(-1)[:nrg * 3 * 16].view(nrg, 3, 16)
    print('V = %d, %d rows, %d rows per workgroup, %d row groups per kind' % (V, B, R, nrg))
    with ops.option('lab_rows_stamps', 1):
        for rep in range(3):
            st.zero_()
            upd()
            torch.cuda.synchronize()
        a = st.cpu().numpy().astype(np.int64)
    rt0 = a[:, :, 10][a[:, :, 10] > 0].min()
    for kind in range(3):
        last = len(PHASES[kind])
        t = a[:, kind, :last + 1]
        life = t[:, last] - t[:, 0]
        rt = (a[:, kind, 11] - a[:, kind, 10]) * 10.0                # ns
        clk = life / np.maximum(rt, 1)                                # cycles per ns = GHz
        print('%-12s life %7.1f k cycles (min %.1f, max %.1f), %.2f GHz; starts %5.1f .. %5.1f us, ends %5.1f .. %5.1f us after '
              'the first stamp' % (KINDS[kind], life.mean() / 1e3, life.min() / 1e3, life.max() / 1e3, np.median(clk),
                                   (a[:, kind, 10].min() - rt0) / 100.0, (a[:, kind, 10].max() - rt0) / 100.0,
                                   (a[:, kind, 11].min() - rt0) / 100.0, (a[:, kind, 11].max() - rt0) / 100.0))
        d = np.diff(t, axis=1)
        print('   phases (k cycles, mean over the row groups): ' +
              '  '.join('%s %.1f' % (PHASES[kind][i] if i < len(PHASES[kind]) else str(i), d[:, i].mean() / 1e3)
                        for i in range(d.shape[1])))


if __name__ == '__main__':
    main()
