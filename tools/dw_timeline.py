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
    a_kind = None
    print('V = %d, %d rows; dw64 = %d, dw_xcd = %d' % (V, B, ops.get_option('dw64'), ops.get_option('dw_xcd')))
    with ops.option('lab_dw_stamps', 1):
        for rep in range(3):
            st.zero_()
            upd()
            torch.cuda.synchronize()
        a = st.cpu().numpy().astype(np.int64)
    a[a[:, 4] == 4, 4] = 3                                       # (small tiles dealt in halves: dw_role kind 3)
    live = a[:, 4] > 0
    # where every block ran (a[:, 7] bits 32..: HW_ID, XCC_ID): CU key = (xcc, se, sh, cu)
    hw = (a[:, 7] >> 32) & 0xffff
    cu_key = (((a[:, 7] >> 48) & 0xf) << 12) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    a[:, 7] &= 0xffffffff
    if live.any():
        kinds = {}
        busy = live & (a[:, 3] - a[:, 0] > 4000)                    # (blocks that found a tile / transitions to do)
        for k, c in zip(a[busy, 4], cu_key[busy]):
            kinds.setdefault(int(c), []).append(int(k))
        from collections import Counter
        mix = Counter(''.join(sorted('gHs'[k - 1] if 1 <= k <= 3 else '?' for k in v)) for v in kinds.values())
        print('%d CUs seen; blocks per CU by kind (g gather, H hidden tile, s small tile): %s' % (len(kinds), dict(mix)))
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
    # when the arguments of a tile had arrived (stamp 3, dw_tile_role) and when its operand loads were out (stamp 1)
    for kind in (2, 3):
        sel = live & (a[:, 4] == kind) & (a[:, 3] > a[:, 0]) & (a[:, 1] > a[:, 0]) & (a[:, 6] > a[:, 0])
        if sel.any():
            r = a[sel]
            print('%-18s entry -> arguments in %5.1f k cycles -> operand loads out %5.1f k -> products done %5.1f k -> exit %5.1f k  (means over %d blocks that did a tile)'
                  % (names[kind], (r[:, 6] - r[:, 0]).mean() / 1e3, (r[:, 1] - r[:, 6]).mean() / 1e3,
                     (r[:, 2] - r[:, 1]).mean() / 1e3, (r[:, 3] - r[:, 2]).mean() / 1e3, sel.sum()))
    # hidden tiles in detail: by XCD (block id & 7), and the phases of the slowest / fastest tenth
    sel = live & (a[:, 4] == 2) & (a[:, 3] > a[:, 0])
    ids = np.nonzero(sel)[0]
    r = a[sel]
    life = (r[:, 3] - r[:, 0]) / 1e3
    head, loop, tail = (r[:, 1] - r[:, 0]) / 1e3, (r[:, 2] - r[:, 1]) / 1e3, (r[:, 3] - r[:, 2]) / 1e3
    print('hidden tiles by XCD: ' + '  '.join('%d: %.0f/%.0f' % (x, life[(ids & 7) == x].mean(), life[(ids & 7) == x].max())
                                              for x in range(8) if ((ids & 7) == x).any()) + '   (k cycles mean/max)')
    order = np.argsort(life)
    n10 = max(1, len(order) // 10)
    for name, pick in (('fastest tenth', order[:n10]), ('median tenth', order[len(order) // 2 - n10 // 2:][:n10]), ('slowest tenth', order[-n10:])):
        print('%-14s life %6.1f = head %5.1f + loop %6.1f + tail %5.1f k cycles; start %5.2f us; ids e.g. %s' %
              (name, life[pick].mean(), head[pick].mean(), loop[pick].mean(), tail[pick].mean(),
               ((r[pick, 5] - t0) / 100.0).mean(), ids[pick][:8].tolist()))
    # the same split by the row (block id >> 3) within the hidden rows: which (tile, segment) rows are slow
    rows = ids >> 3
    ur = np.unique(rows)
    print('by block row (mean life k cycles): ' + ' '.join('%d:%.0f' % (q, life[rows == q].mean()) for q in ur[:80]))
    # the slowest small tiles: block id, tile index (problem * slots + tile), life, start
    sel = live & (a[:, 4] == 3) & (a[:, 3] > a[:, 0])
    ids = np.nonzero(sel)[0]
    r = a[sel]
    life = (r[:, 3] - r[:, 0]) / 1e3
    order = np.argsort(-life)[:24]
    print('slowest small tiles (block id, index, life k cycles = head + loop + tail, start us):')
    for i in order:
        print('   %5d %4d  %6.1f = %5.1f + %5.1f + %5.1f   %5.1f' % (ids[i], r[i, 7], life[i], (r[i, 1] - r[i, 0]) / 1e3,
              (r[i, 2] - r[i, 1]) / 1e3, (r[i, 3] - r[i, 2]) / 1e3, (r[i, 5] - t0) / 100.0))
    idx = r[:, 7]
    print('small tiles by index (mean life k cycles over the segments): ' +
          ' '.join('%d:%.0f' % (q, life[idx == q].mean()) for q in np.unique(idx)))
    ops.prof_collect()
    ops.prof_enable(True)
    for _ in range(100):
        upd()
    ops.prof_enable(False)
    stt = ops.prof_collect()
    print({k: round(1e3 * v[1] / v[0], 2) for k, v in stt.items() if v[0]})


if __name__ == '__main__':
    main()
