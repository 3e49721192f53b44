"""Lab (verdict r2 item 5): what would an update cost if the target networks' work left the per-update chain?

Within a cycle theta' and the replay storage are constant (train.py:148-155: update_target_net after the n_batches loop),
so target actor -> target critic -> y could be evaluated once per cycle for all n_batches x B sampled rows.  Measured here,
as chained hipGraph replays of 100 updates (the form train_batches replays):
  a. the update as shipped (target groups inside ddpg_rows_kernel, gather of the next batch in the tail launch)
  b. the same without the gather blocks in the tail launch
  c. b + option lab_no_target: the target groups exit at once, the main-critic groups do not wait (wrong numbers, right
     timing): the upper bound of what hoisting the targets can save per update
and the price: d. one forward of actor + critic (curious_policy_forward with Q) over n_batches x B = 25 600 rows, on the
row-local and on the tiled route -- the per-cycle precompute, existing kernels, no HER gather included.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    from curious_amd import ops
    from curious_amd.ddpg import CAPTURE_MODE
    torch.cuda.set_device(0)
    params, dims, p, worker = bench.build_job(use_graph=False)
    bench.prefill(p, 2048, seed=0)
    for _ in range(2):
        bench.cycle(p, worker)
    S = p.sample_transitions
    p._train_device_prologue(100)
    p._sample_packed()

    def upd(i, her):
        kw = dict(next_batch=p._pp[(i & 1) ^ 1], storage=p._pool.storage, buf_stride=p._pool.buf_stride, tasks=S.tasks,
                  params=S.params(p.clip_obs, p.relative_goals), rng=p._rng_desc) if her else {}
        ops.ddpg_update(p.net_cfg, p.theta, p.theta_target, p._pp[i & 1], p._layout, p.batch_size, p._workspace, p.grad,
                        p._losses, p._Q_pi, p._m, p._v, step_ctr=p._step_ctr, alpha_tab=p._alpha_tab,
                        tab_base=p._alpha_base, params_unchanged=i > 0, **kw)

    def chain(her):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            for i in range(100):
                upd(i, her)
        return g

    state = (p.theta.clone(), p._m.clone(), p._v.clone())
    res = {}
    for name, her, no_target in (('a. as shipped', True, 0), ('b. no gather in the tail', False, 0),
                                 ('c. no gather, no target groups', False, 1)):
        with ops.option('lab_no_target', no_target):
            g = chain(her)
        p._step_ctr.fill_(0)
        res[name] = timed(g.replay) * 10.0                       # ms per 100 updates -> us per update
        p.theta.copy_(state[0]); p._m.copy_(state[1]); p._v.copy_(state[2])
        print('%-34s %.2f us per update (chained replay of 100)' % (name, res[name]), flush=True)
    n = 100 * p.batch_size
    o = torch.randn([n, p.dimo], device='cuda')
    g_ = torch.randn([n, p.dimg], device='cuda')
    td = torch.zeros([n, p.dimtd], device='cuda')
    td[:, 0] = 1
    ws = torch.empty(ops.workspace_floats(p.net_cfg, n), device='cuda')
    pi = torch.empty([n, 4], device='cuda')
    Q = torch.empty([n, 1], device='cuda')
    for route in (1, 0):
        with ops.option('rows', route):
            ms = timed(lambda: ops.policy_forward(p.net_cfg, p.theta_target, o, g_, td, n, 200.0, ws, pi, Q), reps=10)
        flop = 2.0 * n * (p.P_Q + p.P_pi)
        print('d. actor + critic forward over %d rows, %s route: %.3f ms = %.2f us per update, %.1f TFLOP/s' %
              (n, 'row-local' if route else 'tiled', ms, ms * 10.0, flop / ms / 1e9), flush=True)
    a, c = res['a. as shipped'], res['c. no gather, no target groups']
    print('upper bound of the saving: %.2f us per update (a - c)' % (a - c))


if __name__ == '__main__':
    main()
