"""Probe: duration of the update's last launch in its three forms (HIP-event brackets, eager):
dw_all_kernel (gradients only), dw_adam_her_kernel without / with the next HER gather."""
import os
import sys

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

    def grads():
        ops.ddpg_grads(p.net_cfg, p.theta, p.theta_target, p._pp[0], p._layout, p.batch_size, p._workspace, p.grad,
                       p._losses, p._Q_pi, step_ctr=p._step_ctr)

    def upd(her):
        kw = dict(next_batch=p._pp[1], storage=p._pool.storage, buf_stride=p._pool.buf_stride, tasks=S.tasks,
                  params=S.params(p.clip_obs, p.relative_goals), rng=p._rng_desc) if her else {}
        ops.ddpg_update(p.net_cfg, p.theta, p.theta_target, p._pp[0], p._layout, p.batch_size, p._workspace, p.grad,
                        p._losses, p._Q_pi, p._m, p._v, step_ctr=p._step_ctr, alpha_tab=p._alpha_tab, **kw)

    for name, fn in (('grads only', grads), ('update, no gather', lambda: upd(False)),
                     ('update + gather', lambda: upd(True))):
        for _ in range(20):
            fn()
        ops.prof_collect()
        ops.prof_enable(True)
        for _ in range(200):
            fn()
        ops.prof_enable(False)
        st = ops.prof_collect()
        print(name, {k: round(1e3 * v[1] / v[0], 2) for k, v in st.items() if v[0] and k.startswith('dw')},
              'sum of all launches: %.1f us' % (1e3 * sum(v[1] for v in st.values()) / 200))


if __name__ == '__main__':
    main()
