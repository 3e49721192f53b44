"""Probe: the stand-alone optimiser launch of the multi-rank path with / without the HER gather and the transposed-copy
tiles (HIP-event brackets, eager)."""
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
    worker.settle(); p.settle()
    S = p.sample_transitions
    keep = p._kept_copies()
    n_Q, n_pi = p.off_pi, p.P_total - p.off_pi

    def grads():
        ops.ddpg_grads(p.net_cfg, p.theta, p.theta_target, p._pp[0], p._layout, p.batch_size, p._workspace, p.grad,
                       p._losses, p._Q_pi, step_ctr=p._step_ctr)

    def adam(k):
        ops.adam_update(p.theta, p._m, p._v, p.grad, n_Q, n_pi, alpha_tab=p._alpha_tab, step_ctr=p._step_ctr,
                        tab_base=p._alpha_base, keep=k)

    def adam_her(k):
        ops.adam_update_and_sample(p.theta, p._m, p._v, p.grad, n_Q, n_pi, p._alpha_tab, p._step_ctr, p._alpha_base,
                                   p._pool.storage, p._pool.buf_stride, p._layout, S.tasks,
                                   S.params(p.clip_obs, p.relative_goals), p._rng_desc, p.batch_size, p._pp[1], keep=k)

    for name, fn in (('adam', lambda: adam(None)), ('adam + copies', lambda: adam(keep)),
                     ('adam + gather', lambda: adam_her(None)), ('adam + gather + copies', lambda: adam_her(keep))):
        def seq():
            grads()
            fn()
        for _ in range(20):
            seq()
        ops.prof_collect()
        ops.prof_enable(True)
        for _ in range(200):
            seq()
        ops.prof_enable(False)
        st = ops.prof_collect()
        print(name, {k: round(1e3 * v[1] / v[0], 2) for k, v in st.items() if v[0] and ('adam' in k or 'transpose' in k)})


if __name__ == '__main__':
    main()
