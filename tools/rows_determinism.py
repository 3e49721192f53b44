"""Stress test of the row-local launch's in-kernel synchronisation (one workgroup barrier per layer, partial tiles
alternating between two LDS buffers, the Q' hand-off): N times the SAME gradient call -- same parameters, same batch --,
every output compared bit for bit with the first call's.  A race shows up as a launch whose gradient differs.
    python tools/rows_determinism.py [N]            (default 20000; also with the whole update in a loop of graphs)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def experts(n):
    """The same for 4 batched experts: 768 workgroups of the row-local launch on 256 CUs, two per CU."""
    params, dims, bank, workers = bench.build_experts_job(use_graph=False)
    bench.prefill(bank[0], 256, seed=0)
    for k in range(2):
        bench.experts_cycle(bank, workers, k)
    bank._prologue(1)
    torch.cuda.synchronize()
    x0 = bank[0]
    # (not the whole slab: the call also gathers every expert's NEXT batch, keyed by a step counter it advances)
    outs = lambda: [bank.grads] + [x._losses for x in bank] + [x._Q_pi for x in bank]
    bank._grads_all(bank._cur)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs()]
    acc = torch.zeros([], dtype=torch.int64, device=ref[0].device)
    bad = 0
    for i in range(n):
        bank._grads_all(bank._cur)
        for t, r in zip(outs(), ref):
            acc += (~torch.eq(t.view(torch.int32), r.view(torch.int32))).sum()
        if (i + 1) % 500 == 0:
            bad = int(acc)
            if bad:
                print('launch <= %d: %d differing words' % (i + 1, bad))
                break
    bank.check_faults()
    print('%d launches of the batched experts\' gradient call: %s' % (n, 'all identical to the first' if not bad else 'MISMATCH'))
    return 1 if bad else 0


def main():
    if len(sys.argv) > 2 and sys.argv[2] == 'experts':
        return experts(int(sys.argv[1]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    params, dims, policy, worker = bench.build_job(use_graph=False)
    bench.prefill(policy, 256, seed=0)
    for _ in range(3):
        bench.cycle(policy, worker)                     # parameters away from their initial values
    policy.stage_batch()
    torch.cuda.synchronize()
    policy._grads()
    torch.cuda.synchronize()
    g0 = policy.grad.clone()
    l0 = policy._losses.clone()
    q0 = policy._Q_pi.clone()
    bad = 0
    acc = torch.zeros([], dtype=torch.int64, device=g0.device)
    for i in range(n):
        policy._grads()
        # compared on the device; one host sync per 1000 launches
        acc += (~torch.eq(policy.grad.view(torch.int32), g0.view(torch.int32))).sum()
        acc += (~torch.eq(policy._losses.view(torch.int32), l0.view(torch.int32))).sum()
        acc += (~torch.eq(policy._Q_pi.view(torch.int32), q0.view(torch.int32))).sum()
        if (i + 1) % 1000 == 0:
            bad = int(acc)
            if bad:
                print('launch <= %d: %d differing words' % (i + 1, bad))
                break
    policy.check_faults(wait=True)
    print('%d launches of the gradient call: %s' % (n, 'all identical to the first' if not bad else 'MISMATCH'))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
