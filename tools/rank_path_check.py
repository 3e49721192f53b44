"""Prints a digest of the parameters after 35 updates of the bench job.  Run it plain, with CURIOUS_FORCE_DIST=1 (split
graphs + eager RCCL all-reduce on a one-rank communicator) and with CURIOUS_GRAPH_ALLREDUCE=1 on top (all-reduce
captured in the update graph): the three digests must be equal (tests/test_gpu_agent.py::test_rank_paths...)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from curious_amd import dist
    dist.init_from_env()
    torch.cuda.set_device(dist.local_device_index())
    import numpy as np
    np.random.seed(1234 + 1000000 * dist.rank())                # train.py:242 (the rollouts' task / goal draws)
    if os.environ.get('CURIOUS_RANK_CHECK_STRUCTURE') == 'task_experts':
        return experts_main()
    V = int(os.environ.get('CURIOUS_RANK_CHECK_V', '1'))        # virtual ranks per process (DDPG virtual_ranks)
    params, dims, policy, worker = bench.build_job(use_graph=os.environ.get('CURIOUS_RANK_CHECK_NOGRAPH', '0') != '1',
                                                   b_r=2 if V > 1 else 256, virtual_ranks=V)
    bench.prefill(policy, 256 if V == 1 else 2048, seed=dist.rank())
    for _ in range(5):
        policy.train()
    policy.train_batches(30)
    for _ in range(int(os.environ.get('CURIOUS_RANK_CHECK_EXTRA', '0'))):      # (debugging aid: runs of 100 updates, no rollout)
        policy.train_batches(100)
    # optional: a guarded run of 6 updates (DDPG.train_batches_guarded, --fault_check sync) in which rank 1 loses a producer
    # of Q' on the FIRST attempt ('inject') or not at all ('clean'): every rank sees the collective flag, every rank replays
    # the run from the state it started with -- the digest must be the clean run's
    guarded = os.environ.get('CURIOUS_RANK_CHECK_GUARDED')
    if guarded:
        from curious_amd import ops
        calls, plain = [0], policy.train_batches

        def train_batches(n):
            calls[0] += 1
            if guarded == 'inject' and calls[0] == 1 and dist.rank() == 1:
                with ops.option('fault_inject', 3), ops.option('qt_spins', 20000):
                    out = plain(n)
                    torch.cuda.synchronize()
                    return out
            return plain(n)
        policy.train_batches = train_batches
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            policy.train_batches_guarded(6)
        policy.train_batches = plain
        assert calls[0] == (2 if guarded == 'inject' else 1), calls
        policy.train_batches(4)
    # optional: whole cycles (rollout + store + updates) so that ranks diverge in data and must agree through collectives
    for _ in range(int(os.environ.get('CURIOUS_RANK_CHECK_CYCLES', '0'))):
        bench.cycle(policy, worker)
    torch.cuda.synchronize()
    h = hashlib.sha256(policy.theta.cpu().numpy().tobytes()).hexdigest()
    line = 'DIGEST %s %d %r captured=%s\n' % (h, int(policy._step_ctr), float(policy._losses[0]),
                                              dist.captured_allreduce_ok() if dist.is_distributed() else None)
    out = os.environ.get('CURIOUS_RANK_CHECK_OUT')
    if out:                                     # one file per rank: ranks of one launcher share (and interleave on) stdout
        with open('%s.rank%d' % (out, dist.rank()), 'w') as f:
            f.write(line)
    sys.stdout.write(line)
    sys.stdout.flush()
    # clean shutdown: captured graphs are dropped before the process group is destroyed
    from curious_amd.experiment.train import shutdown
    shutdown([policy])


def experts_main():
    """The same for the batched task experts (BASELINE configs[4]): 4 experts, every update of all of them in one launch
    sequence; with several ranks ONE all-reduce of the [4, P] gradient block per update."""
    from curious_amd import dist
    params, dims, bank, workers = bench.build_experts_job(use_graph=os.environ.get('CURIOUS_RANK_CHECK_GRAPH', '1') != '0')
    bench.prefill(bank[0], 256, seed=dist.rank())
    bank.train_batches(3)                                     # singles
    bank.train_batches(32)                                    # chains of 10 + singles
    for k in range(int(os.environ.get('CURIOUS_RANK_CHECK_CYCLES', '0'))):
        bench.experts_cycle(bank, workers, k)
    torch.cuda.synchronize()
    bank.check_faults()
    assert bank.batched
    h = hashlib.sha256()
    for x in bank:
        h.update(x.theta.cpu().numpy().tobytes())
    line = 'DIGEST %s %d %r captured=%s\n' % (h.hexdigest(), int(bank[0]._step_ctr), float(bank[3]._losses[0]),
                                              dist.captured_allreduce_ok() if dist.is_distributed() else None)
    out = os.environ.get('CURIOUS_RANK_CHECK_OUT')
    if out:
        with open('%s.rank%d' % (out, dist.rank()), 'w') as f:
            f.write(line)
    sys.stdout.write(line)
    sys.stdout.flush()
    from curious_amd.experiment.train import shutdown
    shutdown(list(bank), bank)


if __name__ == '__main__':
    main()
