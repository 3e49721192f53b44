"""Data-parallel Adam on a flat device vector.  Mirrors MpiAdam baselines/common/mpi_adam.py:6-50.

`var_list` is replaced by the flat parameter vector itself (a GPU tensor, or a slice of the fused
[theta_Q | theta_pi] vector); `comm` by the RCCL process group.  update() = all-reduce(SUM) of the local gradient
(mpi_adam.py:26; not averaged unless scale_grad_by_procs) + the fused Adam kernel; sync() = broadcast from rank 0;
check_synced() compares a 128-bit checksum of the parameter bits instead of broadcasting the whole vector.
"""
import numpy as np
import torch

from curious_amd import dist, ops


class MpiAdam:
    def __init__(self, var_list, *, beta1=0.9, beta2=0.999, epsilon=1e-08, scale_grad_by_procs=True, comm=None):
        assert isinstance(var_list, torch.Tensor) and var_list.is_cuda and var_list.dim() == 1, \
            'pass the flat GPU parameter vector'
        self.theta = var_list
        self.var_list = var_list
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.scale_grad_by_procs = scale_grad_by_procs
        self.m = torch.zeros_like(var_list)
        self.v = torch.zeros_like(var_list)
        self.t = 0
        self.comm = comm

    def getflat(self):
        return self.theta

    def setfromflat(self, x):
        self.theta.copy_(torch.as_tensor(x, dtype=torch.float32))

    def alpha(self, stepsize, t=None):
        return ops.adam_alpha(stepsize, self.t if t is None else t, self.beta1, self.beta2)

    def alpha_table(self, stepsize, ts):
        """alpha(stepsize, t) for every t of `ts` (float32 array, the same values)."""
        return ops.adam_alpha_table(stepsize, ts, self.beta1, self.beta2)

    def update(self, localg, stepsize):
        if self.t % 100 == 0:                                        # mpi_adam.py:22-23
            self.check_synced()
        g = localg if isinstance(localg, torch.Tensor) else torch.as_tensor(np.asarray(localg, dtype=np.float32))
        g = g.to(self.theta.device, dtype=torch.float32)
        if dist.is_distributed():
            g = g.clone()
            dist.allreduce_sum_(g)                                   # mpi_adam.py:26
            if self.scale_grad_by_procs:
                g /= dist.world_size()
        self.t += 1
        ops.adam_update(self.theta, self.m, self.v, g, self.theta.numel(), 0, self.alpha(stepsize), 0.0,
                        self.beta1, self.beta2, self.epsilon)

    def sync(self):
        dist.broadcast_(self.theta, 0)                               # mpi_adam.py:37-40

    def check_synced(self):
        """mpi_adam.py:42-50 without moving the parameters: all ranks must hold bit-identical vectors."""
        if not dist.is_distributed():
            return
        out = torch.zeros(2, dtype=torch.int64, device=self.theta.device)
        ops.param_checksum(self.theta, out)
        mine = out.clone()
        dist.broadcast_(out, 0)
        if not torch.equal(out, mine):                               # an exception, not an assert: survives python -O
            raise dist.RankDivergence('parameters diverged between ranks (rank %d)' % dist.rank())
