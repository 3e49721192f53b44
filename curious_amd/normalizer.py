"""Running mean/std normaliser on the GPU.  Mirrors Normalizer baselines/her/normalizer.py:10-118.

Same constructor and methods (`sess` is accepted and ignored -- there is no TF session).  State lives in two device
vectors: acc = [local_sum | local_sumsq | local_count] and state = [sum | sumsq | count | mean | std] (count starts
at 1, normalizer.py:37-39).  `recompute_stats` replaces the three MPI all-reduces per normaliser
(normalizer.py:84-94) by one RCCL all-reduce of `acc` (or of a buffer shared by several normalisers, see
`recompute_many`) followed by a device-side divide by the world size (MEAN over ranks, as in the reference).
"""
import numpy as np
import torch

from curious_amd import dist, ops


class Normalizer:
    def __init__(self, size, eps=1e-2, default_clip_range=np.inf, sess=None, _acc=None, _state=None):
        self.size = size
        self.eps = eps
        self.default_clip_range = default_clip_range
        dev = torch.device('cuda', torch.cuda.current_device())
        self.acc = _acc if _acc is not None else torch.zeros(2 * size + 1, dtype=torch.float32, device=dev)
        # (_acc / _state: storage handed in by the owner -- DDPG packs both normalisers' accumulators into one buffer,
        #  an ExpertBank carves every expert's state out of its slab row)
        self.state = _state if _state is not None else torch.zeros(4 * size + 1, dtype=torch.float32, device=dev)
        self.state.zero_()
        self.state[2 * size] = 1.0                                   # count_tf = ones (normalizer.py:37-39)
        self.state[3 * size + 1:] = 1.0                              # std = ones     (normalizer.py:43-45)
        self._scratch = None

    # views with the reference's attribute names
    @property
    def local_sum(self): return self.acc[:self.size]
    @property
    def local_sumsq(self): return self.acc[self.size:2 * self.size]
    @property
    def local_count(self): return self.acc[2 * self.size:]
    @property
    def mean(self): return self.state[2 * self.size + 1:3 * self.size + 1]
    @property
    def std(self): return self.state[3 * self.size + 1:]

    def update(self, v):
        """v: GPU tensor view [..., size] (rows may be strided views of a packed batch) or a NumPy array."""
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(np.ascontiguousarray(np.asarray(v, dtype=np.float32))).to(self.acc.device)
        v = v.reshape(-1, self.size) if v.is_contiguous() else v
        assert v.dim() == 2 and v.shape[1] == self.size and v.stride(1) == 1
        n = v.shape[0]
        need = ops.norm_scratch_doubles(n, self.size)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float64, device=self.acc.device)
        # rows = v's storage with its own row stride; column offset 0 relative to v's data pointer
        ops.norm_update(v, n, v.stride(0), 0, self.size, self.acc, self._scratch)

    def recompute_stats(self):
        dist.allreduce_sum_(self.acc)                                # normalizer.py:84-94 (SUM, then / size)
        ops.norm_recompute(self.acc, self.state, self.size, dist.world_size(), self.eps)

    def normalize(self, v, clip_range=None):
        if clip_range is None:
            clip_range = self.default_clip_range
        v = torch.as_tensor(v, dtype=torch.float32, device=self.state.device)
        return torch.clamp((v - self.mean) / self.std, -clip_range, clip_range)        # normalizer.py:72-77

    def denormalize(self, v):
        v = torch.as_tensor(v, dtype=torch.float32, device=self.state.device)
        return self.mean + v * self.std


def recompute_many(normalizers, packed=None, ranks_per_process=1, total_ranks=None):
    """recompute_stats for several normalisers with ONE all-reduce: their `acc` vectors must be slices of one
    device buffer (DDPG allocates o_stats / g_stats that way).  packed: that buffer (required when it is
    itself a view of something larger -- an expert's slab row).  ranks_per_process: virtual ranks (DDPG.virtual_ranks) --
    the accumulators of a process already hold the sum over its virtual ranks, the mean is over world x that many --
    or over total_ranks when the job's ranks are laid out unevenly over the processes (dist.virtual_layout)."""
    if packed is None:
        packed = normalizers[0].acc._base if normalizers[0].acc._base is not None else normalizers[0].acc
        for nz in normalizers:
            assert (nz.acc._base if nz.acc._base is not None else nz.acc) is packed
    lo, hi = packed.data_ptr(), packed.data_ptr() + 4 * packed.numel()
    for nz in normalizers:
        assert lo <= nz.acc.data_ptr() and nz.acc.data_ptr() + 4 * nz.acc.numel() <= hi
    assert packed.numel() == sum(nz.acc.numel() for nz in normalizers)
    dist.allreduce_sum_(packed)
    ws = int(total_ranks) if total_ranks else dist.world_size() * int(ranks_per_process)
    for nz in normalizers:
        ops.norm_recompute(nz.acc, nz.state, nz.size, ws, nz.eps)


class IdentityNormalizer:
    """normalizer.py:121-140."""

    def __init__(self, size, std=1.):
        self.size = size
        dev = torch.device('cuda', torch.cuda.current_device())
        self.mean = torch.zeros(size, device=dev)
        self.std = std * torch.ones(size, device=dev)

    def update(self, x):
        pass

    def normalize(self, x, clip_range=None):
        return x / self.std

    def denormalize(self, x):
        return self.std * x

    def synchronize(self):
        pass

    def recompute_stats(self):
        pass
