"""One process per GPU over RCCL/xGMI: the collectives that replace the reference's mpi4py calls.

Reference call sites: C1/C2 MpiAdam Allreduce(SUM) mpi_adam.py:26; C3 Bcast mpi_adam.py:39; C4
check_synced mpi_adam.py:42-50; C5 Normalizer._mpi_average normalizer.py:84-94; C9 competence gathers
rollout.py:332-336; C11 mpi_moments.py:16.  `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used for
the CPU tests of this layer.  With a single process every function is the identity.
"""
import os

import numpy as np
import torch
import torch.distributed as td


RANK_SEED_STRIDE = 1000003   # what separates the device RNG keys of consecutive (global) ranks


class RankDivergence(RuntimeError):
    """MpiAdam.check_synced (mpi_adam.py:42-50): the ranks no longer hold bit-identical parameters."""


def _forced():
    """CURIOUS_FORCE_DIST=1: take the multi-rank code paths (RCCL communicator, split update graphs, collectives) even
    with WORLD_SIZE=1 -- lets a single-GPU box exercise them against the real RCCL library."""
    return os.environ.get('CURIOUS_FORCE_DIST', '0') == '1'


def is_distributed():
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _forced())


def rank():
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def world_size():
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


class _stdout_to_stderr:
    """gloo announces every process group on the process's STDOUT from C++ ("[Gloo] Rank 0 is connected to ...").  A job
    whose stdout is a protocol (bench.py: ONE JSON line) must not carry that: while a group is being created, file
    descriptor 1 points at stderr."""

    def __enter__(self):
        import sys
        try:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        except OSError:
            self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  No-op for 1 rank."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not _forced()) or (td.is_available() and td.is_initialized()):
        return
    if backend is None:
        # CURIOUS_DIST_BACKEND=gloo lets several ranks share one GPU (functional testing of the N > 1 path on a
        # single-GPU box; RCCL refuses two ranks on one device)
        backend = os.environ.get('CURIOUS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device_index())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    with _stdout_to_stderr():
        td.init_process_group(backend=backend)
        host_group()                                                 # (collective: every rank creates it here)
        if td.get_backend() == 'gloo' or _HOST_GROUP:
            # gloo connects (and announces itself) lazily, at the first collective of a group: have that happen here
            t = torch.zeros(1)
            td.all_reduce(t, group=host_group())


_HOST_GROUP = None


def host_group():
    """Process group for the small HOST-side exchanges (competence records C9, exploit flags, log scalars C11, pickled
    objects C12): gloo over TCP, CPU tensors.  With RCCL as the main backend these would otherwise be GPU collectives
    whose results the host can only read after a stream synchronisation -- i.e. after every update already enqueued --
    while the host-side form just blocks Python, which runs ahead of the GPU anyway.  None = use the default group (the
    main backend is gloo already, or the side group could not be created)."""
    global _HOST_GROUP
    if _HOST_GROUP is None and td.is_available() and td.is_initialized():
        if td.get_backend() == 'gloo':
            _HOST_GROUP = False
        else:
            try:
                _HOST_GROUP = td.new_group(backend='gloo')
            except Exception:                                        # no gloo in this build: GPU collectives it is
                _HOST_GROUP = False
    return _HOST_GROUP or None


def local_device_index():
    """GPU of this rank: LOCAL_RANK, folded onto the visible devices."""
    n = max(1, torch.cuda.device_count())
    return int(os.environ.get('LOCAL_RANK', '0')) % n


def allreduce_sum_(t):
    """In-place SUM over ranks of a tensor (GPU: RCCL all-reduce on the current stream)."""
    if is_distributed():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def broadcast_(t, root=0):
    if is_distributed():
        td.broadcast(t, src=root)
    return t


def allgather(t):
    """[n, ...] per rank -> [world*n, ...] in rank order."""
    if not is_distributed():
        return t
    out = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(out, t.contiguous())
    return torch.cat(out, dim=0)


def _comm_device():
    if is_distributed() and td.get_backend() == 'nccl' and host_group() is None:
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def allreduce_sum_numpy(x):
    """Small host arrays (log scalars, C11): on the host-side group."""
    if not is_distributed():
        return x
    t = torch.as_tensor(np.asarray(x, dtype=np.float64)).to(_comm_device())
    td.all_reduce(t, op=td.ReduceOp.SUM, group=host_group())
    return t.cpu().numpy()


def allgather_numpy(x, uneven=False):
    """[n, ...] host array per rank -> [sum of the ranks' n, ...] in rank order (C9): on the host-side group.  uneven: the
    ranks' n may differ (processes that stand for different numbers of virtual ranks, virtual_layout): the lengths go
    round first, the blocks travel padded to the longest."""
    if not is_distributed():
        return np.asarray(x)
    x = np.ascontiguousarray(x)
    if uneven:
        lens = allgather_numpy(np.array([x.shape[0]], np.int64)).astype(np.int64)
        m = int(lens.max())
        if x.shape[0] < m:
            x = np.concatenate([x, np.zeros((m - x.shape[0],) + x.shape[1:], x.dtype)])
    t = torch.as_tensor(x).to(_comm_device())
    out = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(out, t.contiguous(), group=host_group())
    if uneven:
        return torch.cat([o[:int(n)] for o, n in zip(out, lens)], dim=0).cpu().numpy()
    return torch.cat(out, dim=0).cpu().numpy()


def virtual_layout(num_cpu, world=None, r=None):
    """--num_cpu R on W processes: process r stands for V_r of the reference's ranks -- R // W of them, one more on the first
    R % W processes -- starting at global rank base_r; exactly R ranks in all (readme.md:16 publishes R = 19: on 8 GPUs
    3 + 3 + 3 + 2 + 2 + 2 + 2 + 2).  Returns (V_r, base_r, R); R <= W: (1, r, W)."""
    world = world_size() if world is None else int(world)
    r = rank() if r is None else int(r)
    num_cpu = int(num_cpu)
    if num_cpu <= world:
        return 1, r, world
    q, rem = divmod(num_cpu, world)
    return q + (1 if r < rem else 0), r * q + min(r, rem), num_cpu


def host_any(flag):
    """True on every rank iff `flag` is true on at least one rank (a host-side exchange: the GPU is not involved)."""
    if not is_distributed():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32).to(_comm_device())
    td.all_reduce(t, op=td.ReduceOp.MAX, group=host_group())
    return bool(int(t.item()))


def broadcast_object(obj, root=0):
    if not is_distributed():
        return obj
    box = [obj]
    if host_group() is not None:
        td.broadcast_object_list(box, src=root, group=host_group(), device=torch.device('cpu'))
    else:
        td.broadcast_object_list(box, src=root)
    return box[0]


def allgather_object(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (pickled, host-side group)."""
    if not is_distributed():
        return [obj]
    out = [None] * world_size()
    td.all_gather_object(out, obj, group=host_group())
    return out


def barrier():
    if is_distributed():
        td.barrier()


_CAPTURED_OK = None


def captured_allreduce_ok():
    """May the gradient all-reduce be captured inside the update hipGraph (DDPG.train_batches)?  CURIOUS_GRAPH_ALLREDUCE
    = 1 / 0 forces the answer; unset, the ranks find out together once per process: every rank captures a small RCCL
    all-reduce in a graph, replays it three times and compares the result with the known sum; the ranks then agree on
    the minimum of their verdicts (an eager all-reduce), so either all of them chain their updates or none does.  A
    watchdog ends the process if the replayed collective never completes (a hang is not recoverable in-process)."""
    global _CAPTURED_OK
    env = os.environ.get('CURIOUS_GRAPH_ALLREDUCE', 'auto')
    if env in ('0', '1'):
        return env == '1'
    if _CAPTURED_OK is not None:
        return _CAPTURED_OK
    if not is_distributed() or td.get_backend() != 'nccl':
        _CAPTURED_OK = False
        return False
    import threading
    dev = torch.device('cuda', torch.cuda.current_device())
    ws, rk = world_size(), rank()
    watchdog = threading.Timer(120.0, lambda: os._exit(17))
    watchdog.daemon = True
    watchdog.start()
    ok = False
    try:
        x = torch.full([4096], float(rk + 1), device=dev)
        y = torch.empty_like(x)
        y.copy_(x)
        td.all_reduce(y)                                             # communicator warm-up outside any capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            y.copy_(x)
            td.all_reduce(y)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ok = bool((y == float(ws * (ws + 1) // 2)).all())
        del g
    except Exception:                                                # any failure: the eager collective is used
        ok = False
    finally:
        watchdog.cancel()
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    td.all_reduce(flag, op=td.ReduceOp.MIN)
    _CAPTURED_OK = bool(flag.item() == 1.0)
    return _CAPTURED_OK
