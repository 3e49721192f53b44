"""Peer-mapped device memory for the fused IPC all-reduce + Adam (csrc/ipc.hip, DDPG(_allreduce='ipc')).

One block per rank holds what the peers read and write -- [theta | grad | flag block] --, allocated with hipMalloc
through the library (curious_ipc_alloc), exported as a 64-byte handle, exchanged over the host-side process group and
imported by every peer.  torch sees the block as ordinary tensors (aliases through __cuda_array_interface__).
"""
import ctypes as C

import torch

from curious_amd import _lib, dist


class _Alias:
    """A region of device memory for torch.as_tensor (no ownership)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=2)


class IpcBlock:
    """floats: sizes of the float32 regions, in order; a [2][8] uint32 flag block follows them."""

    FLAG_WORDS = 16

    def __init__(self, floats):
        self.sizes = [int(n) for n in floats]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 63) & ~63
        self.flag_off = off
        self.bytes = 4 * (off + 64)
        p = C.c_void_p()
        _lib.check(_lib.lib().curious_ipc_alloc(self.bytes, C.byref(p)), 'curious_ipc_alloc')
        self.base = p.value
        self.peers = None                                            # [rank] -> base pointer of that rank's block here
        self._keep = []

    def tensor(self, i):
        t = torch.as_tensor(_Alias(self.base + 4 * self.offsets[i], self.sizes[i], '<f4'), device='cuda')
        self._keep.append(t)
        return t

    def flags(self):
        t = torch.as_tensor(_Alias(self.base + 4 * self.flag_off, self.FLAG_WORDS, '<i4'), device='cuda')
        self._keep.append(t)
        return t

    def connect(self):
        """Collective: every rank exports its block and imports every peer's."""
        if self.peers is not None:
            return
        L = _lib.lib()
        h = C.create_string_buffer(64)
        _lib.check(L.curious_ipc_export(C.c_void_p(self.base), h), 'curious_ipc_export')
        handles = dist.allgather_object(h.raw)
        self.peers = []
        for r, raw in enumerate(handles):
            if r == dist.rank():
                self.peers.append(self.base)
                continue
            p = C.c_void_p()
            _lib.check(L.curious_ipc_import(raw, C.byref(p)), 'curious_ipc_import')
            self.peers.append(p.value)
        dist.barrier()

    def peer_ptr(self, r, i):
        return self.peers[r] + 4 * self.offsets[i]

    def peer_flags(self, r):
        return self.peers[r] + 4 * self.flag_off

    def disconnect(self):
        """Collective: unmap the peers' blocks; no owner goes on before every mapping of its block is gone."""
        if self.peers is None:
            return
        torch.cuda.synchronize()
        dist.barrier()
        for r, p in enumerate(self.peers):
            if r != dist.rank():
                _lib.check(_lib.lib().curious_ipc_close(C.c_void_p(p)), 'curious_ipc_close')
        self.peers = None
        dist.barrier()
