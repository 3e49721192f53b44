"""Guard of the in-kernel Q' hand-off of the row-local update (include/curious_hip.h, curious_workspace_fault_offset): the
fault word travels to the host asynchronously, a faulted update froze the parameters, the verdict is raised as
HandoffFault.  Mixed into curious_amd.ddpg.DDPG."""
import torch

from curious_amd import _lib, dist, ops

FAULT_CHECK_EVERY = 8   # cycles between asynchronous reads of the hand-off fault word (DDPG.update_target_net)


class HandoffFault(_lib.CuriousHipError):
    """A consumer of the in-kernel Q' hand-off of the row-local update gave up (include/curious_hip.h,
    curious_workspace_fault_offset).  The optimiser skipped every update since: parameters, moments and target are those
    of the last good update."""


class FaultsMixin:
    def _enqueue_fault_check(self):
        """Asynchronous D2H copy of the workspace's fault word, stream-ordered behind everything enqueued so far."""
        if getattr(self, '_fault', None) is None:
            self._fault = ops.fault_word(self.net_cfg, self._Bt, self._workspace)
            self._fault_pin = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._fault_ev = torch.cuda.Event()
        self._fault_pin.copy_(self._fault, non_blocking=True)
        self._fault_ev.record()
        self._fault_pending = True

    def check_faults(self, wait=True):
        """Raises HandoffFault when a consumer of Q' gave up in an update since the last check.  wait=False looks only
        at a copy that has already arrived (the training loop: the verdict of cycle c is read during cycle c + 1);
        wait=True enqueues a fresh copy and waits for it.  The word is cleared before raising, so a caller that catches
        the exception can go on training from the last good parameters."""
        if wait:
            self._enqueue_fault_check()
            self._ipc_verdict(wait=True)                             # (fused IPC all-reduce: a wait for a peer gave up)
        elif dist.is_distributed():
            return                                                   # read at a fixed cycle count: update_target_net
        self._fault_verdict(wait)

    def _fault_verdict(self, wait):
        if not getattr(self, '_fault_pending', False):
            return
        if wait:
            self._fault_ev.synchronize()
        elif not self._fault_ev.query():
            return
        self._fault_pending = False
        n = int(self._fault_pin[0])
        if n:
            ops.fault_word(self.net_cfg, self._Bt, self._workspace, 64).zero_()
            raise HandoffFault("%d consumer wave(s) of the row-local update never received Q' from their target group "
                               '(agent %s, rank %d): the optimiser was skipped from that update on' %
                               (n, self.scope, dist.rank()))
