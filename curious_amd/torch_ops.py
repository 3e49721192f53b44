"""PyTorch custom-op face of the C ABI: `torch.ops.curious_hip.*` (SURVEY 8b names a `TORCH_LIBRARY(curious_hip, ...)`
layer).

The entry points whose arguments are tensors and scalars are registered with `torch.library.custom_op` over the same
`libcurious_hip.so` the ctypes binding loads (curious_amd/_lib.py), so callers get the dispatcher's type / device checks,
`mutates_args` aliasing information and the current stream of the calling context for free:

    torch.ops.curious_hip.polyak_update(target, main, 0.95)
    torch.ops.curious_hip.adam_update(theta, m, v, grad, n_Q, n_pi, alpha_Q, alpha_pi)
    torch.ops.curious_hip.param_checksum(theta)                       -> int64[2]
    torch.ops.curious_hip.norm_update(rows, col_off, dim, acc)
    torch.ops.curious_hip.norm_recompute(acc, state, world_size, eps)
    torch.ops.curious_hip.policy_forward(cfg_i, cfg_f, theta, o, g, td, clip_obs, compute_Q) -> (pi, Q)
    torch.ops.curious_hip.ddpg_grads(cfg_i, cfg_f, theta, theta_target, batch, batch_layout, grad) -> (losses, Q_pi)

Entry points that take the reference's table-shaped arguments (record / batch layouts, task tables, sample plans:
`curious_her_sample`, `curious_store_episodes`, `curious_ddpg_update*`, the env kernels) stay on the struct-carrying
ctypes binding (curious_amd/ops.py): the dispatcher schema language has no struct type, and flattening ~60 layout
fields into integer lists per call would only move the unchecked part from pointers to list positions.  Both faces
call the same symbols; there is no second implementation.
"""
import ctypes as C

import numpy as np
import torch

from curious_amd import _lib, ops
from curious_amd._lib import check, lib, ptr

_NS = 'curious_hip'


def _cfg(cfg_i, cfg_f):
    """cfg_i = [dimo, dimg, dimu, dimtd, hidden, layers, modular, clip_pos_returns, normalize_obs],
    cfg_f = [max_u, gamma, clip_return, action_l2, norm_clip]  (curious_net_cfg_t)."""
    dimo, dimg, dimu, dimtd, hidden, layers, modular, clip_pos, norm_obs = [int(x) for x in cfg_i]
    max_u, gamma, clip_return, action_l2, norm_clip = [float(x) for x in cfg_f]
    return ops.make_net_cfg(dimo, dimg, dimu, dimtd, hidden, layers, modular, max_u, gamma, clip_return, action_l2,
                            clip_pos, norm_obs, norm_clip)


@torch.library.custom_op(_NS + '::polyak_update', mutates_args=('target',), device_types='cuda')
def polyak_update(target: torch.Tensor, main: torch.Tensor, polyak: float) -> None:
    ops.polyak_update(target, main, polyak)                                          # ddpg.py:459-462


@torch.library.custom_op(_NS + '::adam_update', mutates_args=('theta', 'm', 'v'), device_types='cuda')
def adam_update(theta: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor, n_Q: int, n_pi: int,
                alpha_Q: float, alpha_pi: float) -> None:
    ops.adam_update(theta, m, v, grad, n_Q, n_pi, alpha_Q, alpha_pi)                 # mpi_adam.py:29-35


@torch.library.custom_op(_NS + '::param_checksum', mutates_args=(), device_types='cuda')
def param_checksum(theta: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(2, dtype=torch.int64, device=theta.device)
    ops.param_checksum(theta, out)                                                   # mpi_adam.py:42-50
    return out


@param_checksum.register_fake
def _(theta):
    return theta.new_empty(2, dtype=torch.int64)


@torch.library.custom_op(_NS + '::norm_update', mutates_args=('acc',), device_types='cuda')
def norm_update(rows: torch.Tensor, col_off: int, dim: int, acc: torch.Tensor) -> None:
    n = rows.shape[0]
    scratch = torch.empty(ops.norm_scratch_doubles(n, dim), dtype=torch.float64, device=rows.device)
    ops.norm_update(rows, n, rows.stride(0), col_off, dim, acc, scratch)             # normalizer.py:64-70


@torch.library.custom_op(_NS + '::norm_recompute', mutates_args=('acc', 'state'), device_types='cuda')
def norm_recompute(acc: torch.Tensor, state: torch.Tensor, world_size: float, eps: float) -> None:
    ops.norm_recompute(acc, state, (state.numel() - 1) // 4, world_size, eps)        # normalizer.py:96-118


@torch.library.custom_op(_NS + '::policy_forward', mutates_args=(), device_types='cuda')
def policy_forward(cfg_i: list[int], cfg_f: list[float], theta: torch.Tensor, o: torch.Tensor, g: torch.Tensor,
                   td: torch.Tensor, clip_obs: float, compute_Q: bool) -> tuple[torch.Tensor, torch.Tensor]:
    cfg = _cfg(cfg_i, cfg_f)
    n = o.shape[0]
    ws = torch.empty(ops.workspace_floats(cfg, n), dtype=torch.float32, device=o.device)
    pi = torch.empty([n, cfg.dimu], dtype=torch.float32, device=o.device)
    Q = torch.empty([n, 1], dtype=torch.float32, device=o.device)
    ops.policy_forward(cfg, theta, o, g, td if cfg.dimtd > 0 else None, n, clip_obs, ws, pi,
                       Q if compute_Q else None)                                     # ddpg.py:129-146
    return pi, Q


@policy_forward.register_fake
def _(cfg_i, cfg_f, theta, o, g, td, clip_obs, compute_Q):
    return o.new_empty([o.shape[0], int(cfg_i[2])]), o.new_empty([o.shape[0], 1])


@torch.library.custom_op(_NS + '::ddpg_grads', mutates_args=('grad',), device_types='cuda')
def ddpg_grads(cfg_i: list[int], cfg_f: list[float], theta: torch.Tensor, theta_target: torch.Tensor,
               batch: torch.Tensor, batch_layout: list[int], grad: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """batch_layout = [off_o, off_td, off_u, off_g, off_o2, off_g2, off_r, off_ag, off_ag2, off_extra, stride]
    (curious_batch_layout_t).  Returns ([Q_loss, pi_loss], Q_pi[B, 1]); `grad` receives [Q_grad | pad | pi_grad]
    (ddpg.py:235-243)."""
    cfg = _cfg(cfg_i, cfg_f)
    B = batch.shape[0]
    BL = _lib.BatchLayout()
    for name, val in zip([f[0] for f in _lib.BatchLayout._fields_], batch_layout):
        setattr(BL, name, int(val))
    ws = torch.zeros(ops.workspace_floats(cfg, B), dtype=torch.float32, device=batch.device)   # (holds the fault word)
    losses = torch.zeros(2, dtype=torch.float32, device=batch.device)
    Q_pi = torch.zeros([B, 1], dtype=torch.float32, device=batch.device)
    check(lib().curious_ddpg_grads(C.byref(cfg), ptr(theta), ptr(theta_target), ptr(batch), C.byref(BL), int(B), None,
                                   None, ptr(ws), ptr(grad), ptr(losses), ptr(Q_pi), None, 0, None,
                                   _lib.current_stream()),
          'curious_ddpg_grads')
    return losses, Q_pi


@ddpg_grads.register_fake
def _(cfg_i, cfg_f, theta, theta_target, batch, batch_layout, grad):
    return batch.new_empty(2), batch.new_empty([batch.shape[0], 1])
