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

The three HOT entry points take the reference's table-shaped arguments (record / batch layouts, task tables, sampler
descriptions) -- the dispatcher's schema language has no struct type.  They get an opaque DESCRIPTOR instead (round 4):
`desc_create(...)` files the structs once and returns an int64 handle, the ops then take the handle + tensors:

    d = torch_ops.desc_create(layout=..., tasks=..., params=..., rng=..., buf_stride=..., n=...)
    torch.ops.curious_hip.her_sample(d, storage, batch)                                   # her.py:99-183, ddpg.py:326-353
    d = torch_ops.desc_create(cfg=..., layout=..., B=..., tab_base=..., tasks=..., params=..., rng=..., buf_stride=...)
    torch.ops.curious_hip.ddpg_update(d, theta, theta_target, batch, workspace, grad, losses, Q_pi, m, v, step_ctr,
                                      alpha_tab, next_batch, storage, params_unchanged)   # ddpg.py:235-248 + mpi_adam
    d = torch_ops.desc_create(cfg=..., ecfg=..., layout=..., n=..., clip_obs=..., noise_scale=..., random_eps=..., seed=...,
                              counter=..., env_id0=..., t0=..., nsteps=..., reward_eps=..., relative_goals=...)
    torch.ops.curious_hip.policy_rollout(d, theta, workspace, u_out, counter_base, episode, tasks, o, ag, g, td, staging,
                                         flags)                                            # rollout.py:226-303 x T

The remaining struct-carrying entry points (`curious_store_episodes`, the env reset / step kernels, the batched experts)
stay on the ctypes binding (curious_amd/ops.py).  Both faces call the same symbols; there is no second implementation.
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


# ------------------------------------------------------------------ descriptor-carrying ops (the three hot entry points)
_DESCS = {}


def desc_create(**fields):
    """File the struct-shaped arguments of a hot entry point (layouts, task tables, sampler / env descriptions, scalars)
    and return an int64 handle for the ops below.  The descriptor keeps what it is given alive (e.g. the device tables a
    SampleRng points into, through `keep=`)."""
    h = (max(_DESCS) + 1) if _DESCS else 1
    _DESCS[h] = dict(fields)
    return h


def desc_free(handle):
    _DESCS.pop(int(handle), None)


def _desc(handle):
    try:
        return _DESCS[int(handle)]
    except KeyError:
        raise _lib.CuriousHipError('unknown descriptor handle %r (torch_ops.desc_create)' % (handle,))


@torch.library.custom_op(_NS + '::her_sample', mutates_args=('batch',), device_types='cuda')
def her_sample(desc: int, storage: torch.Tensor, batch: torch.Tensor) -> None:
    d = _desc(desc)                                                                  # her.py:99-183, ddpg.py:326-353
    ops.her_sample(storage, d['buf_stride'], d['layout'], d['tasks'], d['params'], d['n'], batch, plan=d.get('plan'),
                   rng=d.get('rng'))


@torch.library.custom_op(_NS + '::ddpg_update',
                         mutates_args=('theta', 'workspace', 'grad', 'losses', 'Q_pi', 'm', 'v', 'step_ctr', 'next_batch'),
                         device_types='cuda')
def ddpg_update(desc: int, theta: torch.Tensor, theta_target: torch.Tensor, batch: torch.Tensor, workspace: torch.Tensor,
                grad: torch.Tensor, losses: torch.Tensor, Q_pi: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                step_ctr: torch.Tensor, alpha_tab: torch.Tensor, next_batch: torch.Tensor, storage: torch.Tensor,
                params_unchanged: bool) -> None:
    d = _desc(desc)                                                                  # ddpg.py:235-248, mpi_adam.py:29-35
    ops.ddpg_update(d['cfg'], theta, theta_target, batch, d['layout'], d['B'], workspace, grad, losses, Q_pi, m, v,
                    step_ctr=step_ctr, alpha_tab=alpha_tab, tab_base=d.get('tab_base', 0), o_stats=d.get('o_stats'),
                    g_stats=d.get('g_stats'), next_batch=next_batch, storage=storage, buf_stride=d['buf_stride'],
                    tasks=d['tasks'], params=d['params'], rng=d['rng'], params_unchanged=params_unchanged)


@torch.library.custom_op(_NS + '::policy_rollout',
                         mutates_args=('workspace', 'u_out', 'episode', 'o', 'ag', 'staging', 'flags'), device_types='cuda')
def policy_rollout(desc: int, theta: torch.Tensor, workspace: torch.Tensor, u_out: torch.Tensor,
                   counter_base: torch.Tensor, episode: torch.Tensor, tasks: torch.Tensor, o: torch.Tensor,
                   ag: torch.Tensor, g: torch.Tensor, td: torch.Tensor, staging: torch.Tensor, flags: torch.Tensor) -> None:
    d = _desc(desc)                                                                  # rollout.py:226-303 for every env
    ops.policy_rollout(d['cfg'], theta, d['n'], d['clip_obs'], workspace, d['noise_scale'], d['random_eps'], d['seed'],
                       d['counter'], u_out, d['ecfg'], d['layout'], d['env_id0'], episode, tasks, d['t0'], d['nsteps'],
                       o, ag, g, td, staging, d['reward_eps'], counter_base=counter_base, flags=flags,
                       o_stats=d.get('o_stats'), g_stats=d.get('g_stats'),
                       relative_goals=bool(d.get('relative_goals', False)))
