"""PyTorch dispatcher face of the C ABI: `torch.ops.curious_hip.*` (SURVEY 8b names a `TORCH_LIBRARY(curious_hip, ...)` layer).

NATIVE (round 6): the entry points whose arguments are tensors and scalars are registered from C++ --
`curious_amd/csrc/torch_library.cpp`, TORCH_LIBRARY_FRAGMENT(curious_hip, m) + TORCH_LIBRARY_IMPL(curious_hip, CUDA, m), built
into `curious_amd/lib/libcurious_torch.so` and loaded here: TORCH_CHECKed arguments, the current stream of the calling
context, schema aliasing (`Tensor(a!)`), and a call into the same `libcurious_hip.so` the ctypes binding loads:

    torch.ops.curious_hip.polyak_update(target, main, 0.95)
    torch.ops.curious_hip.adam_update(theta, m, v, grad, n_Q, n_pi, alpha_Q, alpha_pi)
    torch.ops.curious_hip.param_checksum(theta)                       -> int64[2]
    torch.ops.curious_hip.norm_update(rows, col_off, dim, acc)
    torch.ops.curious_hip.norm_recompute(acc, state, world_size, eps)
    torch.ops.curious_hip.policy_forward(cfg_i, cfg_f, theta, o, g, td, clip_obs, compute_Q) -> (pi, Q)
    torch.ops.curious_hip.ddpg_grads(cfg_i, cfg_f, theta, theta_target, batch, batch_layout, grad) -> (losses, Q_pi)

    cfg_i = [dimo, dimg, dimu, dimtd, hidden, layers, modular, clip_pos_returns, normalize_obs],
    cfg_f = [max_u, gamma, clip_return, action_l2, norm_clip]                                  (curious_net_cfg_t)
    batch_layout = [off_o, off_td, off_u, off_g, off_o2, off_g2, off_r, off_ag, off_ag2, off_extra, stride]

The three HOT entry points take the reference's table-shaped arguments (record / batch layouts, task tables, sampler
descriptions) -- the dispatcher's schema language has no struct type.  They get an opaque DESCRIPTOR instead (round 4):
`desc_create(...)` files the structs once and returns an int64 handle, the ops (registered here, in the same namespace, with
`torch.library.custom_op`) then take the handle + tensors:

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
stay on the ctypes binding (curious_amd/ops.py).  Every face calls the same symbols; there is no second implementation.
"""
import os

import torch

from curious_amd import _lib, ops
from curious_amd._lib import lib

_NS = 'curious_hip'


def _load_native():
    """libcurious_torch.so (csrc/torch_library.cpp: TORCH_LIBRARY_FRAGMENT(curious_hip, ...) + the CUDA-key implementations in
    C++ over the C ABI), built by curious_amd.build.build_torch_library.  No fallback: a missing or stale library raises."""
    from curious_amd import build
    lib()                                                    # the C ABI first (the same file the native ops link to)
    path = build.TORCH_LIB
    stamp = os.path.join(build.LIBDIR, 'build_torch.sha256')
    if not os.path.exists(path) or not os.path.exists(stamp) or open(stamp).read().strip() != build._torch_digest():
        raise _lib.CuriousHipError('libcurious_torch.so is missing or was built from other sources: run '
                                   '`python -m curious_amd.build` (torch.ops.curious_hip.* have no Python fallback)')
    torch.ops.load_library(path)


_load_native()


@torch.library.register_fake(_NS + '::param_checksum')
def _(theta):
    return theta.new_empty(2, dtype=torch.int64)


@torch.library.register_fake(_NS + '::policy_forward')
def _(cfg_i, cfg_f, theta, o, g, td, clip_obs, compute_Q):
    return o.new_empty([o.shape[0], int(cfg_i[2])]), o.new_empty([o.shape[0], 1])


@torch.library.register_fake(_NS + '::ddpg_grads')
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
