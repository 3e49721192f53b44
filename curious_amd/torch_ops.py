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
`desc_create(...)` files the structs once (a `curious_torch_desc_t` of pointers to them, registered with the C++ library) and
returns an int64 handle; the ops -- native as well -- then take the handle + tensors:

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
import ctypes as C
import os

import torch

from curious_amd import _lib
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


# ------------------------------------------------------------------ descriptors of the three hot entry points
class TorchDesc(C.Structure):
    """curious_torch_desc_t of csrc/torch_library.cpp, field by field: pointers to the C ABI's own structs + the scalars of
    the hot entry points.  Pointers an op does not need stay NULL."""
    _fields_ = [('cfg', C.c_void_p), ('L', C.c_void_p), ('BL', C.c_void_p), ('tasks', C.c_void_p), ('P', C.c_void_p),
                ('rng', C.c_void_p), ('plan', C.c_void_p), ('E', C.c_void_p), ('o_stats', C.c_void_p),
                ('g_stats', C.c_void_p), ('buf_stride', C.c_int64), ('tab_base', C.c_int64), ('seed', C.c_uint64),
                ('counter', C.c_uint64), ('noise_scale', C.c_double), ('random_eps', C.c_double),
                ('reward_eps', C.c_double), ('clip_obs', C.c_float), ('n', C.c_int32), ('B', C.c_int32),
                ('env_id0', C.c_int32), ('t0', C.c_int32), ('nsteps', C.c_int32), ('off_change', C.c_int32),
                ('off_success', C.c_int32), ('relative_goals', C.c_int32)]


_DESCS = {}


def desc_create(layout=None, tasks=None, params=None, rng=None, plan=None, cfg=None, ecfg=None, o_stats=None, g_stats=None,
                buf_stride=0, tab_base=0, n=0, B=0, clip_obs=0.0, noise_scale=0.0, random_eps=0.0, seed=0, counter=0,
                env_id0=0, t0=0, nsteps=0, reward_eps=0.0, relative_goals=False, keep=None):
    """File the struct-shaped arguments of a hot entry point (layouts, task tables, sampler / env descriptions, scalars)
    ONCE and return an int64 handle for torch.ops.curious_hip.{her_sample, ddpg_update, policy_rollout}.  The descriptor
    keeps what it is given alive (e.g. the device tables a SampleRng points into, through `keep=`) until desc_free."""
    d = TorchDesc()
    held = [layout, tasks, params, rng, plan, cfg, ecfg, o_stats, g_stats, keep]
    if layout is not None:
        L = layout.c_layout()
        d.L = C.addressof(L)
        held.append(L)
        if getattr(layout, 'batch_cols', None) is not None:
            BL = layout.c_batch_layout()
            d.BL = C.addressof(BL)
            held.append(BL)
        d.off_change = int(layout.off.get('change', 0))
        d.off_success = int(layout.off.get('info_is_success', 0))
    for name, obj in (('tasks', tasks), ('P', params), ('rng', rng), ('plan', plan), ('cfg', cfg), ('E', ecfg)):
        if obj is not None:
            setattr(d, name, C.addressof(obj))
    d.o_stats = o_stats.data_ptr() if o_stats is not None else None
    d.g_stats = g_stats.data_ptr() if g_stats is not None else None
    d.buf_stride, d.tab_base = int(buf_stride), int(tab_base)
    d.seed, d.counter = int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter) & 0xFFFFFFFFFFFFFFFF
    d.noise_scale, d.random_eps, d.reward_eps, d.clip_obs = float(noise_scale), float(random_eps), float(reward_eps), float(clip_obs)
    d.n, d.B, d.env_id0, d.t0, d.nsteps = int(n), int(B), int(env_id0), int(t0), int(nsteps)
    d.relative_goals = int(bool(relative_goals))
    h = int(torch.ops.curious_hip.desc_register(C.addressof(d)))
    _DESCS[h] = (d, held)
    return h


def desc_free(handle):
    if _DESCS.pop(int(handle), None) is not None:
        torch.ops.curious_hip.desc_release(int(handle))
