"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/curious_hip.h).

Tensors must live on the GPU (`cuda` device of PyTorch-ROCm); PyTorch is only the allocator / stream
provider here.  All calls enqueue on torch's current stream and never synchronise.
"""
import ctypes as C

import numpy as np
import torch

from curious_amd import _lib
from curious_amd._lib import check, current_stream, lib, ptr


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.CuriousHipError('%s must be a GPU tensor (curious_amd has no CPU path)' % name)
    return t


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int(0)
    check(lib().curious_device_info(name, 256, C.byref(cu)), 'curious_device_info')
    return name.value.decode(), cu.value


def her_sample(storage, buf_stride, layout, tasks, params, n, batch, plan=None, rng=None):
    """storage: float32 GPU tensor holding [nbuf][capacity][T+1][row_stride]; batch: [n, batch_stride]."""
    _dev(storage, 'storage')
    _dev(batch, 'batch')
    L = layout.c_layout()
    BL = layout.c_batch_layout()
    assert batch.stride(0) == BL.stride and batch.shape[0] >= n
    check(lib().curious_her_sample(ptr(storage), int(buf_stride), C.byref(L), C.byref(tasks), C.byref(params),
                                   C.byref(plan) if plan is not None else None,
                                   C.byref(rng) if rng is not None else None,
                                   int(n), ptr(batch), C.byref(BL), current_stream()), 'curious_her_sample')


def make_plan(ep, t, u_her, u_off, buf=None, task_to_replay=None, out_row=None):
    """Device arrays -> SamplePlan (keeps references alive on the returned object)."""
    p = _lib.SamplePlan()
    p._keep = (ep, t, u_her, u_off, buf, task_to_replay, out_row)
    assert ep.dtype == torch.int32 and t.dtype == torch.int32
    assert u_her.dtype == torch.float64 and u_off.dtype == torch.float64
    p.ep, p.t, p.u_her, p.u_off = ep.data_ptr(), t.data_ptr(), u_her.data_ptr(), u_off.data_ptr()
    p.buf = buf.data_ptr() if buf is not None else None
    p.task_to_replay = task_to_replay.data_ptr() if task_to_replay is not None else None
    p.out_row = out_row.data_ptr() if out_row is not None else None
    return p


def store_episodes(storage, staging, layout, pair_src, pair_dst):
    _dev(storage, 'storage')
    _dev(staging, 'staging')
    assert pair_src.dtype == torch.int32 and pair_dst.dtype == torch.int64
    L = layout.c_layout()
    check(lib().curious_store_episodes(ptr(storage), ptr(staging), C.byref(L), ptr(pair_src), ptr(pair_dst),
                                       int(pair_src.numel()), current_stream()), 'curious_store_episodes')


def route_store_episodes(storage, staging, layout, active, ntasks, n_route, n_episodes, cur_size, buf_alias, capacity,
                         seed, call, skip, pair_src, pair_dst, n_pairs, n_ranks=1, tab_stride=0, seed_stride=0,
                         tasks=None):
    """Device-side routing + copy of a batch of episodes (curious_route_store_episodes).  n_ranks > 1: n_episodes per
    virtual rank, every rank into its own buffers (curious_route_store_episodes_ranks).  tasks: the routing launch
    evaluates the activity flags itself and writes them to `active` (curious_activity_route_store_episodes)."""
    L = layout.c_layout()
    if tasks is not None:
        check(lib().curious_activity_route_store_episodes(
            ptr(_dev(storage, 'storage')), ptr(_dev(staging, 'staging')), C.byref(L), C.byref(tasks),
            int(layout.off['change']), ptr(active), int(n_route), int(n_episodes), int(n_ranks), ptr(cur_size),
            ptr(buf_alias), int(tab_stride), int(capacity), int(seed) & 0xFFFFFFFFFFFFFFFF,
            int(seed_stride) & 0xFFFFFFFFFFFFFFFF, int(call), ptr(skip), ptr(pair_src), ptr(pair_dst), ptr(n_pairs),
            current_stream()), 'curious_activity_route_store_episodes')
        return
    if n_ranks > 1:
        check(lib().curious_route_store_episodes_ranks(
            ptr(_dev(storage, 'storage')), ptr(_dev(staging, 'staging')), C.byref(L), ptr(active), int(ntasks),
            int(n_route), int(n_episodes), int(n_ranks), ptr(cur_size), ptr(buf_alias), int(tab_stride), int(capacity),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(seed_stride) & 0xFFFFFFFFFFFFFFFF, int(call), ptr(skip), ptr(pair_src),
            ptr(pair_dst), ptr(n_pairs), current_stream()), 'curious_route_store_episodes_ranks')
        return
    check(lib().curious_route_store_episodes(ptr(_dev(storage, 'storage')), ptr(_dev(staging, 'staging')), C.byref(L),
                                             ptr(active), int(ntasks), int(n_route), int(n_episodes), ptr(cur_size),
                                             ptr(buf_alias), int(capacity), int(seed) & 0xFFFFFFFFFFFFFFFF, int(call),
                                             ptr(skip), ptr(pair_src), ptr(pair_dst), ptr(n_pairs), current_stream()),
          'curious_route_store_episodes')


def store_slots_host(seed, call, task, size, episodes):
    """The random slots curious_route_store_episodes draws for `episodes` of `task` once the buffer is full (host)."""
    ep = np.ascontiguousarray(episodes, dtype=np.int32)
    out = np.empty(ep.size, np.int64)
    check(lib().curious_store_slots_host(int(seed) & 0xFFFFFFFFFFFFFFFF, int(call), int(task), int(size), int(ep.size),
                                         C.c_void_p(ep.ctypes.data), C.c_void_p(out.ctypes.data)),
          'curious_store_slots_host')
    return out


def episode_activity(staging, layout, tasks, n_episodes, active):
    L = layout.c_layout()
    check(lib().curious_episode_activity(ptr(_dev(staging, 'staging')), C.byref(L), C.byref(tasks),
                                         int(layout.off['change']), int(n_episodes), ptr(active),
                                         current_stream()), 'curious_episode_activity')


def norm_scratch_doubles(n_rows, dim):
    return int(lib().curious_norm_scratch_doubles(int(n_rows), int(dim)))


def norm_update(rows, n_rows, stride, col_off, dim, acc, scratch):
    check(lib().curious_norm_update(ptr(_dev(rows, 'rows')), int(n_rows), int(stride), int(col_off), int(dim),
                                    ptr(acc), ptr(scratch), current_stream()), 'curious_norm_update')


def norm_update_pair(rows, n_rows, stride, off_a, dim_a, off_b, dim_b, acc_a, acc_b, state_a, state_b, eps_a, eps_b,
                     scratch, skip=None):
    """skip: device float; non-zero (the NaN word of a rollout) = accumulate nothing."""
    check(lib().curious_norm_update_pair(ptr(_dev(rows, 'rows')), int(n_rows), int(stride), int(off_a), int(dim_a),
                                         int(off_b), int(dim_b), ptr(acc_a), ptr(acc_b), ptr(state_a), ptr(state_b),
                                         float(eps_a), float(eps_b), ptr(scratch), ptr(skip), current_stream()),
          'curious_norm_update_pair')


def norm_pair_scratch_doubles(n_rows, dim_a, dim_b):
    return int(lib().curious_norm_pair_scratch_doubles(int(n_rows), int(dim_a), int(dim_b)))


def norm_recompute(acc, state, dim, world_size, eps):
    check(lib().curious_norm_recompute(ptr(_dev(acc, 'acc')), ptr(state), int(dim), float(world_size), float(eps),
                                       current_stream()), 'curious_norm_recompute')


def make_net_cfg(dimo, dimg, dimu, dimtd, hidden, layers, modular, max_u, gamma, clip_return, action_l2,
                 clip_pos_returns=True, normalize_obs=False, norm_clip=5.0, loss_rows=0):
    c = _lib.NetCfg()
    c.dimo, c.dimg, c.dimu, c.dimtd, c.hidden, c.layers = dimo, dimg, dimu, dimtd, hidden, layers
    c.modular = int(bool(modular))
    c.max_u, c.gamma, c.action_l2 = float(max_u), float(gamma), float(action_l2)
    c.clip_return = float(min(clip_return, 3.0e38))
    c.clip_pos_returns, c.normalize_obs, c.norm_clip = int(bool(clip_pos_returns)), int(bool(normalize_obs)), \
        float(min(norm_clip, 3.0e38))
    c.loss_rows = int(loss_rows)             # > 0: the batch holds the minibatches of B / loss_rows virtual ranks
    return c


def param_counts(cfg):
    return int(lib().curious_param_count_Q(C.byref(cfg))), int(lib().curious_param_count_pi(C.byref(cfg)))


def param_layout(cfg):
    """(P_Q, P_pi, offset of theta_pi, total floats) of the padded parameter vector."""
    return (int(lib().curious_param_count_Q(C.byref(cfg))), int(lib().curious_param_count_pi(C.byref(cfg))),
            int(lib().curious_param_offset_pi(C.byref(cfg))), int(lib().curious_param_total(C.byref(cfg))))


def pad_params(cfg, flat):
    """Reference flat vector [theta_Q | theta_pi] (NumPy) -> padded layout (NumPy float32)."""
    PQ, Ppi, off, total = param_layout(cfg)
    flat = np.asarray(flat, dtype=np.float32)
    assert flat.shape[0] == PQ + Ppi
    out = np.zeros(total, np.float32)
    out[:PQ] = flat[:PQ]
    out[off:off + Ppi] = flat[PQ:]
    return out


def unpad_params(cfg, padded):
    PQ, Ppi, off, total = param_layout(cfg)
    padded = np.asarray(padded)
    return np.concatenate([padded[:PQ], padded[off:off + Ppi]])


def workspace_floats(cfg, B):
    return int(lib().curious_workspace_floats(C.byref(cfg), int(B)))


def _next_batch(layout, next_batch, storage, buf_stride, tasks, params, rng):
    """curious_next_batch_t (None without a next batch); the ctypes objects it points to stay referenced on it."""
    if next_batch is None:
        return None
    L = layout.c_layout()
    N = _lib.NextBatch()
    N.storage, N.buf_stride = ptr(_dev(storage, 'storage')), int(buf_stride)
    N.L, N.tasks, N.P, N.rng = C.pointer(L), C.pointer(tasks), C.pointer(params), C.pointer(rng)
    N.batch = ptr(_dev(next_batch, 'next_batch'))
    N._keep = (L, tasks, params, rng)
    return N


def ddpg_grads(cfg, theta_main, theta_target, batch, layout, B, workspace, grad, out_losses, out_Q_pi,
               o_stats=None, g_stats=None, step_ctr=None, params_unchanged=False, next_batch=None, storage=None,
               buf_stride=0, tasks=None, params=None, rng=None):
    """params_unchanged: since the previous call on this workspace only an optimiser call that was given
    ddpg_transposed(cfg, B, workspace) has written theta_main (the transposed weight copies in the workspace are current).
    next_batch (+ storage, buf_stride, tasks, params, rng): the device-drawn gather of the next update's batch as part of
    the call (multi-rank path: in spare workgroups of the row-local launch)."""
    BL = layout.c_batch_layout()
    N = _next_batch(layout, next_batch, storage, buf_stride, tasks, params, rng)
    check(lib().curious_ddpg_grads(C.byref(cfg), ptr(_dev(theta_main, 'theta_main')), ptr(theta_target),
                                   ptr(_dev(batch, 'batch')), C.byref(BL), int(B), ptr(o_stats), ptr(g_stats),
                                   ptr(workspace), ptr(grad), ptr(out_losses), ptr(out_Q_pi), ptr(step_ctr),
                                   int(bool(params_unchanged)), C.byref(N) if N is not None else None,
                                   current_stream()), 'curious_ddpg_grads')


def ddpg_transposed(cfg, B, workspace):
    """curious_transposed_t of the weight copies the gradient pass keeps in `workspace` (n == 0: none for this shape)."""
    T = _lib.Transposed()
    check(lib().curious_ddpg_transposed(C.byref(cfg), int(B), ptr(_dev(workspace, 'workspace')), C.byref(T)),
          'curious_ddpg_transposed')
    return T


def ddpg_update(cfg, theta_main, theta_target, batch, layout, B, workspace, grad, out_losses, out_Q_pi, m, v,
                step_ctr=None, alpha_tab=None, tab_base=0, alpha_Q=0.0, alpha_pi=0.0, beta1=0.9, beta2=0.999,
                epsilon=1e-08, o_stats=None, g_stats=None, next_batch=None, storage=None, buf_stride=0, tasks=None,
                params=None, rng=None, params_unchanged=False):
    """One whole single-rank update (curious_ddpg_update): gradients with Adam applied in the weight-gradient launch;
    with `next_batch` (a staging tensor other than `batch`) the device-drawn HER gather of the next update rides along.
    params_unchanged: nothing but the previous call of this op on this workspace has written theta_main since (the
    library then skips rebuilding the transposed copies it keeps in the workspace)."""
    BL = layout.c_batch_layout()
    A = _adam_state(m, v, alpha_tab, tab_base, alpha_Q, alpha_pi, beta1, beta2, epsilon, params_unchanged)
    nb = None
    if next_batch is not None:
        L = layout.c_layout()
        N = _lib.NextBatch()
        N.storage, N.buf_stride = ptr(_dev(storage, 'storage')), int(buf_stride)
        N.L, N.tasks, N.P, N.rng = C.pointer(L), C.pointer(tasks), C.pointer(params), C.pointer(rng)
        N.batch = ptr(_dev(next_batch, 'next_batch'))
        nb = C.byref(N)
    check(lib().curious_ddpg_update(C.byref(cfg), ptr(_dev(theta_main, 'theta_main')), ptr(theta_target),
                                    ptr(_dev(batch, 'batch')), C.byref(BL), int(B), ptr(o_stats), ptr(g_stats),
                                    ptr(workspace), ptr(grad), ptr(out_losses), ptr(out_Q_pi), ptr(step_ctr),
                                    C.byref(A), nb, current_stream()), 'curious_ddpg_update')


def _adam_state(m, v, alpha_tab, tab_base, alpha_Q, alpha_pi, beta1, beta2, epsilon, params_unchanged=False):
    f = np.float32
    A = _lib.AdamState()
    A.m, A.v = ptr(_dev(m, 'm')), ptr(_dev(v, 'v'))
    A.alpha_tab = ptr(alpha_tab)
    A.tab_base, A.tab_len = int(tab_base), int(alpha_tab.shape[0]) if alpha_tab is not None else 0
    A.alpha_Q, A.alpha_pi = float(alpha_Q), float(alpha_pi)
    A.beta1, A.one_minus_beta1 = float(f(beta1)), float(f(1 - beta1))
    A.beta2, A.one_minus_beta2 = float(f(beta2)), float(f(1 - beta2))
    A.epsilon = float(f(epsilon))
    A.params_unchanged = int(bool(params_unchanged))
    return A


def ddpg_grads_experts(cfg, n_experts, expert_stride, grad_stride, theta_main, theta_target, batch, layout, B, workspace,
                       grad, out_losses, out_Q_pi, step_ctr, params_unchanged=False, seed_stride=0, next_batch=None,
                       storage=None, buf_stride=0, tasks=None, params=None, rng=None, o_stats=None, g_stats=None):
    """curious_ddpg_grads for n_experts agents in one launch sequence (the first half of a data-parallel batched
    update); tensors as in ddpg_update_experts.  next_batch: every expert's next batch is gathered in the same launch."""
    BL = layout.c_batch_layout()
    N = _next_batch(layout, next_batch, storage, buf_stride, tasks, params, rng)
    check(lib().curious_ddpg_grads_experts(C.byref(cfg), int(n_experts), int(expert_stride), int(grad_stride),
                                           ptr(_dev(theta_main, 'theta_main')), ptr(theta_target),
                                           ptr(_dev(batch, 'batch')), C.byref(BL), int(B), ptr(o_stats), ptr(g_stats),
                                           ptr(workspace), ptr(grad),
                                           ptr(out_losses), ptr(out_Q_pi), ptr(step_ctr), int(bool(params_unchanged)),
                                           int(seed_stride) & 0xFFFFFFFFFFFFFFFF, C.byref(N) if N is not None else None,
                                           current_stream()), 'curious_ddpg_grads_experts')


def adam_update_and_sample_experts(n_experts, expert_stride, grad_stride, seed_stride, theta, m, v, grad, n_Q, n_pi,
                                   alpha_tab, step_ctr, tab_base, storage, buf_stride, layout, tasks, params, rng, n,
                                   batch, beta1=0.9, beta2=0.999, epsilon=1e-08, keep=None):
    """adam_update_and_sample for n_experts agents in one launch (the second half of a data-parallel batched update)."""
    f = np.float32
    L = layout.c_layout()
    BL = layout.c_batch_layout()
    check(lib().curious_adam_update_and_sample_experts(
        int(n_experts), int(expert_stride), int(grad_stride), int(seed_stride) & 0xFFFFFFFFFFFFFFFF,
        ptr(_dev(theta, 'theta')), ptr(m), ptr(v), ptr(grad), int(n_Q), int(n_pi), ptr(alpha_tab), ptr(step_ctr),
        int(tab_base), int(alpha_tab.shape[0]), float(f(beta1)), float(f(1 - beta1)), float(f(beta2)),
        float(f(1 - beta2)), float(f(epsilon)), ptr(storage), int(buf_stride), C.byref(L),
        C.byref(tasks), C.byref(params), C.byref(rng), int(n), ptr(batch), C.byref(BL),
        C.byref(keep) if keep is not None else None, current_stream()), 'curious_adam_update_and_sample_experts')


def ddpg_update_experts(cfg, n_experts, expert_stride, grad_stride, seed_stride, theta_main, theta_target, batch, layout,
                        B, workspace, grad, out_losses, out_Q_pi, m, v, step_ctr, alpha_tab, tab_base, next_batch,
                        storage, buf_stride, tasks, params, rng, beta1=0.9, beta2=0.999, epsilon=1e-08,
                        params_unchanged=False, o_stats=None, g_stats=None):
    """One update of n_experts agents in one launch sequence (curious_ddpg_update_experts).  Every tensor is expert
    0's view of a slab [n_experts, expert_stride] (the gradients: of a block [n_experts, grad_stride]); `rng` is expert
    0's sampler description."""
    BL = layout.c_batch_layout()
    A = _adam_state(m, v, alpha_tab, tab_base, 0.0, 0.0, beta1, beta2, epsilon, params_unchanged)
    L = layout.c_layout()
    N = _lib.NextBatch()
    N.storage, N.buf_stride = ptr(_dev(storage, 'storage')), int(buf_stride)
    N.L, N.tasks, N.P, N.rng = C.pointer(L), C.pointer(tasks), C.pointer(params), C.pointer(rng)
    N.batch = ptr(_dev(next_batch, 'next_batch'))
    check(lib().curious_ddpg_update_experts(C.byref(cfg), int(n_experts), int(expert_stride), int(grad_stride),
                                            int(seed_stride) & 0xFFFFFFFFFFFFFFFF,
                                            ptr(_dev(theta_main, 'theta_main')), ptr(theta_target),
                                            ptr(_dev(batch, 'batch')), C.byref(BL), int(B), ptr(o_stats), ptr(g_stats),
                                            ptr(workspace), ptr(grad),
                                            ptr(out_losses), ptr(out_Q_pi), ptr(step_ctr), C.byref(A), C.byref(N),
                                            current_stream()), 'curious_ddpg_update_experts')


def policy_forward(cfg, theta, o, g, td, n, clip_obs, workspace, out_pi, out_Q=None, ag=None,
                   relative_goals=False, o_stats=None, g_stats=None):
    _dev(o, 'o')
    check(lib().curious_policy_forward(C.byref(cfg), ptr(_dev(theta, 'theta')), ptr(o), int(o.stride(0)),
                                       ptr(ag), int(ag.stride(0)) if ag is not None else 0, ptr(g),
                                       int(g.stride(0)), ptr(td), int(td.stride(0)) if td is not None else 0,
                                       int(n), float(clip_obs), int(bool(relative_goals)), ptr(o_stats),
                                       ptr(g_stats), ptr(workspace), ptr(out_pi), ptr(out_Q), current_stream()),
          'curious_policy_forward')


def action_noise(u, n, dimu, noise_scale, random_eps, max_u, randn=None, binom=None, unif=None, seed=0, counter=0):
    check(lib().curious_action_noise(ptr(_dev(u, 'u')), int(u.stride(0)), int(n), int(dimu), float(noise_scale),
                                     float(random_eps), float(max_u), ptr(randn), ptr(binom), ptr(unif),
                                     int(seed), int(counter), current_stream()), 'curious_action_noise')


def adam_alpha(stepsize, t, beta1=0.9, beta2=0.999):
    """mpi_adam.py:30 in float64, rounded to float32 the way NumPy 1.x multiplies it into float32 arrays."""
    return np.float32(stepsize * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t))


def adam_alpha_table(stepsize, ts, beta1=0.9, beta2=0.999):
    """adam_alpha for a run of step numbers, element for element the same float32 values: the powers come from Python's
    float ** int (one libm call each, as in the scalar form -- NumPy's vectorised power may round differently), the
    rest are IEEE-exact element-wise operations in the scalar form's order."""
    tl = [int(t) for t in np.asarray(ts).tolist()]
    b1, b2 = float(beta1), float(beta2)
    p1 = np.array([b1 ** t for t in tl], np.float64)
    p2 = np.array([b2 ** t for t in tl], np.float64)
    return (stepsize * np.sqrt(1 - p2) / (1 - p1)).astype(np.float32)


def adam_update(theta, m, v, grad, n_Q, n_pi, alpha_Q=None, alpha_pi=None, beta1=0.9, beta2=0.999, epsilon=1e-08,
                alpha_tab=None, step_ctr=None, tab_base=0, keep=None):
    f = np.float32
    ah = None
    if alpha_tab is None:
        ah = (C.c_float * 2)(float(alpha_Q), float(alpha_pi if alpha_pi is not None else 0.0))
    check(lib().curious_adam_update(ptr(_dev(theta, 'theta')), ptr(m), ptr(v), ptr(grad), int(n_Q), int(n_pi),
                                    ptr(alpha_tab), ptr(step_ctr), int(tab_base),
                                    int(alpha_tab.shape[0]) if alpha_tab is not None else 0, ah,
                                    float(f(beta1)), float(f(1 - beta1)), float(f(beta2)), float(f(1 - beta2)),
                                    float(f(epsilon)), C.byref(keep) if keep is not None else None, current_stream()),
          'curious_adam_update')


def adam_update_and_sample(theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, storage, buf_stride, layout,
                           tasks, params, rng, n, batch, beta1=0.9, beta2=0.999, epsilon=1e-08, keep=None):
    """Fused Adam (table-driven step sizes) + device-drawn HER gather of the next update.  keep: ddpg_transposed(...) of
    the workspace whose transposed weight copies this call keeps current."""
    f = np.float32
    L = layout.c_layout()
    BL = layout.c_batch_layout()
    check(lib().curious_adam_update_and_sample(
        ptr(_dev(theta, 'theta')), ptr(m), ptr(v), ptr(grad), int(n_Q), int(n_pi), ptr(alpha_tab), ptr(step_ctr),
        int(tab_base), int(alpha_tab.shape[0]), None, float(f(beta1)), float(f(1 - beta1)), float(f(beta2)),
        float(f(1 - beta2)), float(f(epsilon)), ptr(_dev(storage, 'storage')), int(buf_stride), C.byref(L),
        C.byref(tasks), C.byref(params), C.byref(rng), int(n), ptr(batch), C.byref(BL),
        C.byref(keep) if keep is not None else None, current_stream()), 'curious_adam_update_and_sample')


def polyak_update(target, main, polyak):
    f = np.float32
    check(lib().curious_polyak_update(ptr(_dev(target, 'target')), ptr(main), int(target.numel()),
                                      float(f(polyak)), float(f(1. - polyak)), current_stream()),
          'curious_polyak_update')


def param_checksum(theta, out):
    check(lib().curious_param_checksum(ptr(_dev(theta, 'theta')), int(theta.numel()), ptr(out), current_stream()),
          'curious_param_checksum')


def make_env_cfg(ntasks, dimo, T, seed, wrap=0):
    e = _lib.EnvCfg()
    e.ntasks, e.dimo, e.T, e.seed, e.wrap = int(ntasks), int(dimo), int(T), int(seed) & 0xFFFFFFFFFFFFFFFF, int(wrap)
    return e


def env_reset(ecfg, layout, env_id0, episode, tasks, goals_raw, n, o, ag, g, td, staging, flags=None, counter=None,
              delta=0):
    """counter / delta: *counter += delta in the same launch (curious_env_reset_count)."""
    L = layout.c_layout()
    check(lib().curious_env_reset_count(C.byref(ecfg), C.byref(L), int(env_id0), ptr(episode), ptr(tasks),
                                        ptr(goals_raw), int(n), ptr(o), ptr(ag), ptr(g), ptr(td), ptr(staging),
                                        ptr(flags), ptr(counter), int(delta), current_stream()), 'curious_env_reset')


def counter_add(p, delta):
    check(lib().curious_counter_add(ptr(_dev(p, 'p')), int(delta), current_stream()), 'curious_counter_add')


def policy_act_env_step(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, u_out, ecfg, layout,
                        env_id0, episode, tasks, t, o, ag, g, td, staging, reward_eps, counter_base=None, flags=None,
                        o_stats=None, g_stats=None, relative_goals=False):
    """o_stats / g_stats: the normalisers' state vectors for networks with input normalisation; relative_goals: the policy
    sees g - ag (both through the *_stats entry)."""
    L = layout.c_layout()
    args = (C.byref(cfg), ptr(_dev(theta, 'theta')), int(n), float(clip_obs), ptr(workspace), float(noise_scale),
            float(random_eps), int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), ptr(counter_base), ptr(u_out),
            int(u_out.stride(0)), C.byref(ecfg), C.byref(L), int(env_id0), ptr(episode), ptr(tasks), int(t), ptr(o),
            ptr(ag), ptr(g), ptr(td), ptr(staging), int(layout.off['change']), int(layout.off['info_is_success']),
            float(reward_eps), ptr(flags))
    if o_stats is not None or g_stats is not None or relative_goals:
        check(lib().curious_policy_act_env_step_stats(*args, int(bool(relative_goals)), ptr(o_stats), ptr(g_stats),
                                                      current_stream()), 'curious_policy_act_env_step_stats')
    else:
        check(lib().curious_policy_act_env_step(*args, current_stream()), 'curious_policy_act_env_step')


def rank_groups(group, seed_stride, exploit=None):
    """curious_rank_groups_t: the envs of a batched rollout as consecutive groups of `group` envs, one per virtual rank;
    exploit: device int32 [number of groups], non-zero = that group acts without exploration noise."""
    g = _lib.RankGroups()
    g.group, g.seed_stride = int(group), int(seed_stride) & 0xFFFFFFFFFFFFFFFF
    g.exploit = exploit.data_ptr() if exploit is not None else None
    g._keep = exploit
    return g


def policy_rollout(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, u_out, ecfg, layout,
                   env_id0, episode, tasks, t0, nsteps, o, ag, g, td, staging, reward_eps, counter_base=None, flags=None,
                   o_stats=None, g_stats=None, relative_goals=False, groups=None):
    """nsteps x policy_act_env_step (steps t0 .. t0 + nsteps - 1, noise counters counter, counter + 1, ...); one launch
    on the row-local route.  groups: rank_groups(...) -- the envs of several virtual ranks in one launch."""
    L = layout.c_layout()
    args = (C.byref(cfg), ptr(_dev(theta, 'theta')), int(n), float(clip_obs), ptr(workspace), float(noise_scale),
            float(random_eps), int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter) & 0xFFFFFFFFFFFFFFFF, ptr(counter_base),
            ptr(u_out),
            int(u_out.stride(0)), C.byref(ecfg), C.byref(L), int(env_id0), ptr(episode), ptr(tasks), int(t0), int(nsteps),
            ptr(o), ptr(ag), ptr(g), ptr(td), ptr(staging), int(layout.off['change']),
            int(layout.off['info_is_success']), float(reward_eps), ptr(flags))
    if groups is not None:
        check(lib().curious_policy_rollout_ranks(*args, int(bool(relative_goals)), ptr(o_stats), ptr(g_stats),
                                                 C.byref(groups), current_stream()), 'curious_policy_rollout_ranks')
    elif o_stats is not None or g_stats is not None or relative_goals:
        check(lib().curious_policy_rollout_stats(*args, int(bool(relative_goals)), ptr(o_stats), ptr(g_stats),
                                                 current_stream()), 'curious_policy_rollout_stats')
    else:
        check(lib().curious_policy_rollout(*args, current_stream()), 'curious_policy_rollout')


def env_step(ecfg, layout, env_id0, episode, tasks, u, t, n, o, ag, g, td, staging, reward_eps, flags=None):
    L = layout.c_layout()
    check(lib().curious_env_step(C.byref(ecfg), C.byref(L), int(env_id0), ptr(episode), ptr(tasks), ptr(u),
                                 int(u.stride(0)), int(t), int(n), ptr(o), ptr(ag), ptr(g), ptr(td), ptr(staging),
                                 int(layout.off['change']), int(layout.off['info_is_success']), float(reward_eps),
                                 ptr(flags), current_stream()), 'curious_env_step')


def prof_enable(on):
    check(lib().curious_prof_enable(int(bool(on))), 'curious_prof_enable')


def prof_collect():
    """{kernel name: (launch count, total ms)} since the last collect (synchronises the device)."""
    n = lib().curious_prof_kernel_count()
    counts = (C.c_int64 * n)()
    ms = (C.c_double * n)()
    check(lib().curious_prof_collect(counts, ms), 'curious_prof_collect')
    return {lib().curious_prof_kernel_name(k).decode(): (int(counts[k]), float(ms[k])) for k in range(n)}


def prof_launch_counts():
    """{kernel name: launches since the library was loaded} (counted with or without event timing; no sync)."""
    n = lib().curious_prof_kernel_count()
    counts = (C.c_int64 * n)()
    check(lib().curious_prof_launch_counts(counts), 'curious_prof_launch_counts')
    return {lib().curious_prof_kernel_name(k).decode(): int(counts[k]) for k in range(n)}


def set_option(name, value):
    """Run-time option of the library (include/curious_hip.h: rows, rows_xcd, xcd_map, fault_inject, qt_spins)."""
    check(lib().curious_set_option(name.encode(), int(value)), 'curious_set_option')


def get_option(name):
    v = int(lib().curious_get_option(name.encode()))
    if v < 0:
        check(-1, 'curious_get_option')
    return v


class option:
    """`with ops.option('rows', 0): ...` -- set an option for a block (tests: the tiled route, fault injection)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.old)


def fault_word(cfg, B, workspace, n=1):
    """int32 view (1 element) of the fault word inside a gradient workspace (curious_workspace_fault_offset).  n = 64:
    the whole (zeroed) block it sits in."""
    off = int(lib().curious_workspace_fault_offset(C.byref(cfg), int(B)))
    if off < 0:
        raise _lib.CuriousHipError('curious_workspace_fault_offset: bad arguments')
    return _dev(workspace, 'workspace')[off:off + n].view(torch.int32)


def dw_stamps(cfg, B, workspace, n_blocks):
    """Lab (option 'lab_dw_stamps'): the [n_blocks, 8] int64 cycle stamps the weight-gradient / optimiser launch left in
    the workspace (curious_workspace_stamps_offset)."""
    off = int(lib().curious_workspace_stamps_offset(C.byref(cfg), int(B)))
    if off < 0:
        raise _lib.CuriousHipError('curious_workspace_stamps_offset: bad arguments')
    return workspace[off:off + 16 * n_blocks].view(torch.int64).view(n_blocks, 8)


def allreduce_adam_ipc(peers, theta, m, v, n_Q, n_pi, alpha_tab, step_ctr, tab_base, epoch, done, err, keep, spins=0,
                       beta1=0.9, beta2=0.999, epsilon=1e-08):
    """curious_allreduce_adam_ipc: reduce-scatter over the peers' gradient vectors + Adam on the owned slice + all-gather
    of the new slices through the peers' staging vectors + the copy into the local `theta` + the transposed copies, one
    kernel (csrc/ipc.hip).  peers: _lib.IpcPeers of mapped pointers; epoch / done / err: local device words."""
    f = np.float32
    check(lib().curious_allreduce_adam_ipc(C.byref(peers), ptr(_dev(theta, 'theta')), ptr(_dev(m, 'm')), ptr(v), int(n_Q),
                                           int(n_pi), ptr(alpha_tab), ptr(step_ctr), int(tab_base),
                                           int(alpha_tab.shape[0]), float(f(beta1)), float(f(1 - beta1)), float(f(beta2)),
                                           float(f(1 - beta2)), float(f(epsilon)), ptr(epoch), ptr(done), ptr(err),
                                           int(spins), C.byref(keep) if keep is not None else None, current_stream()),
          'curious_allreduce_adam_ipc')
