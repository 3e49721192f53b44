"""CPU-side checks of the C-ABI library: it builds, loads, and exports every declared symbol."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from curious_amd.build import build
    from curious_amd import _lib
    path = build(verbose=False)
    assert os.path.exists(path)
    L = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'curious_hip.h')).read()
    declared = set(re.findall(r'\b(curious_[a-z_A-Z0-9]+)\s*\(', header))
    assert declared, 'no declarations found'
    assert declared == set(_lib.PROTOTYPES.keys())
    for name in declared:
        assert hasattr(L, name), name
    assert L.curious_abi_version() == _lib.ABI_VERSION
    from curious_amd.build import source_digest
    assert L.curious_build_digest().decode() == source_digest()


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device, so this runs on CPU."""
    import ctypes as C
    from curious_amd import _lib
    L = _lib.lib()
    rc = L.curious_her_sample(None, 0, None, None, None, None, None, 4, None, None, None)
    assert rc != 0
    assert b'NULL' in L.curious_last_error()
    cfg = _lib.NetCfg()
    cfg.dimo, cfg.dimg, cfg.dimu, cfg.dimtd, cfg.hidden, cfg.layers, cfg.modular = 40, 12, 4, 4, 256, 3, 1
    assert L.curious_param_count_Q(C.byref(cfg)) == 147457
    assert L.curious_param_count_pi(C.byref(cfg)) == 147204
    # the fused update refuses a missing optimiser state and a next batch that aliases the current one
    rc = L.curious_ddpg_update(C.byref(cfg), None, None, None, None, 256, None, None, None, None, None, None, None,
                               None, None, None)
    assert rc != 0 and b'optimiser' in L.curious_last_error()
    st = _lib.AdamState()
    st.m, st.v = 64, 64                                       # never dereferenced: validation fails first
    nb = _lib.NextBatch()
    nb.batch = 4096
    rc = L.curious_ddpg_update(C.byref(cfg), None, None, C.c_void_p(4096), None, 256, None, None, None, None, None,
                               None, None, C.byref(st), C.byref(nb), None)
    assert rc != 0 and b'staging' in L.curious_last_error()
    cfg.dimo, cfg.dimg, cfg.dimtd = 52, 24, 8
    assert L.curious_param_count_pi(C.byref(cfg)) == (52 + 8) * 256 + 256 + 24 * 256 + 2 * (256 * 256 + 256) + 256 * 4 + 4


def test_product_refuses_cpu_tensors():
    import torch
    from curious_amd import ops, _lib
    with pytest.raises(_lib.CuriousHipError):
        ops.polyak_update(torch.zeros(4), torch.zeros(4), 0.95)


def test_host_descriptor_code_runs_up_to_the_launch_without_gpu():
    """On a machine without a GPU the entry points still run all of their host code -- layout arithmetic, workspace carving,
    problem descriptors, route selection -- and fail only at the kernel launch.  Run under tools/sanitize_cpu.sh this is
    the AddressSanitizer / UBSan coverage of the host side."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip('would launch kernels on fake pointers')
    from curious_amd import _lib
    L = _lib.lib()
    for dims in ((40, 12, 4, 3), (52, 24, 8, 3), (40, 12, 4, 2), (37, 12, 4, 3)):     # rows route, Arm8, 2 layers, generic
        cfg = _lib.NetCfg()
        cfg.dimo, cfg.dimg, cfg.dimtd, cfg.layers = dims
        cfg.dimu, cfg.hidden, cfg.modular = 4, 256, 1
        cfg.max_u, cfg.gamma, cfg.clip_return, cfg.action_l2, cfg.clip_pos_returns = 1.0, 0.98, 50.0, 1.0, 1
        ws = L.curious_workspace_floats(C.byref(cfg), 256)
        assert ws > 0 and L.curious_param_total(C.byref(cfg)) > L.curious_param_offset_pi(C.byref(cfg)) > 0
        BL = _lib.BatchLayout()
        BL.off_o, BL.off_td, BL.off_u, BL.off_g, BL.off_o2, BL.off_g2, BL.off_r = 0, 56, 64, 68, 92, 148, 172
        BL.off_ag, BL.off_ag2, BL.off_extra, BL.stride = 176, 200, 224, 256
        fake = [C.c_void_p(0x10000000 + 0x1000000 * i) for i in range(8)]          # never dereferenced on the host
        rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 256, None, None, fake[3], fake[4],
                                  fake[5], fake[6], None, 0, None, None)
        assert rc != 0 and b'launch failed' in L.curious_last_error()
        st = _lib.AdamState()
        st.m, st.v = fake[6].value, fake[7].value
        rc = L.curious_ddpg_update(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 256, None, None, fake[3], fake[4],
                                   fake[5], fake[6], None, C.byref(st), None, None)
        assert rc != 0 and b'launch failed' in L.curious_last_error()
        rc = L.curious_policy_forward(C.byref(cfg), fake[0], fake[1], cfg.dimo, None, 0, fake[2], cfg.dimg, fake[3],
                                      cfg.dimtd, 64, 200.0, 0, None, None, fake[4], fake[5], fake[6], None)
        assert rc != 0 and b'launch failed' in L.curious_last_error()


def test_store_slots_host_is_the_documented_philox_draw():
    """curious_store_slots_host (no GPU involved): slot = (Philox4x32-10(ctr = (episode, task, call, 31), key = seed).x
    * size) >> 32 -- checked against the oracle's NumPy Philox, plus range and determinism."""
    import numpy as np
    from curious_amd import ops
    from oracle.env import philox4x32
    seed, call, task, size = 0x1234567890ABCDEF, 7, 3, 20000
    eps = np.arange(0, 300, 3, dtype=np.int32)
    got = ops.store_slots_host(seed, call, task, size, eps)
    n = eps.size
    x, _, _, _ = philox4x32(eps.astype(np.uint32), np.full(n, task, np.uint32), np.full(n, call, np.uint32),
                            np.full(n, 31, np.uint32), np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))
    want = (x.astype(np.uint64) * np.uint64(size)) >> np.uint64(32)
    assert got.dtype == np.int64 and np.array_equal(got, want.astype(np.int64))
    assert got.min() >= 0 and got.max() < size and len(set(got.tolist())) > 90
    assert np.array_equal(got, ops.store_slots_host(seed, call, task, size, eps))
    assert not np.array_equal(got, ops.store_slots_host(seed, call + 1, task, size, eps))


def test_transposed_copy_description_and_routing_entry_without_gpu():
    """curious_ddpg_transposed is pure host code (workspace carve + parameter layout); curious_route_store_episodes
    validates and then fails at its launch on a machine without a GPU (sanitizer coverage of both)."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip('would launch kernels on fake pointers')
    from curious_amd import _lib
    L = _lib.lib()
    cfg = _lib.NetCfg()
    cfg.dimo, cfg.dimg, cfg.dimtd, cfg.layers, cfg.dimu, cfg.hidden, cfg.modular = 40, 12, 4, 3, 4, 256, 1
    cfg.max_u, cfg.gamma, cfg.clip_return, cfg.action_l2, cfg.clip_pos_returns = 1.0, 0.98, 50.0, 1.0, 1
    T = _lib.Transposed()
    base = 0x40000000
    assert L.curious_ddpg_transposed(C.byref(cfg), 256, C.c_void_p(base), C.byref(T)) == 0
    assert T.n == 4 and T.dim == 256
    offs, dsts = list(T.src_off)[:4], list(T.dst)[:4]
    n_Q, total = L.curious_param_offset_pi(C.byref(cfg)), L.curious_param_total(C.byref(cfg))
    assert offs[0] < offs[1] < n_Q <= offs[2] < offs[3] < total and all(o % 4 == 0 for o in offs)
    ws_bytes = 4 * L.curious_workspace_floats(C.byref(cfg), 256)
    assert all(base <= d and d + 4 * 256 * 256 <= base + ws_bytes for d in dsts) and len(set(dsts)) == 4
    cfg.hidden = 64                                             # no copies are kept for shapes off the row-local route
    assert L.curious_ddpg_transposed(C.byref(cfg), 256, C.c_void_p(base), C.byref(T)) == 0 and T.n == 0
    lay = _lib.Layout()
    lay.T, lay.row_stride = 50, 88
    fake = [C.c_void_p(0x10000000 + 0x1000000 * i) for i in range(8)]
    rc = L.curious_route_store_episodes(fake[0], fake[1], C.byref(lay), fake[2], 4, 4, 4096, fake[3], fake[4], 20000, 1, 1,
                                        None, fake[5], fake[6], fake[7], None)
    assert rc != 0 and b'at most' in L.curious_last_error()
    rc = L.curious_route_store_episodes(fake[0], fake[1], C.byref(lay), fake[2], 4, 4, 256, fake[3], fake[4], 20000, 1, 1,
                                        None, fake[5], fake[6], fake[7], None)
    assert rc != 0 and b'launch failed' in L.curious_last_error()


def test_round3_entry_points_host_code_without_gpu():
    """Options, the fault-word offset, the halves of the data-parallel expert update and the gradient call with a next
    batch: validation + descriptor / grid arithmetic up to the failing launch (sanitizer coverage of their host side)."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip('would launch kernels on fake pointers')
    from curious_amd import _lib
    L = _lib.lib()
    # options: read per call, unknown names refused
    assert L.curious_get_option(b'rows') == 1 and L.curious_set_option(b'rows', 0) == 0 and L.curious_get_option(b'rows') == 0
    assert L.curious_set_option(b'rows', 1) == 0
    assert L.curious_set_option(b'nope', 1) != 0 and b'unknown option' in L.curious_last_error()
    assert L.curious_set_option(b'xcd_map', 5) != 0 and L.curious_get_option(b'nope') == -1
    assert L.curious_get_option(b'resident') == 1 and L.curious_get_option(b'res_spins') == 1 << 20
    cfg = _lib.NetCfg()
    cfg.dimo, cfg.dimg, cfg.dimtd, cfg.layers, cfg.dimu, cfg.hidden, cfg.modular = 40, 12, 4, 3, 4, 256, 1
    cfg.max_u, cfg.gamma, cfg.clip_return, cfg.action_l2, cfg.clip_pos_returns = 1.0, 0.98, 50.0, 1.0, 1
    ws = L.curious_workspace_floats(C.byref(cfg), 256)
    off = L.curious_workspace_fault_offset(C.byref(cfg), 256)
    assert 0 < off < ws and off % 64 == 0 and L.curious_workspace_fault_offset(C.byref(cfg), 0) == -1
    T = _lib.Transposed()
    base = 0x40000000
    assert L.curious_ddpg_transposed(C.byref(cfg), 256, C.c_void_p(base), C.byref(T)) == 0
    assert T.fault == base + 4 * off                           # the optimiser finds the fault word through `keep`
    BL = _lib.BatchLayout()
    BL.off_o, BL.off_td, BL.off_u, BL.off_g, BL.off_o2, BL.off_g2, BL.off_r = 0, 40, 44, 48, 60, 100, 112
    BL.off_ag, BL.off_ag2, BL.off_extra, BL.stride = 113, 125, 137, 152
    fake = [C.c_void_p(0x10000000 + 0x1000000 * i) for i in range(12)]
    P = L.curious_param_total(C.byref(cfg))
    # strides are validated first
    rc = L.curious_ddpg_grads_experts(C.byref(cfg), 4, 1000, P, fake[0], fake[1], fake[2], C.byref(BL), 256, None, None,
                                      fake[3], fake[4], fake[5], fake[6], fake[7], 0, 0, None, None)
    assert rc != 0 and b'expert_stride' in L.curious_last_error()
    rc = L.curious_ddpg_grads_experts(C.byref(cfg), 4, 1 << 22, P - 64, fake[0], fake[1], fake[2], C.byref(BL), 256, None, None,
                                      fake[3], fake[4], fake[5], fake[6], fake[7], 0, 0, None, None)
    assert rc != 0 and b'grad_stride' in L.curious_last_error()
    # a next batch must be keyed by the call's own step counter and must not alias the current batch
    lay = _lib.Layout()
    lay.T, lay.dimo, lay.dimag, lay.dimg, lay.dimu, lay.dimtd, lay.dimextra = 50, 40, 12, 12, 4, 4, 13
    lay.off_o, lay.off_ag, lay.off_g, lay.off_u, lay.off_td, lay.off_extra, lay.row_stride = 0, 40, 52, 64, 68, 72, 88
    tasks, sp, rng = _lib.Tasks(), _lib.SampleParams(), _lib.SampleRng()
    tasks.ntasks = 4
    rng.prop_prefix, rng.cur_size, rng.nbuf, rng.step_ctr = fake[8].value, fake[9].value, 5, fake[10].value
    nb = _lib.NextBatch()
    nb.storage, nb.L, nb.tasks, nb.P, nb.rng = fake[11].value, C.pointer(lay), C.pointer(tasks), C.pointer(sp), C.pointer(rng)
    nb.batch = fake[2].value
    rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 256, None, None, fake[3], fake[4],
                              fake[5], fake[6], fake[10], 0, C.byref(nb), None)
    assert rc != 0 and b'staging' in L.curious_last_error()
    nb.batch = fake[7].value
    rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 256, None, None, fake[3], fake[4],
                              fake[5], fake[6], fake[9], 0, C.byref(nb), None)
    assert rc != 0 and b'step counter' in L.curious_last_error()
    # ... and with everything in order the call walks its host code to the launch (single agent, then 4 experts)
    rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 256, None, None, fake[3], fake[4],
                              fake[5], fake[6], fake[10], 0, C.byref(nb), None)
    assert rc != 0 and b'launch failed' in L.curious_last_error()
    rc = L.curious_ddpg_grads_experts(C.byref(cfg), 4, 1 << 22, P, fake[0], fake[1], fake[2], C.byref(BL), 256, None, None,
                                      fake[3], fake[4], fake[5], fake[6], fake[10], 1, 104729, C.byref(nb), None)
    assert rc != 0 and b'launch failed' in L.curious_last_error()
    # the optimiser half for the experts, without a gather (storage NULL)
    rc = L.curious_adam_update_and_sample_experts(4, 1 << 22, P, 104729, fake[0], fake[1], fake[2], fake[3],
                                                  L.curious_param_offset_pi(C.byref(cfg)),
                                                  P - L.curious_param_offset_pi(C.byref(cfg)), fake[4], fake[10], 0, 4096,
                                                  0.9, 0.1, 0.999, 0.001, 1e-8, None, 0, C.byref(lay), C.byref(tasks),
                                                  C.byref(sp), C.byref(rng), 256, None, C.byref(BL), C.byref(T), None)
    assert rc != 0 and b'launch failed' in L.curious_last_error()
    rc = L.curious_adam_update_and_sample_experts(4, 100, P, 0, fake[0], fake[1], fake[2], fake[3], 64, 64, fake[4],
                                                  fake[10], 0, 4096, 0.9, 0.1, 0.999, 0.001, 1e-8, None, 0, C.byref(lay),
                                                  C.byref(tasks), C.byref(sp), C.byref(rng), 256, None, C.byref(BL), None, None)
    assert rc != 0 and b'bad strides' in L.curious_last_error()


def test_round4_entry_points_host_code_without_gpu():
    """The fused IPC all-reduce + Adam call, the set-up calls of its peer mappings and the stamp offset of the lab build:
    argument validation up to the failing launch / allocation (sanitizer coverage of their host side, tools/sanitize_cpu.sh)."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip('would launch kernels on fake pointers')
    from curious_amd import _lib
    L = _lib.lib()
    cfg = _lib.NetCfg()
    cfg.dimo, cfg.dimg, cfg.dimtd, cfg.layers, cfg.dimu, cfg.hidden, cfg.modular = 40, 12, 4, 3, 4, 256, 1
    cfg.max_u, cfg.gamma, cfg.clip_return, cfg.action_l2, cfg.clip_pos_returns = 1.0, 0.98, 50.0, 1.0, 1
    ws = L.curious_workspace_floats(C.byref(cfg), 256)
    so = L.curious_workspace_stamps_offset(C.byref(cfg), 256)
    assert 0 < so < ws
    fake = [C.c_void_p(0x10000000 + 0x1000000 * i) for i in range(10)]
    P = _lib.IpcPeers()
    f32 = C.c_float

    def call(peers, n_Q, n_pi, keep=None, tab_len=4096):
        return L.curious_allreduce_adam_ipc(C.byref(peers) if peers is not None else None, fake[9], fake[0], fake[1], n_Q,
                                            n_pi, fake[2], fake[3], 0, tab_len, f32(0.9), f32(0.1), f32(0.999),
                                            f32(0.001), f32(1e-8), fake[4], fake[4], fake[5], 0, keep, None)
    assert call(None, 64, 64) != 0 and b'NULL argument' in L.curious_last_error()
    P.world, P.rank = 9, 0
    assert call(P, 64, 64) != 0 and b'world must be' in L.curious_last_error()
    P.world, P.rank = 2, 2
    assert call(P, 64, 64) != 0 and b'world must be' in L.curious_last_error()
    P.world, P.rank = 4, 1
    assert call(P, 64, 65) != 0 and b'divide by the world size' in L.curious_last_error()
    assert call(P, 64, 64) != 0 and b'rank 0 is not mapped' in L.curious_last_error()
    for r in range(4):
        P.grad[r], P.stage[r], P.flags[r] = fake[6].value + 4096 * r, fake[7].value + 4096 * r, fake[8].value + 64 * r
    P.stage[3] = None
    assert call(P, 64, 64) != 0 and b'rank 3 is not mapped' in L.curious_last_error()
    P.stage[3] = fake[7].value + 4096 * 3
    T = _lib.Transposed()
    T.n, T.dim = 2, 100                                        # not a multiple of the optimiser's tile
    assert call(P, 64, 64, C.byref(T)) != 0 and b'transposed copies' in L.curious_last_error()
    assert call(P, 64, 64, tab_len=0) != 0 and b'NULL argument' in L.curious_last_error()
    assert call(P, 64, 64) != 0 and b'launch failed' in L.curious_last_error()     # everything in order: up to the launch
    # the mappings: bad arguments are refused, without a device the allocation itself fails cleanly
    out = C.c_void_p()
    assert L.curious_ipc_alloc(0, C.byref(out)) != 0 and b'bad argument' in L.curious_last_error()
    assert L.curious_ipc_alloc(4096, None) != 0
    assert L.curious_ipc_alloc(4096, C.byref(out)) != 0 and b'hipExtMallocWithFlags' in L.curious_last_error()
    h = C.create_string_buffer(64)
    assert L.curious_ipc_export(None, h) != 0 and L.curious_ipc_import(None, C.byref(out)) != 0
    assert L.curious_ipc_close(None) == 0 and L.curious_ipc_free(None) == 0


def test_round5_entry_points_host_code_without_gpu():
    """The entry points of ABI 9 (virtual ranks) and the options of round 6: the several-rank rollout, the routed stores of
    several ranks (with and without the activity test inside), the counting env reset, the rank fields of the sampler
    description (64-bit offsets: pools of 33 GB) and `loss_rows` -- argument validation and descriptor arithmetic up to the
    failing launch (sanitizer coverage of their host side, tools/sanitize_cpu.sh)."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip('would launch kernels on fake pointers')
    from curious_amd import _lib
    L = _lib.lib()
    fake = [C.c_void_p(0x10000000 + 0x1000000 * i) for i in range(16)]
    lay = _lib.Layout()
    lay.T, lay.dimo, lay.dimag, lay.dimg, lay.dimu, lay.dimtd, lay.dimextra = 50, 40, 12, 12, 4, 4, 13
    lay.off_o, lay.off_ag, lay.off_g, lay.off_u, lay.off_td, lay.off_extra, lay.row_stride = 0, 40, 52, 64, 68, 72, 88
    tasks = _lib.Tasks()
    tasks.ntasks = 4
    for j in range(4):
        tasks.len[j] = 3
    # ---- routed store of several ranks
    def route(n_ranks=3, n_ep=4, ntasks=4, n_route=4, tab_stride=21, capacity=20000, n_pairs=fake[8], storage=fake[0]):
        return L.curious_route_store_episodes_ranks(storage, fake[1], C.byref(lay), fake[2], ntasks, n_route, n_ep, n_ranks,
                                                    fake[3], fake[4], tab_stride, capacity, 7, 1000003, 1, None, fake[5],
                                                    fake[6], n_pairs, None)
    assert route(n_ranks=0) != 0 and b'bad rank arguments' in L.curious_last_error()
    assert route(n_ranks=5000) != 0 and b'bad rank arguments' in L.curious_last_error()
    assert route(tab_stride=-1) != 0 and b'bad rank arguments' in L.curious_last_error()
    assert route(storage=None) != 0 and b'NULL argument' in L.curious_last_error()
    assert route(n_route=5) != 0 and b'bad task / capacity' in L.curious_last_error()
    assert route(capacity=1 << 31) != 0 and b'bad task / capacity' in L.curious_last_error()
    assert route(n_ep=1 << 20) != 0 and b'episodes per call' in L.curious_last_error()
    assert route(n_ranks=19, n_ep=1024, n_route=4) != 0 and b'65 535 pairs' in L.curious_last_error()
    assert route(n_ep=0) == 0                                                                         # nothing to do
    assert route() != 0 and b'launch failed' in L.curious_last_error()
    # ... with the activity test inside the routing launch
    def act_route(tk=tasks, active=fake[2], n_ranks=3):
        return L.curious_activity_route_store_episodes(fake[0], fake[1], C.byref(lay), C.byref(tk) if tk is not None else None,
                                                       72, active, 4, 4, n_ranks, fake[3], fake[4], 21, 20000, 7, 1000003, 1,
                                                       None, fake[5], fake[6], fake[8], None)
    assert act_route(tk=None) != 0 and b'NULL argument' in L.curious_last_error()
    assert act_route(active=None) != 0 and b'NULL argument' in L.curious_last_error()
    assert act_route(n_ranks=0) != 0 and b'bad rank arguments' in L.curious_last_error()
    assert act_route() != 0 and b'launch failed' in L.curious_last_error()
    # ---- the counting env reset
    E = _lib.EnvCfg()
    E.ntasks, E.dimo, E.T, E.seed = 4, 40, 50, 1
    def reset(env=E, n=8, counter=fake[14]):
        return L.curious_env_reset_count(C.byref(env), C.byref(lay), 0, fake[0], fake[1], fake[2], n, fake[3], fake[4],
                                         fake[5], fake[6], fake[7], None, counter, 50, None)
    bad = _lib.EnvCfg()
    bad.ntasks, bad.dimo, bad.T, bad.seed = 4, 200, 50, 1
    assert reset(env=bad) != 0 and b'at most 128' in L.curious_last_error()
    bad.dimo, bad.ntasks = 40, 5
    assert reset(env=bad) != 0                                                              # layout and env disagree
    assert reset() != 0 and b'launch failed' in L.curious_last_error()
    assert reset(counter=None) != 0 and b'launch failed' in L.curious_last_error()           # (the counter is optional)
    # ---- the several-rank rollout
    cfg = _lib.NetCfg()
    cfg.dimo, cfg.dimg, cfg.dimtd, cfg.layers, cfg.dimu, cfg.hidden, cfg.modular = 40, 12, 4, 3, 4, 256, 1
    cfg.max_u, cfg.gamma, cfg.clip_return, cfg.action_l2, cfg.clip_pos_returns = 1.0, 0.98, 50.0, 1.0, 1
    G = _lib.RankGroups()
    G.group, G.seed_stride, G.exploit = 2, 1000003, fake[13].value
    def rollout(groups=G, n=40, nsteps=50, t0=0, theta=fake[0]):
        return L.curious_policy_rollout_ranks(C.byref(cfg), theta, n, 200.0, fake[1], 0.2, 0.3, 5, 1, fake[14], fake[2], 4,
                                              C.byref(E), C.byref(lay), 0, fake[3], fake[4], t0, nsteps, fake[5], fake[6],
                                              fake[7], fake[8], fake[9], 72, 84, 0.05, fake[10], 0, None, None,
                                              C.byref(groups) if groups is not None else None, None)
    assert rollout(nsteps=0) != 0 and b'nsteps must be positive' in L.curious_last_error()
    neg = _lib.RankGroups()
    neg.group = -1
    assert rollout(groups=neg) != 0 and b'negative group size' in L.curious_last_error()
    assert rollout(theta=None) != 0 and b'NULL argument' in L.curious_last_error()
    assert rollout(t0=10, nsteps=50) != 0 and b't out of range' in L.curious_last_error()
    assert rollout() != 0 and b'launch failed' in L.curious_last_error()
    assert rollout(groups=None) != 0 and b'launch failed' in L.curious_last_error()
    # ---- the sampler description of several ranks: rank_rows must divide the batch; offsets are 64 bit (pools of 33 GB)
    sp, rng = _lib.SampleParams(), _lib.SampleRng()
    sp.future_p, sp.reward_eps, sp.clip_obs = 0.8, 0.05, 200.0
    BL = _lib.BatchLayout()
    BL.off_o, BL.off_td, BL.off_u, BL.off_g, BL.off_o2, BL.off_g2, BL.off_r = 0, 40, 44, 48, 60, 100, 112
    BL.off_ag, BL.off_ag2, BL.off_extra, BL.stride = 113, 125, 137, 152
    rng.prop_prefix, rng.cur_size, rng.buf_alias, rng.buf_task = fake[8].value, fake[9].value, fake[10].value, fake[11].value
    rng.nbuf, rng.step_ctr, rng.seed = 5, fake[12].value, 3
    buf_stride = 20000 * 51 * 88                                   # floats per buffer: 19 x 5 of them are 8.5 G floats
    def sample(n=19 * 256, rank_rows=256, tab_stride=21):
        rng.rank_rows, rng.rank_tab_stride, rng.rank_seed_stride = rank_rows, tab_stride, 1000003
        return L.curious_her_sample(fake[0], buf_stride, C.byref(lay), C.byref(tasks), C.byref(sp), None, C.byref(rng), n,
                                    fake[1], C.byref(BL), None)
    assert sample(rank_rows=100) != 0 and b'rank' in L.curious_last_error()
    assert sample(tab_stride=-1) != 0 and b'rank' in L.curious_last_error()
    assert sample() != 0 and b'launch failed' in L.curious_last_error()
    # ---- loss_rows: the batch must be a whole number of ranks
    cfg.loss_rows = 256
    nb_ = None
    rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 300, None, None, fake[3], fake[4],
                              fake[5], fake[6], fake[12], 0, nb_, None)
    assert rc != 0 and b'whole number of ranks' in L.curious_last_error()
    ws = L.curious_workspace_floats(C.byref(cfg), 19 * 256)
    assert ws > L.curious_workspace_floats(C.byref(cfg), 256)
    rc = L.curious_ddpg_grads(C.byref(cfg), fake[0], fake[1], fake[2], C.byref(BL), 19 * 256, None, None, fake[3], fake[4],
                              fake[5], fake[6], fake[12], 0, nb_, None)
    assert rc != 0 and b'launch failed' in L.curious_last_error()                 # (the 16-row form's host code: rows16)
    # ---- the options of round 6
    for name, dflt in ((b'rows16', 1280), (b'dw64', 0), (b'lab_rows_stamps', 0)):
        assert L.curious_get_option(name) == dflt, name
        assert L.curious_set_option(name, 7) == 0 and L.curious_get_option(name) == 7
        assert L.curious_set_option(name, dflt) == 0


def test_torch_library_loads_and_registers_every_op_without_a_gpu():
    """curious_amd/lib/libcurious_torch.so (csrc/torch_library.cpp: TORCH_LIBRARY_FRAGMENT(curious_hip, ...) in C++ over the
    C ABI) loads next to libcurious_hip.so and registers every op with its aliasing schema -- the tensor / scalar
    ops and the three hot entry points behind a registered descriptor; there is no CPU implementation to fall back to; an
    unknown descriptor handle is an error, not a wild pointer."""
    import torch
    from curious_amd.build import build_torch_library
    build_torch_library(verbose=False)
    import curious_amd.torch_ops  # noqa: F401
    native = ('polyak_update', 'adam_update', 'param_checksum', 'norm_update', 'norm_recompute', 'policy_forward', 'ddpg_grads')
    for name in native + ('her_sample', 'ddpg_update', 'policy_rollout', 'desc_register', 'desc_release'):
        assert hasattr(torch.ops.curious_hip, name), name
    schema = str(torch.ops.curious_hip.polyak_update.default._schema)
    assert 'Tensor(a!) target' in schema and 'float polyak' in schema, schema
    assert '-> (Tensor, Tensor)' in str(torch.ops.curious_hip.ddpg_grads.default._schema)
    # implemented in C++ (no Python kernel behind the CUDA key), nothing behind the CPU key
    assert torch._C._dispatch_has_kernel_for_dispatch_key('curious_hip::polyak_update', 'CUDA')
    assert not torch._C._dispatch_has_kernel_for_dispatch_key('curious_hip::polyak_update', 'CPU')
    with pytest.raises(NotImplementedError):
        torch.ops.curious_hip.polyak_update(torch.zeros(4), torch.zeros(4), 0.95)
    import curious_amd.torch_ops as T
    h = T.desc_create(n=3, B=5, seed=2 ** 63 + 1)
    assert h in T._DESCS and T._DESCS[h][0].n == 3 and T._DESCS[h][0].seed == 2 ** 63 + 1
    T.desc_free(h)
    assert h not in T._DESCS
    assert C.sizeof(T.TorchDesc) == 176                                 # (curious_torch_desc_t: 10 pointers, 4 x 8, 3 doubles, float, 8 x int32)
