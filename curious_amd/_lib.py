"""ctypes binding of libcurious_hip.so (the C ABI declared in include/curious_hip.h).

The product path has no CPU fallback: `lib()` raises if the shared library has not been built
(`python -m curious_amd.build`) and every wrapper raises `CuriousHipError` on a non-zero status.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libcurious_hip.so')

ABI_VERSION = 10         # CURIOUS_ABI_VERSION of include/curious_hip.h
MAX_TASKS = 16
MAX_TASK_DIMS = 8

RELABEL_BUFFER_TASK, RELABEL_GIVEN_TASK, RELABEL_CURRENT_TASK, RELABEL_FLAT = 0, 1, 2, 3


class CuriousHipError(RuntimeError):
    pass


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ['T', 'dimo', 'dimag', 'dimg', 'dimu', 'dimtd', 'dimextra',
                 'off_o', 'off_ag', 'off_g', 'off_u', 'off_td', 'off_extra', 'row_stride']]


class BatchLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ['off_o', 'off_td', 'off_u', 'off_g', 'off_o2', 'off_g2', 'off_r', 'off_ag', 'off_ag2',
                 'off_extra', 'stride']]


class Tasks(C.Structure):
    _fields_ = [('ntasks', C.c_int32), ('len', C.c_int32 * MAX_TASKS),
                ('g_id', (C.c_int32 * MAX_TASK_DIMS) * MAX_TASKS),
                ('ag_id', (C.c_int32 * MAX_TASK_DIMS) * MAX_TASKS)]


class SampleParams(C.Structure):
    _fields_ = [('future_p', C.c_double), ('reward_eps', C.c_double), ('clip_obs', C.c_float),
                ('relative_goals', C.c_int32), ('relabel_mode', C.c_int32), ('flat_reward', C.c_int32)]


class SamplePlan(C.Structure):
    _fields_ = [('buf', C.c_void_p), ('ep', C.c_void_p), ('t', C.c_void_p), ('u_her', C.c_void_p),
                ('u_off', C.c_void_p), ('task_to_replay', C.c_void_p), ('out_row', C.c_void_p)]


class SampleRng(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('step_ctr', C.c_void_p), ('step_host', C.c_int64),
                ('prop_prefix', C.c_void_p), ('cur_size', C.c_void_p), ('buf_alias', C.c_void_p),
                ('buf_task', C.c_void_p), ('nbuf', C.c_int32), ('rank_rows', C.c_int32),
                ('rank_tab_stride', C.c_int64), ('rank_seed_stride', C.c_uint64)]


class NetCfg(C.Structure):
    _fields_ = [('dimo', C.c_int32), ('dimg', C.c_int32), ('dimu', C.c_int32), ('dimtd', C.c_int32),
                ('hidden', C.c_int32), ('layers', C.c_int32), ('modular', C.c_int32),
                ('max_u', C.c_float), ('gamma', C.c_float), ('clip_return', C.c_float), ('action_l2', C.c_float),
                ('clip_pos_returns', C.c_int32), ('normalize_obs', C.c_int32), ('norm_clip', C.c_float),
                ('loss_rows', C.c_int32)]


class AdamState(C.Structure):
    _fields_ = [('m', C.c_void_p), ('v', C.c_void_p), ('alpha_tab', C.c_void_p), ('tab_base', C.c_int64),
                ('tab_len', C.c_int32), ('alpha_Q', C.c_float), ('alpha_pi', C.c_float), ('beta1', C.c_float),
                ('one_minus_beta1', C.c_float), ('beta2', C.c_float), ('one_minus_beta2', C.c_float),
                ('epsilon', C.c_float), ('params_unchanged', C.c_int32)]


class Transposed(C.Structure):
    _fields_ = [('n', C.c_int32), ('dim', C.c_int32), ('src_off', C.c_int64 * 8), ('dst', C.c_void_p * 8),
                ('fault', C.c_void_p), ('fault_flag', C.c_int64)]


class IpcPeers(C.Structure):
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('grad', C.c_void_p * 8), ('stage', C.c_void_p * 8),
                ('flags', C.c_void_p * 8)]


class NextBatch(C.Structure):
    _fields_ = [('storage', C.c_void_p), ('buf_stride', C.c_int64), ('L', C.POINTER(Layout)),
                ('tasks', C.POINTER(Tasks)), ('P', C.POINTER(SampleParams)), ('rng', C.POINTER(SampleRng)),
                ('batch', C.c_void_p)]


class RankGroups(C.Structure):
    _fields_ = [('group', C.c_int32), ('reserved', C.c_int32), ('seed_stride', C.c_uint64), ('exploit', C.c_void_p)]


class EnvCfg(C.Structure):
    _fields_ = [('ntasks', C.c_int32), ('dimo', C.c_int32), ('T', C.c_int32), ('wrap', C.c_int32), ('seed', C.c_uint64)]


_P = C.c_void_p
_I32, _I64, _U64, _F, _D = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol include/curious_hip.h declares
PROTOTYPES = {
    'curious_last_error': (C.c_char_p, []),
    'curious_abi_version': (C.c_int, []),
    'curious_build_digest': (C.c_char_p, []),
    'curious_device_info': (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    'curious_prof_enable': (C.c_int, [C.c_int]),
    'curious_prof_kernel_count': (C.c_int, []),
    'curious_prof_kernel_name': (C.c_char_p, [C.c_int]),
    'curious_prof_collect': (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'curious_prof_launch_counts': (C.c_int, [C.POINTER(C.c_int64)]),
    'curious_set_option': (C.c_int, [C.c_char_p, _I64]),
    'curious_get_option': (_I64, [C.c_char_p]),
    'curious_workspace_fault_offset': (_I64, [C.POINTER(NetCfg), _I32]),
    'curious_workspace_stamps_offset': (_I64, [C.POINTER(NetCfg), _I32]),
    'curious_ipc_alloc': (C.c_int, [_I64, C.POINTER(C.c_void_p)]),
    'curious_ipc_free': (C.c_int, [_P]),
    'curious_ipc_export': (C.c_int, [_P, C.c_char_p]),
    'curious_ipc_import': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'curious_ipc_close': (C.c_int, [_P]),
    'curious_allreduce_adam_ipc': (C.c_int, [C.POINTER(IpcPeers), _P, _P, _P, _I64, _I64, _P, _P, _I64, _I32, _F, _F, _F,
                                             _F, _F, _P, _P, _P, _I32, C.POINTER(Transposed), _P]),
    'curious_her_sample': (C.c_int, [_P, _I64, C.POINTER(Layout), C.POINTER(Tasks), C.POINTER(SampleParams),
                                     C.POINTER(SamplePlan), C.POINTER(SampleRng), _I32, _P, C.POINTER(BatchLayout),
                                     _P]),
    'curious_store_episodes': (C.c_int, [_P, _P, C.POINTER(Layout), _P, _P, _I32, _P]),
    'curious_episode_activity': (C.c_int, [_P, C.POINTER(Layout), C.POINTER(Tasks), _I32, _I32, _P, _P]),
    'curious_norm_update': (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'curious_norm_scratch_doubles': (_I64, [_I32, _I32]),
    'curious_norm_recompute': (C.c_int, [_P, _P, _I32, _F, _F, _P]),
    'curious_norm_pair_scratch_doubles': (_I64, [_I32, _I32, _I32]),
    'curious_norm_update_pair': (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _F, _F, _P, _P,
                                           _P]),
    'curious_param_count_Q': (_I64, [C.POINTER(NetCfg)]),
    'curious_param_count_pi': (_I64, [C.POINTER(NetCfg)]),
    'curious_param_offset_pi': (_I64, [C.POINTER(NetCfg)]),
    'curious_param_total': (_I64, [C.POINTER(NetCfg)]),
    'curious_workspace_floats': (_I64, [C.POINTER(NetCfg), _I32]),
    'curious_ddpg_grads': (C.c_int, [C.POINTER(NetCfg), _P, _P, _P, C.POINTER(BatchLayout), _I32, _P, _P, _P, _P,
                                     _P, _P, _P, _I32, C.POINTER(NextBatch), _P]),
    'curious_ddpg_transposed': (C.c_int, [C.POINTER(NetCfg), _I32, _P, C.POINTER(Transposed)]),
    'curious_ddpg_update': (C.c_int, [C.POINTER(NetCfg), _P, _P, _P, C.POINTER(BatchLayout), _I32, _P, _P, _P, _P,
                                      _P, _P, _P, C.POINTER(AdamState), C.POINTER(NextBatch), _P]),
    'curious_ddpg_update_experts': (C.c_int, [C.POINTER(NetCfg), _I32, _I64, _I64, _U64, _P, _P, _P,
                                              C.POINTER(BatchLayout), _I32, _P, _P, _P, _P, _P, _P, _P,
                                              C.POINTER(AdamState), C.POINTER(NextBatch), _P]),
    'curious_ddpg_grads_experts': (C.c_int, [C.POINTER(NetCfg), _I32, _I64, _I64, _P, _P, _P, C.POINTER(BatchLayout),
                                             _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _U64, C.POINTER(NextBatch), _P]),
    'curious_adam_update_and_sample_experts': (C.c_int, [_I32, _I64, _I64, _U64, _P, _P, _P, _P, _I64, _I64, _P, _P,
                                                         _I64, _I32, _F, _F, _F, _F, _F, _P, _I64, C.POINTER(Layout),
                                                         C.POINTER(Tasks), C.POINTER(SampleParams),
                                                         C.POINTER(SampleRng), _I32, _P, C.POINTER(BatchLayout),
                                                         C.POINTER(Transposed), _P]),
    'curious_policy_forward': (C.c_int, [C.POINTER(NetCfg), _P, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _I32, _F,
                                         _I32, _P, _P, _P, _P, _P, _P]),
    'curious_action_noise': (C.c_int, [_P, _I32, _I32, _I32, _D, _D, _D, _P, _P, _P, _U64, _U64, _P]),
    'curious_adam_update': (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I64, _I32, C.POINTER(C.c_float), _F, _F,
                                      _F, _F, _F, C.POINTER(Transposed), _P]),
    'curious_adam_update_and_sample': (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _I64, _I32, C.POINTER(C.c_float),
                                                 _F, _F, _F, _F, _F, _P, _I64, C.POINTER(Layout), C.POINTER(Tasks),
                                                 C.POINTER(SampleParams), C.POINTER(SampleRng), _I32, _P,
                                                 C.POINTER(BatchLayout), C.POINTER(Transposed), _P]),
    'curious_polyak_update': (C.c_int, [_P, _P, _I64, _F, _F, _P]),
    'curious_param_checksum': (C.c_int, [_P, _I64, _P, _P]),
    'curious_policy_act_env_step': (C.c_int, [C.POINTER(NetCfg), _P, _I32, _F, _P, _D, _D, _U64, _U64, _P, _P, _I32,
                                              C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _I32, _P, _P, _P, _P,
                                              _P, _I32, _I32, _D, _P, _P]),
    'curious_policy_rollout': (C.c_int, [C.POINTER(NetCfg), _P, _I32, _F, _P, _D, _D, _U64, _U64, _P, _P, _I32,
                                         C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _I32, _I32, _P, _P, _P, _P,
                                         _P, _I32, _I32, _D, _P, _P]),
    'curious_policy_act_env_step_stats': (C.c_int, [C.POINTER(NetCfg), _P, _I32, _F, _P, _D, _D, _U64, _U64, _P, _P, _I32,
                                                    C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _I32, _P, _P, _P,
                                                    _P, _P, _I32, _I32, _D, _P, _I32, _P, _P, _P]),
    'curious_policy_rollout_stats': (C.c_int, [C.POINTER(NetCfg), _P, _I32, _F, _P, _D, _D, _U64, _U64, _P, _P, _I32,
                                               C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _I32, _I32, _P, _P, _P,
                                               _P, _P, _I32, _I32, _D, _P, _I32, _P, _P, _P]),
    'curious_policy_rollout_ranks': (C.c_int, [C.POINTER(NetCfg), _P, _I32, _F, _P, _D, _D, _U64, _U64, _P, _P, _I32,
                                               C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _I32, _I32, _P, _P, _P,
                                               _P, _P, _I32, _I32, _D, _P, _I32, _P, _P, C.POINTER(RankGroups), _P]),
    'curious_route_store_episodes': (C.c_int, [_P, _P, C.POINTER(Layout), _P, _I32, _I32, _I32, _P, _P, _I64, _U64, _U64,
                                               _P, _P, _P, _P, _P]),
    'curious_route_store_episodes_ranks': (C.c_int, [_P, _P, C.POINTER(Layout), _P, _I32, _I32, _I32, _I32, _P, _P, _I64,
                                                     _I64, _U64, _U64, _U64, _P, _P, _P, _P, _P]),
    'curious_activity_route_store_episodes': (C.c_int, [_P, _P, C.POINTER(Layout), C.POINTER(Tasks), _I32, _P, _I32, _I32,
                                                        _I32, _P, _P, _I64, _I64, _U64, _U64, _U64, _P, _P, _P, _P, _P]),
    'curious_store_slots_host': (C.c_int, [_U64, _U64, _I32, _I64, _I32, _P, _P]),
    'curious_counter_add': (C.c_int, [_P, _I64, _P]),
    'curious_env_reset': (C.c_int, [C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _P, _I32, _P, _P, _P, _P,
                                    _P, _P, _P]),
    'curious_env_reset_count': (C.c_int, [C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _P, _I32, _P, _P, _P, _P,
                                          _P, _P, _P, _I64, _P]),
    'curious_env_step': (C.c_int, [C.POINTER(EnvCfg), C.POINTER(Layout), _I32, _P, _P, _P, _I32, _I32, _I32, _P,
                                   _P, _P, _P, _P, _I32, _I32, _D, _P, _P]),
}

_lib = None


def lib():
    """The loaded library.  Raises (loudly) when it is missing -- there is no fallback path."""
    global _lib
    if _lib is None:
        override = os.environ.get('CURIOUS_LIB')        # e.g. the ASan / UBSan host build of tools/sanitize_cpu.sh
        path = override or LIB_PATH
        if not os.path.exists(path):
            raise CuriousHipError(
                'libcurious_hip.so is not built (%s). Run `python -m curious_amd.build` '
                '(hipcc --offload-arch=gfx950); curious_amd has no CPU fallback.' % path)
        # PyTorch-ROCm bundles its own libamdhip64; the library must bind to THAT runtime instance (device pointers
        # and streams are shared with torch), so torch is imported -- and its HIP runtime loaded -- first.
        import torch  # noqa: F401
        hip = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        if os.path.exists(hip):
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)            # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.curious_abi_version() != ABI_VERSION:
            raise CuriousHipError('libcurious_hip ABI version %d, the binding expects %d: rebuild with '
                                  '`python -m curious_amd.build`' % (L.curious_abi_version(), ABI_VERSION))
        from curious_amd.build import source_digest
        built = L.curious_build_digest().decode()
        if not override and built != source_digest():
            raise CuriousHipError('libcurious_hip.so was built from other sources than the ones in curious_amd/csrc '
                                  '(digest %s...): rebuild with `python -m curious_amd.build`' % built[:12])
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().curious_last_error()
        raise CuriousHipError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None as c_void_p value."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_tasks(tasks_ag_id, tasks_g_id):
    nb = len(tasks_g_id)
    if nb > MAX_TASKS:
        raise CuriousHipError('at most %d tasks are supported' % MAX_TASKS)
    t = Tasks()
    t.ntasks = nb
    for j in range(nb):
        g_ids = list(tasks_g_id[j])
        ag_ids = list(tasks_ag_id[j])[:len(g_ids)]      # her.py:148-149
        if len(g_ids) > MAX_TASK_DIMS:
            raise CuriousHipError('at most %d goal dims per task are supported' % MAX_TASK_DIMS)
        t.len[j] = len(g_ids)
        for k, (a, b) in enumerate(zip(ag_ids, g_ids)):
            t.ag_id[j][k] = int(a)
            t.g_id[j][k] = int(b)
    return t
