"""Small helpers on the hot path.  Mirrors baselines/her/util.py:13-53,129-191 (TF-free parts)."""
import functools
import importlib
import inspect
import os

import numpy as np

from curious_amd import dist

# The reference resolves plugin strings "pkg.module:name" (util.py:40-46).  Its own strings are mapped onto the
# MI355X-native counterparts so that existing parameter dicts (config.py:26,50,60,84) keep working.
_REFERENCE_ALIASES = {
    'baselines.her.actor_critic:ActorCritic': 'curious_amd.actor_critic:ActorCritic',
    'baselines.her.actor_critic:MultiTaskActorCritic': 'curious_amd.actor_critic:MultiTaskActorCritic',
    'baselines.her.her:make_sample_her_transitions': 'curious_amd.her:make_sample_her_transitions',
    'baselines.her.her:make_sample_multi_task_her_transitions':
        'curious_amd.her:make_sample_multi_task_her_transitions',
}


def freeze_setup_objects():
    """Call once after a job is set up (modules imported, agents and workers built, graphs warm).  The training cycle is
    paced by the host (rollout flags are read one cycle late), so a host stall is a GPU stall: a full (generation-2)
    collection of CPython's cycle collector walks every container object the imports created -- ~50 ms with torch
    loaded, ten cycles' worth of GPU time -- and finds nothing.  gc.freeze() moves what exists now into the permanent
    generation; later collections only look at objects created by the loop itself."""
    import gc
    gc.collect()
    gc.freeze()


def thaw_setup_objects():
    """End of the job: hand the frozen objects back to the collector (a process that runs several jobs one after the
    other -- the test session -- would otherwise never free the cyclic garbage of the earlier ones)."""
    import gc
    gc.unfreeze()


def store_args(method):
    """Decorator: copy the call's arguments (with defaults) onto `self` (util.py:13-37)."""
    sig = inspect.signature(method)
    names = [p for p in sig.parameters][1:]

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        values = {n: p.default for n, p in sig.parameters.items()
                  if p.default is not inspect.Parameter.empty and p.kind != inspect.Parameter.VAR_KEYWORD}
        values.update(zip(names, args))
        values.update(kwargs)
        self.__dict__.update(values)
        return method(self, *args, **kwargs)
    return wrapper


def import_function(spec):
    spec = _REFERENCE_ALIASES.get(spec, spec)
    mod_name, fn_name = spec.split(':')
    return getattr(importlib.import_module(mod_name), fn_name)


def convert_episode_to_batch_major(episode):
    """Lists of per-step arrays [T(+1)][B, d] -> arrays [B, T(+1), d] (util.py:174-184)."""
    return {k: np.array(v).copy().swapaxes(0, 1) for k, v in episode.items()}


def transitions_in_episode_batch(episode_batch):
    shape = episode_batch['u'].shape
    return shape[0] * shape[1]


def mpi_average(value, weight=1):
    """Cross-rank mean of a scalar / list of scalars (util.py:141-146 -> mpi_moments.py:6-31).
    weight: the ranks this process stands for (virtual ranks: its value is already the mean over them) -- with an uneven
    layout (19 ranks on 8 processes: 3 3 3 2 2 2 2 2) the plain mean over processes is not the mean over ranks."""
    if isinstance(value, list) and len(value) == 0:
        value = [0.]
    if not isinstance(value, list):
        value = [value]
    x = np.asarray(value, dtype=np.float64)
    packed = np.array([x.sum() * weight, float(x.size) * weight])
    packed = dist.allreduce_sum_numpy(packed)
    return packed[0] / packed[1]


def mpi_fork(n, extra_mpi_args=()):
    """The reference re-executes itself under mpirun (util.py:148-171).  Here ranks are started by
    `python -m torch.distributed.run` (one process per GPU), so every process is a 'child'."""
    return 'child'


def find_save_path(dir, trial_id):
    i = 0
    while True:
        save_dir = dir + str(trial_id + i * 100) + '/'
        if not os.path.exists(save_dir):
            os.makedirs(save_dir)
            return save_dir
        i += 1


class BackgroundWriter:
    """Jobs (callables that write files) run one after the other on a worker thread: the training loop hands a policy
    snapshot over and goes on while it is pickled and written (train.py:195-205 on the training thread cost ~6 ms per
    epoch).  Order is kept; an exception of a job is raised at the next submit() / close()."""

    def __init__(self):
        import queue
        import threading
        self._q = queue.Queue()
        self._err = None
        self._t = threading.Thread(target=self._run, name='curious-writer', daemon=True)
        self._t.start()

    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            try:
                if self._err is None:
                    job()
            except BaseException as err:                            # kept for the training thread
                self._err = err

    def _check(self):
        if self._err is not None:
            err, self._err = self._err, None
            raise err

    def submit(self, job):
        self._check()
        self._q.put(job)

    def close(self):
        """Waits for every job handed in so far."""
        if self._t.is_alive():
            self._q.put(None)
            self._t.join()
        self._check()


def PolicySnapshot(policy):
    """What pickling a policy (DDPG.__getstate__: constructor arguments + weights as host arrays) needs, taken at one
    point in time: an uninitialised instance of the policy's class that carries only that state (`.state`) and pickles to
    exactly the bytes the policy itself would have produced then."""
    snap = object.__new__(type(policy))
    snap.__dict__['_snapshot_state'] = policy.__getstate__()
    snap.__dict__['state'] = snap.__dict__['_snapshot_state']
    return snap
