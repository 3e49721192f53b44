"""Parameter dicts and factories.  Mirrors baselines/her/experiment/config.py (same function names, same dict keys).

prepare_params / configure_dims / configure_her / configure_buffer / configure_ddpg keep their signatures; the
environment comes from curious_amd.envs (synthetic GPU arm) unless params['make_env'] is already set to a factory
of real environments.
"""
import numpy as np

from curious_amd import logger
from curious_amd.ddpg import DDPG
from curious_amd.envs import EnvFactory, sparse_reward_fun
from curious_amd.replay_buffer import ReplayBuffer, make_pooled_buffers
from curious_amd.util import import_function

DEFAULT_ENV_PARAMS = {'FetchReach-v1': {'n_cycles': 10}}

# what both parameter sets share (config.py:21-45 / 55-83: same keys, same values)
_COMMON = dict(
    max_u=1., layers=3, hidden=256, Q_lr=0.001, pi_lr=0.001, buffer_size=int(1E6), polyak=0.95, action_l2=1.0,
    clip_obs=200., scope='ddpg', relative_goals=False,                                   # DDPG
    n_cycles=25, rollout_batch_size=2, n_batches=100, batch_size=256, n_test_rollouts=5, test_with_polyak=False,  # loop
    random_eps=0.3, noise_eps=0.2, her_replay_k=4, norm_eps=0.01, norm_clip=5)           # exploration, HER, normaliser

DEFAULT_PARAMS = dict(_COMMON, network_class='curious_amd.actor_critic:ActorCritic',
                      her_sampling_func='curious_amd.her:make_sample_her_transitions', queue_length=200)
MULTI_TASK_PARAMS = dict(_COMMON, network_class='curious_amd.actor_critic:MultiTaskActorCritic',
                         her_sampling_func='curious_amd.her:make_sample_multi_task_her_transitions',
                         queue_length=300, eps_task=0.4)

CACHED_ENVS = {}


def cached_make_env(make_env):
    if make_env not in CACHED_ENVS:
        CACHED_ENVS[make_env] = make_env()
    return CACHED_ENVS[make_env]


# what prepare_params moves into ddpg_params (config.py:129-139); a '_name' copy stays in the params for the logs
_DDPG_KEYS = ('hidden', 'layers', 'network_class', 'polyak', 'batch_size', 'Q_lr', 'pi_lr', 'norm_eps', 'norm_clip',
              'max_u', 'action_l2', 'clip_obs', 'scope', 'relative_goals')
_DEVICE_KEYS = ('rng_mode', 'use_graph', 'seed', 'async_store', 'virtual_ranks', 'rank_base',
                'total_ranks')                                     # MI355X-side knobs (not in the reference)


def prepare_params(kwargs):
    """config.py:106-144: env-derived entries (tasks, horizon, gamma), the learning-rate shorthand, the DDPG sub-dict."""
    if 'make_env' not in kwargs:
        kwargs['make_env'] = EnvFactory(kwargs['env_name'])
    probe = cached_make_env(kwargs['make_env'])
    core = probe.unwrapped
    if kwargs['structure'] == 'flat':
        core.set_flat_env()
    kwargs.update(nb_tasks=core.nb_tasks, tasks_g_id=core.tasks_g_id, tasks_ag_id=core.tasks_ag_id)
    assert hasattr(probe, '_max_episode_steps')
    horizon = probe._max_episode_steps
    kwargs['T'] = horizon
    probe.reset()
    if isinstance(kwargs['max_u'], list):
        kwargs['max_u'] = np.array(kwargs['max_u'])
    kwargs['gamma'] = 1. - 1. / horizon
    if 'lr' in kwargs:                                              # one rate for both networks
        kwargs['pi_lr'] = kwargs['Q_lr'] = kwargs.pop('lr')
    ddpg = {}
    for key in _DDPG_KEYS:
        ddpg[key] = kwargs['_' + key] = kwargs.pop(key)
    ddpg.update((key, kwargs[key]) for key in _DEVICE_KEYS if key in kwargs)
    kwargs['ddpg_params'] = ddpg
    return kwargs


def log_params(params, logger=logger):
    for name in sorted(params):
        logger.info('%s: %s' % (name, params[name]))


def configure_her(params):
    """config.py:152-174.  The reward closure carries the kernel-side reward description of the env."""
    env = cached_make_env(params['make_env'])
    env.reset()
    core = env.unwrapped
    if params['structure'] == 'flat':
        core.set_flat_env()
    spec = getattr(core, 'reward_spec', None)
    if spec is not None and not params.get('host_reward', False):
        reward_fun = sparse_reward_fun(spec)                        # evaluated inside the HER kernel
    else:
        # a real environment (gym_flowers): the reference's closure, evaluated on the host per sampled batch
        def reward_fun(ag_2, g, task_descr=None, info=None):        # config.py:158-159
            return core.compute_reward(achieved_goal=ag_2, goal=g, task_descr=task_descr, info=info)
    passed_on = ('tasks_ag_id', 'tasks_g_id', 'goal_replay', 'her_replay_k', 'task_replay')
    factory = import_function(params['her_sampling_func'])
    return factory(reward_fun=reward_fun, **{key: params[key] for key in passed_on})


def simple_goal_subtract(a, b):
    assert a.shape == b.shape
    return a - b


def dims_to_shapes(input_dims):
    return {name: ((dim,) if dim > 0 else ()) for name, dim in input_dims.items()}


def configure_buffer(dims, params):
    """config.py:184-216: nb_tasks+1 buffers when 'buffer' in task_replay, else one -- here on one HBM pool."""
    T = params['T']
    sampler = configure_her(params)
    # per-episode shapes: T + 1 observations / achieved goals, T of everything else (config.py:200-208)
    shapes = {name: ((T + 1 if name == 'o' else T),) + tail for name, tail in dims_to_shapes(dims).items()}
    shapes['g'] = (T, dims['g'])
    shapes['ag'] = (T + 1, dims['ag'])
    if params['structure'] in ('curious', 'task_experts'):
        shapes['task_descr'] = (T, dims['task_descr'])
        shapes['change'] = (T, dims['ag'])
    else:
        shapes.pop('task_descr', None)
    per_rollout = params['rollout_batch_size']
    capacity = params['buffer_size'] // per_rollout * per_rollout   # in transitions (config.py:210)
    if 'buffer' in params['task_replay']:
        return make_pooled_buffers(shapes, capacity, T, sampler, params['nb_tasks'] + 1, alias_from=5,
                                   n_ranks=int(params.get('virtual_ranks', 1)))
    return ReplayBuffer(shapes, capacity, T, sampler)


def configure_ddpg(dims, params, buffers, reuse=False, use_mpi=True, clip_return=True, t_id=None, **hooks):
    """config.py:219-254.  `hooks`: construction hooks of this implementation (curious_amd.experts.ExpertBank)."""
    sampler = configure_her(params)
    gamma = params['gamma']
    cached_make_env(params['make_env']).reset()
    ddpg_params = params['ddpg_params']
    ddpg_params.update(input_dims=dims.copy(), T=params['T'], gamma=gamma, clip_pos_returns=True,
                       clip_return=(1. / (1. - gamma)) if clip_return else np.inf,
                       rollout_batch_size=params['rollout_batch_size'], subtract_goals=simple_goal_subtract,
                       sample_transitions=sampler,
                       **{key: params[key] for key in ('task_replay', 'structure', 'tasks_ag_id', 'tasks_g_id',
                                                       'eps_task')})
    kw = dict(ddpg_params)
    if t_id is not None:
        # the reference's experts get different TensorFlow initialisations; here expert t_id draws its weights (and its
        # device RNG streams) from seed + t_id
        kw.update(t_id=t_id, seed=int(ddpg_params.get('seed', 0)) + int(t_id))
    kw['info'] = {'env_name': params['env_name']}
    kw.update(hooks)
    return DDPG(reuse=reuse, **kw, buffers=buffers, use_mpi=use_mpi)


def configure_dims(params):
    """config.py:257-275: observation / action / goal widths from the env's spaces, one 'info_<key>' entry per info value."""
    env = cached_make_env(params['make_env'])
    spaces = env.observation_space.spaces
    dims = dict(o=spaces['observation'].shape[0], u=env.action_space.shape[0], g=spaces['desired_goal'].shape[0],
                ag=spaces['achieved_goal'].shape[0], task_descr=params['nb_tasks'])
    for name, value in env.unwrapped.info.items():
        dims['info_' + name] = int(np.atleast_1d(np.array(value)).shape[0])
    return dims
