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


def prepare_params(kwargs):
    """config.py:106-144."""
    ddpg_params = dict()
    env_name = kwargs['env_name']
    if 'make_env' not in kwargs:
        kwargs['make_env'] = EnvFactory(env_name)
    tmp_env = cached_make_env(kwargs['make_env'])
    if kwargs['structure'] == 'flat':
        tmp_env.unwrapped.set_flat_env()
    kwargs['nb_tasks'] = tmp_env.unwrapped.nb_tasks
    kwargs['tasks_g_id'] = tmp_env.unwrapped.tasks_g_id
    kwargs['tasks_ag_id'] = tmp_env.unwrapped.tasks_ag_id
    assert hasattr(tmp_env, '_max_episode_steps')
    kwargs['T'] = tmp_env._max_episode_steps
    tmp_env.reset()
    kwargs['max_u'] = np.array(kwargs['max_u']) if isinstance(kwargs['max_u'], list) else kwargs['max_u']
    kwargs['gamma'] = 1. - 1. / kwargs['T']
    if 'lr' in kwargs:
        kwargs['pi_lr'] = kwargs['lr']
        kwargs['Q_lr'] = kwargs['lr']
        del kwargs['lr']
    for name in ['hidden', 'layers', 'network_class', 'polyak', 'batch_size', 'Q_lr', 'pi_lr', 'norm_eps',
                 'norm_clip', 'max_u', 'action_l2', 'clip_obs', 'scope', 'relative_goals']:
        ddpg_params[name] = kwargs[name]
        kwargs['_' + name] = kwargs[name]
        del kwargs[name]
    for name in ['rng_mode', 'use_graph', 'seed', 'async_store']:   # MI355X-side knobs (not in the reference)
        if name in kwargs:
            ddpg_params[name] = kwargs[name]
    kwargs['ddpg_params'] = ddpg_params
    return kwargs


def log_params(params, logger=logger):
    for key in sorted(params.keys()):
        logger.info('{}: {}'.format(key, params[key]))


def configure_her(params):
    """config.py:152-174.  The reward closure carries the kernel-side reward description of the env."""
    env = cached_make_env(params['make_env'])
    env.reset()
    if params['structure'] == 'flat':
        env.unwrapped.set_flat_env()
    spec = getattr(env.unwrapped, 'reward_spec', None)
    if spec is not None and not params.get('host_reward', False):
        reward_fun = sparse_reward_fun(spec)                        # evaluated inside the HER kernel
    else:
        # a real environment (gym_flowers): the reference's closure, evaluated on the host per sampled batch
        def reward_fun(ag_2, g, task_descr=None, info=None):        # config.py:158-159
            return env.unwrapped.compute_reward(achieved_goal=ag_2, goal=g, task_descr=task_descr, info=info)
    her_params = {
        'reward_fun': reward_fun,
        'tasks_ag_id': params['tasks_ag_id'],
        'tasks_g_id': params['tasks_g_id'],
        'goal_replay': params['goal_replay'],
        'her_replay_k': params['her_replay_k'],
        'task_replay': params['task_replay'],
    }
    her_sampling_func = import_function(params['her_sampling_func'])
    return her_sampling_func(**her_params)


def simple_goal_subtract(a, b):
    assert a.shape == b.shape
    return a - b


def dims_to_shapes(input_dims):
    return {key: tuple([val]) if val > 0 else tuple() for key, val in input_dims.items()}


def configure_buffer(dims, params):
    """config.py:184-216: nb_tasks+1 buffers when 'buffer' in task_replay, else one -- here on one HBM pool."""
    T = params['T']
    structure = params['structure']
    buffer_size = params['buffer_size']
    rollout_batch_size = params['rollout_batch_size']
    task_replay = params['task_replay']
    sample_her_transitions = configure_her(params)
    input_shapes = dims_to_shapes(dims)
    dimg, dimag = dims['g'], dims['ag']
    buffer_shapes = {key: (T if key != 'o' else T + 1, *input_shapes[key]) for key, val in input_shapes.items()}
    buffer_shapes['g'] = (buffer_shapes['g'][0], dimg)
    buffer_shapes['ag'] = (T + 1, dimag)
    buffer_size = (buffer_size // rollout_batch_size) * rollout_batch_size
    if structure in ('curious', 'task_experts'):
        buffer_shapes['task_descr'] = (buffer_shapes['g'][0], dims['task_descr'])
        buffer_shapes['change'] = (buffer_shapes['g'][0], dimag)
    else:
        buffer_shapes.pop('task_descr', None)
    if 'buffer' in task_replay:
        return make_pooled_buffers(buffer_shapes, buffer_size, T, sample_her_transitions, params['nb_tasks'] + 1,
                                   alias_from=5)
    return ReplayBuffer(buffer_shapes, buffer_size, T, sample_her_transitions)


def configure_ddpg(dims, params, buffers, reuse=False, use_mpi=True, clip_return=True, t_id=None, **hooks):
    """config.py:219-254.  `hooks`: construction hooks of this implementation (curious_amd.experts.ExpertBank)."""
    sample_her_transitions = configure_her(params)
    gamma = params['gamma']
    rollout_batch_size = params['rollout_batch_size']
    ddpg_params = params['ddpg_params']
    input_dims = dims.copy()
    env = cached_make_env(params['make_env'])
    env.reset()
    ddpg_params.update({'input_dims': input_dims,
                        'T': params['T'],
                        'clip_pos_returns': True,
                        'clip_return': (1. / (1. - gamma)) if clip_return else np.inf,
                        'rollout_batch_size': rollout_batch_size,
                        'subtract_goals': simple_goal_subtract,
                        'sample_transitions': sample_her_transitions,
                        'gamma': gamma,
                        'task_replay': params['task_replay'],
                        'structure': params['structure'],
                        'tasks_ag_id': params['tasks_ag_id'],
                        'tasks_g_id': params['tasks_g_id'],
                        'eps_task': params['eps_task']})
    kw = dict(ddpg_params)
    if t_id is not None:
        # the reference's experts get different TensorFlow initialisations; here expert t_id draws its weights (and its
        # device RNG streams) from seed + t_id
        kw.update({'t_id': t_id, 'seed': int(ddpg_params.get('seed', 0)) + int(t_id)})
    kw['info'] = {'env_name': params['env_name']}
    kw.update(hooks)
    return DDPG(reuse=reuse, **kw, buffers=buffers, use_mpi=use_mpi)


def configure_dims(params):
    """config.py:257-275."""
    env = cached_make_env(params['make_env'])
    info = env.unwrapped.info
    dims = {
        'o': env.observation_space.spaces['observation'].shape[0],
        'u': env.action_space.shape[0],
        'g': env.observation_space.spaces['desired_goal'].shape[0],
        'ag': env.observation_space.spaces['achieved_goal'].shape[0],
    }
    dims['task_descr'] = params['nb_tasks']
    for key, value in info.items():
        value = np.array(value)
        if value.ndim == 0:
            value = value.reshape(1)
        dims['info_{}'.format(key)] = value.shape[0]
    return dims
