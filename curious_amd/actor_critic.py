"""Network descriptors.  Mirrors the class names of baselines/her/actor_critic.py so that the reference's
`network_class` plugin strings keep resolving (config.py:26,60 -> ddpg.py:63).

The reference classes build TensorFlow graphs; here a class only says which of the two architectures the fused HIP
kernels (curious_amd/csrc/mlp.hip) should evaluate:
  MultiTaskActorCritic  state branch [o | task_descr (| u/max_u)] + bias-free goal branch   (actor_critic.py:51-98)
  ActorCritic           single input [o | g (| u/max_u)]                                     (actor_critic.py:5-48)
"""


class ActorCritic:
    modular = False


class MultiTaskActorCritic:
    modular = True
