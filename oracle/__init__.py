"""CPU oracle for the CURIOUS rollout-and-update hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the reference
algorithm (flowersteam/curious, ``baselines/her``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product (``curious_amd``) never does and fails loudly when its HIP library is
missing.

Parity status
-------------
* pinned by golden vectors captured from the imported reference
  (``tools/gen_golden.py`` -> ``tests/golden/*.npz``): HER samplers, ReplayBuffer,
  CompetenceQueue, the epsilon-proportional task probabilities, episode layout of
  RolloutWorker, MpiAdam.update arithmetic, Normalizer.update/synchronize.
* restated from source only (TensorFlow-1 graph code cannot be imported here; checked
  by float64 finite differences + an independent torch-autograd implementation):
  MultiTaskActorCritic / ActorCritic forward, DDPG losses and gradients, Polyak,
  Normalizer.recompute_stats / normalize, DDPG.get_actions post-processing,
  DDPG.sample_batch proportions.
* parity unpinned upstream: the reward and goal construction live in the un-vendored
  third-party ``gym_flowers`` package (no version pinned by the reference); the reward
  used here is the documented sparse per-task L2 threshold (``oracle/reward.py``).
"""
