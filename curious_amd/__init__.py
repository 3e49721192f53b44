"""curious_amd -- MI355X-native hot path of flowersteam/curious (baselines/her rollout-and-update loop).

Host-side mirror of the reference's Python interfaces (DDPG, ReplayBuffer, HER samplers, Normalizer, MpiAdam,
RolloutWorker, CompetenceQueue, config/train) over the C ABI of libcurious_hip.so (include/curious_hip.h).
There is no CPU compute path: GPU work fails loudly when the library or the GPU is missing.
"""
__version__ = '0.1.0'
