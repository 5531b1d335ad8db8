"""radiation_ppo_amd -- MI355X-native radiation-search PPO hot path.

Hand-written HIP kernels (gfx950) behind a plain C ABI (include/radsearch.h, librs_hip.so), with a
Python host side that mirrors the reference's two call surfaces:

  * radiation_ppo_amd.envs.RadSearch      <- gym_rad_search RadSearch (step/reset dict API)
  * radiation_ppo_amd.envs.RadSearchVec   <- the batched form the GPU actually runs
  * radiation_ppo_amd.train.train_PPO     <- algos/multiagent/train.py train_PPO(...).train()

There is no CPU fallback: importing the env classes loads librs_hip.so and raises if it is missing.
"""
__version__ = "0.1.0"
