"""Import shim: lets the reference's callers keep their import paths.

`gym.make('marinenav_env:marinenav_env-v0', seed=..., schedule=...)` (train_IQN_model.py:96,100) imports a module
named `marinenav_env` and then looks the id up; `import marinenav_env.envs.marinenav_env as marinenav_env`
(run_experiments.py:10) expects `MarineNavEnv` there.  Both resolve to the MI355X facade
(distributional_rl_navigation_amd/marinenav_env/env.py).  gym itself is optional: without it, use
`distributional_rl_navigation_amd.marinenav_env.env.make`, which takes the same id string.
"""
from distributional_rl_navigation_amd.marinenav_env.env import MarineNavEnv, make, register_with_gym

REGISTERED = register_with_gym()
__all__ = ["MarineNavEnv", "make", "REGISTERED"]
