"""`import marinenav_env.envs.marinenav_env as marinenav_env` (run_experiments.py:10) -> MarineNavEnv, Core, Obstacle."""
from distributional_rl_navigation_amd.marinenav_env.env import Core, MarineNavEnv, Obstacle  # noqa: F401
