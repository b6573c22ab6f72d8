"""`marinenav_env.envs:MarineNavEnv` -- the entry point name the reference registers (marinenav_env/__init__.py:3-6)."""
from distributional_rl_navigation_amd.marinenav_env.env import Core, MarineNavEnv, Obstacle  # noqa: F401
