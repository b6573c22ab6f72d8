from .vec_env import VecMarineNavEnv  # noqa: F401
