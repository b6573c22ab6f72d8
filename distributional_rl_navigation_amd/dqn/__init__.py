from .policy import DQNPolicy  # noqa: F401
