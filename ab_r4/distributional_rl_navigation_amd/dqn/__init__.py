from .policy import DQNPolicy  # noqa: F401
from .agent import DQNAgent  # noqa: F401
