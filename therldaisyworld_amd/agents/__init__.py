from .greedy import Greedy  # noqa: F401
from .mlp import MLP  # noqa: F401
