from .greedy import Greedy  # noqa: F401
