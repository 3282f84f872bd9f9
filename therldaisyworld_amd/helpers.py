"""Keyword plumbing of the drop-in constructor (ref: daisy/helpers.py:3-8)."""


def query_kwargs(key, default, **kwargs):
    """Value of ``key`` in ``kwargs`` or ``default``; unknown keys are simply never asked for, which
    is how the reference ends up ignoring e.g. ``batch_size=...``."""
    return kwargs.get(key, default)
