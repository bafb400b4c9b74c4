from . import common_ops, pointgroup_ops, hais_ops, softgroup_ops  # noqa: F401
