"""Alias of dist_amd.utils.distributed (the reference's utils/distributed.py surface)."""
from .utils.distributed import *  # noqa: F401,F403
from .utils.distributed import GradReducer, init_process_group, destroy, barrier, all_reduce_max  # noqa: F401
