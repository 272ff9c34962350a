"""MI355X-native GP fit path (drop-in for the hot path of Spatial_GP_repo/utils.py)."""
__all__ = ["synthetic", "engine", "build"]

import os as _os

# The fit path keeps several HIP streams busy at once (the two factorisation chains, their look-ahead
# products, the units of a group); the ROCm runtime multiplexes all streams of a process onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default), and two streams that land on the same queue serialise --
# which pair does depends on the order the streams were created in (measured on MI355X: the same unit of work
# at N=4096 takes 7.2 or 11.5 ms depending on whether an unrelated context was created first).  Eight queues
# give every stream of one evaluation its own.  Read by the runtime when HIP initialises, i.e. at the first
# device call of the process: import this package before touching the GPU; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
