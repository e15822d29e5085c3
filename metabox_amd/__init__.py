"""metabox_amd — MI355X-native MetaBBO rollout engine behind MetaBox's Agent/Optimizer/Environment surface.

Hot path: thousands of independent (problem x run) optimizer instances step in lock-step inside
hand-written gfx950 HIP kernels reached through the C-ABI in include/mbx.h (libmbx.so).
"""
__version__ = "0.1.0"
