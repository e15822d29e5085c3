"""Environments: ``PBO_Env`` (the reference's problem + optimizer pair, a B = 1 view over the kernels) and ``BatchedPBO_Env``
(B lock-step pairs on one GPU)."""
from . import basic_environment as _env

PBO_Env = _env.PBO_Env
BatchedPBO_Env = _env.BatchedPBO_Env
__all__ = ['PBO_Env', 'BatchedPBO_Env']
