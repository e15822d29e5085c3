from .basic_environment import BatchedPBO_Env, PBO_Env
