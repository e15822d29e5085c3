"""Problem base class of the plugin surface.

Mirrors the reference's ``Basic_Problem`` contract (reference:
src/problem/basic_problem.py:4-37): ``reset()`` zeroes the ``T1`` evaluation
timer, ``eval(x)`` accepts one individual (1-D -> scalar), a population
(2-D -> [N]) or any N-D stack (flattened to [-1, dim]) and accumulates the
evaluation wall time in milliseconds into ``T1``; ``func(x)`` is the
[N, dim] -> [N] objective supplied by the subclass.

In this framework ``func`` of the built-in suites runs on the GPU through the
C-ABI (``mbx_eval``); there is no CPU implementation in the product.
"""
import time

import numpy as np


class Basic_Problem:
    T1 = 0.0

    def reset(self):
        self.T1 = 0

    def eval(self, x):
        t0 = time.perf_counter()
        x = np.asarray(x)
        if x.ndim == 1:
            y = self.func(x.reshape(1, -1))[0]
        elif x.ndim == 2:
            y = self.func(x)
        else:
            y = self.func(x.reshape(-1, x.shape[-1]))
        self.T1 += (time.perf_counter() - t0) * 1000
        return y

    def func(self, x):
        raise NotImplementedError
