"""Import harness for the upstream reference (container-only; never runs on the GPU box).

Stubs the third-party modules that are absent in this image (none of them is on the
hot path) and puts /root/reference/src on sys.path so that the
reference's `problem`, `optimizer`, `agent`, `environment` packages import.
Used only by tools/gen_golden_*.py to produce the fixtures under tests/golden/.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _noop(*a, **k):
            return None
        return _noop


def install():
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference checkout not present (expected on the build container only)")
    for name in ("deap", "deap.base", "deap.creator", "deap.tools", "deap.algorithms", "deap.cma",
                 "skopt", "cmaes"):
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)


def ref_config(argv, scratch):
    """Build the reference's config Namespace with its own get_config()."""
    install()
    from config import get_config  # noqa: reference module
    os.makedirs(scratch, exist_ok=True)
    return get_config(list(argv) + ["--agent_save_dir", scratch + "/agent_model/",
                                    "--log_dir", scratch + "/log/"])
