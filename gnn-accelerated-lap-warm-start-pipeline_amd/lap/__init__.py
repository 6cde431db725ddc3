"""`lap` -- MI355X-native stand-in for the reference's fork of the `lap` package.

Same surface as /root/reference/LAP/lap/__init__.py:15-27 for the functions on the
warm-start hot path:

    lapjv(cost, extend_cost=False, cost_limit=inf, return_cost=True) -> (opt, x, y)   int32
    lapjv_seeded(C, u, v, eps=1e-12)                                  -> (x, y, cost)  int64

Both run on the GPU through liblapwarm_hip.so (hand-written HIP, gfx950); there is no CPU
implementation in this package.  `lapmod` (sparse LAPMOD) is outside the hot path and raises.
"""
from ._lapjv import lapjv, LARGE_ as LARGE, FP_1_ as FP_1, FP_2_ as FP_2, FP_DYNAMIC_ as FP_DYNAMIC
from ._seeded_jv import lapjv_seeded

__version__ = "0.5.12+mi355x"


def lapmod(*args, **kwargs):
    raise NotImplementedError(
        "lapmod (sparse LAPMOD, LAP/_lapjv_cpp/lapmod.cpp) is not part of the warm-start hot path "
        "and is not built here")


__all__ = ["lapjv", "lapjv_seeded", "lapmod", "FP_1", "FP_2", "FP_DYNAMIC", "LARGE"]
