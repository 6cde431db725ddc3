"""ctypes binding of liblapwarm_hip.so (C ABI: include/lapwarm_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or there is no HIP
device when a compute entry point is called, the call raises.
"""
from __future__ import annotations

import ctypes as ct
from pathlib import Path

import os

_PKG_ROOT = Path(__file__).resolve().parent.parent
# LAPWARM_HIP_LIB selects another build of the same library (e.g. the -DLAPWARM_STAMPS diagnostic one)
LIB_PATH = Path(os.environ.get("LAPWARM_HIP_LIB", _PKG_ROOT / "liblapwarm_hip.so"))

c_dp = ct.POINTER(ct.c_double)
c_fp = ct.POINTER(ct.c_float)
c_ip = ct.POINTER(ct.c_int)
c_llp = ct.POINTER(ct.c_longlong)
c_vp = ct.c_void_p

# name -> (restype, argtypes); every symbol declared in include/lapwarm_hip.h
SIGNATURES = {
    "lapjv_seeded": (ct.c_int, [c_dp, ct.c_int, ct.c_int, c_llp, c_llp, c_dp, c_dp, ct.c_double]),
    "lapwarm_lapjv_dense": (ct.c_int, [c_dp, ct.c_int, c_ip, c_ip]),
    "lapwarm_row_features": (ct.c_int, [c_dp, ct.c_int, c_fp, c_fp]),
    "lapwarm_min_trick": (ct.c_int, [c_dp, ct.c_int, c_dp, c_dp]),
    "lapwarm_row_min": (ct.c_int, [c_dp, ct.c_int, c_dp, c_dp]),
    "lapwarm_rowmin_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp]),
    "lapwarm_project_feasible": (ct.c_int, [c_dp, ct.c_int, c_dp, c_dp, ct.c_int, ct.c_double]),
    "lapwarm_reduce_costs": (ct.c_int, [c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_dp, c_dp]),
    "lapwarm_warmstart_lapjv": (ct.c_int, [c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_ip, c_ip]),
    "lapwarm_seeded_workspace_bytes": (ct.c_size_t, [ct.c_int, ct.c_int]),
    "lapwarm_lapjv_workspace_bytes": (ct.c_size_t, [ct.c_int, ct.c_int]),
    "lapwarm_seeded_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, ct.c_double, c_vp, c_vp,
                                          c_vp, c_vp, c_vp, ct.c_size_t, ct.c_int, c_vp]),
    "lapwarm_lapjv_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         ct.c_size_t, ct.c_int, c_vp]),
    "lapwarm_lapjv_duals_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               ct.c_size_t, ct.c_int, c_vp]),
    "lapwarm_sweep_workspace_bytes": (ct.c_size_t, [ct.c_int, ct.c_int]),
    "lapwarm_colmin_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp, ct.c_size_t, c_vp]),
    "lapwarm_row_features_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp, c_vp,
                                                ct.c_size_t, c_vp]),
    "lapwarm_project_round_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, c_vp, c_vp,
                                                 ct.c_size_t, c_vp]),
    "lapwarm_reduce_costs_batched": (ct.c_int, [c_vp, ct.c_int, ct.c_int, c_vp, c_vp, ct.c_int, c_vp, c_vp,
                                                c_vp, ct.c_size_t, c_vp]),
    "lapwarm_refine_aggregate_batched": (ct.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ct.c_int, ct.c_int,
                                                    ct.c_int, c_vp]),
    "lapwarm_refine_aggregate_wsum": (ct.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ct.c_int, ct.c_int, c_vp]),
    "lapwarm_profile_enable": (None, [ct.c_int]),
    "lapwarm_profile_last_solver_ms": (ct.c_double, []),
    "lapwarm_solver_uses_helpers": (ct.c_int, [ct.c_int]),
    "lapwarm_coop_members": (ct.c_int, [ct.c_int]),
    "lapwarm_last_error": (ct.c_char_p, []),
    "lapwarm_device_count": (ct.c_int, []),
    "lapwarm_build_info": (ct.c_char_p, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64 /
    libhsa-runtime64 and always loads those; if this library had already pulled in the copies
    under /opt/rocm, the process would hold two runtimes and torch would report no GPU
    (observed on the MI355X box: `torch.cuda.is_available()` False after a lap.lapjv call).
    So when a torch installation exists, its libamdhip64 is loaded first -- without importing
    torch -- and liblapwarm_hip's DT_NEEDED entry (same SONAME) binds to it.
    LAPWARM_SYSTEM_HIP=1 opts out (processes that never import torch)."""
    import importlib.util
    import os
    import sys
    if os.environ.get("LAPWARM_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if bundled.exists():
        ct.CDLL(str(bundled), mode=ct.RTLD_GLOBAL)


def hip_runtimes_in_process():
    """Paths of the libamdhip64 copies mapped into this process (diagnostics / tests)."""
    seen = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                seen.add(line.split()[-1])
    return sorted(seen)


def load():
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {_PKG_ROOT / 'csrc'}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback for the HIP path.")
        _share_hip_runtime_with_torch()
        lib = ct.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI drift, surface it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def require_device():
    lib = load()
    if lib.lapwarm_device_count() < 1:
        raise RuntimeError("liblapwarm_hip: no HIP device visible; the MI355X path has no CPU fallback")
    return lib


def last_error() -> str:
    return load().lapwarm_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    """Map library-level failures (not the reference's per-instance codes) to exceptions."""
    if rc <= -1000:
        raise RuntimeError(f"{what}: HIP runtime error {-1000 - rc}: {last_error()}")
    if rc == -5:
        raise ValueError(f"{what}: n exceeds the 16384 limit of this build")
    return rc
