"""Row features for OneGNN, computed by the HIP row-sweep kernel (dense_sweeps.hip).

`compute_row_features(C)` keeps the reference's contract (gnn/features.py:161-243): a float64
(n, n) cost matrix in, a float32 (n, 21) descriptor out, (0, 0) for n == 0: 13 fp64 row
statistics cast to float32 followed by 8 positional encodings.

`compute_row_features_torch(cost)` is the device-resident entry the harness' CUDA branch calls
(scripts/gnn_benchmark.py:233-240).  Unlike the reference's torch variant (gnn/features.py:246-351,
float32 maths, ddof=1, argmin-only column counts) it returns exactly the same statistics as
`compute_row_features`, because the same fp64 kernel produces both.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from lap import _hip

POS_FREQS = (1, 2, 4, 8)
ROW_FEATURE_DIM = 13 + 2 * len(POS_FREQS)
TOPK = 16


def positional_encodings(n: int) -> np.ndarray:
    """(n, 8) float32 table, sin/cos(2 pi i f / max(1, n-1)) for f in (1,2,4,8)
    (gnn/features.py:21-31).  O(n) host work, cached per n by the pipeline."""
    if n <= 0:
        return np.zeros((0, 2 * len(POS_FREQS)), dtype=np.float32)
    pos = np.arange(n, dtype=np.float64)
    scale = max(1, n - 1)
    cols = []
    for f in POS_FREQS:
        ang = 2.0 * np.pi * pos * f / scale
        cols.append(np.sin(ang))
        cols.append(np.cos(ang))
    return np.stack(cols, axis=-1).astype(np.float32)


def compute_row_features(C: np.ndarray, return_topk: bool = False):
    """Host arrays in/out; the work happens on the GPU."""
    C = np.ascontiguousarray(np.asarray(C, dtype=np.float64))
    n = C.shape[0]
    if n == 0:
        out = np.zeros((0, 0), dtype=np.float32)
        return (out, np.zeros((0, TOPK), dtype=np.float32)) if return_topk else out
    if C.ndim != 2 or C.shape[1] != n:
        raise ValueError("compute_row_features on the MI355X path expects a square cost matrix")
    lib = _hip.require_device()
    feat = np.empty((n, ROW_FEATURE_DIM), dtype=np.float32)
    topk = np.empty((n, TOPK), dtype=np.float32)
    rc = lib.lapwarm_row_features(C.ctypes.data_as(_hip.c_dp), n, feat.ctypes.data_as(_hip.c_fp),
                                  topk.ctypes.data_as(_hip.c_fp))
    if _hip.check(rc, "compute_row_features") != 0:
        raise RuntimeError(f"compute_row_features failed (code {rc})")
    return (feat, topk) if return_topk else feat


_POSENC_CACHE = {}


def _posenc_device(n: int, device):
    import torch
    key = (n, str(device))
    if key not in _POSENC_CACHE:
        _POSENC_CACHE[key] = torch.from_numpy(positional_encodings(n)).to(device)
    return _POSENC_CACHE[key]


def row_features_device(C, return_topk: bool = True):
    """C: CUDA float64 tensor (B, n, n) or (n, n) -> feat (B, n, 21) f32 [, topk (B, n, 16) f32].
    Enqueued on the current torch stream; no host synchronisation."""
    import torch
    if not C.is_cuda or C.dtype != torch.float64:
        raise TypeError("row_features_device expects a CUDA float64 tensor")
    squeeze = C.ndim == 2
    if squeeze:
        C = C.unsqueeze(0)
    C = C.contiguous()
    B, n, m = C.shape
    if n != m or n < 1:
        raise ValueError("square, non-empty cost matrices expected")
    lib = _hip.require_device()
    feat = torch.empty((B, n, ROW_FEATURE_DIM), dtype=torch.float32, device=C.device)
    topk = torch.empty((B, n, TOPK), dtype=torch.float32, device=C.device)
    ws_bytes = lib.lapwarm_sweep_workspace_bytes(B, n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=C.device)
    pos = _posenc_device(n, C.device)
    stream = torch.cuda.current_stream(C.device).cuda_stream
    rc = lib.lapwarm_row_features_batched(C.data_ptr(), B, n, pos.data_ptr(), feat.data_ptr(),
                                          topk.data_ptr(), ws.data_ptr(), ws_bytes, ct.c_void_p(stream))
    if _hip.check(rc, "row_features_device") != 0:
        raise RuntimeError(f"row_features_device failed (code {rc})")
    if squeeze:
        feat, topk = feat[0], topk[0]
    return (feat, topk) if return_topk else feat


def compute_row_features_torch(cost):
    """Device-resident variant: CUDA tensor (n, n) of any float dtype -> (n, 21) float32 on the
    same device.  float32 inputs are widened exactly to float64 before the sweep."""
    import torch
    if cost.ndim != 2:
        raise ValueError("cost must be (n, m)")
    if cost.shape[0] == 0:
        return torch.zeros((0, 0), dtype=torch.float32, device=cost.device)
    if not cost.is_cuda:
        raise RuntimeError("compute_row_features_torch runs on the GPU only (no CPU fallback here)")
    return row_features_device(cost.to(torch.float64), return_topk=False)


def min_trick_device(C, u):
    """v[b][j] = min_i (C[b][i][j] - u[b][i]) in fp64 on the device (scripts/gnn_benchmark.py:262).
    C (B,n,n) f64 CUDA, u (B,n) any float dtype CUDA -> v (B,n) f64."""
    import torch
    C = C.contiguous()
    B, n, _ = C.shape
    u64 = u.to(torch.float64).contiguous()
    lib = _hip.require_device()
    v = torch.empty((B, n), dtype=torch.float64, device=C.device)
    ws_bytes = lib.lapwarm_sweep_workspace_bytes(B, n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=C.device)
    stream = torch.cuda.current_stream(C.device).cuda_stream
    rc = lib.lapwarm_colmin_batched(C.data_ptr(), B, n, u64.data_ptr(), v.data_ptr(), ws.data_ptr(),
                                    ws_bytes, ct.c_void_p(stream))
    if _hip.check(rc, "min_trick_device") != 0:
        raise RuntimeError(f"min_trick_device failed (code {rc})")
    return v
