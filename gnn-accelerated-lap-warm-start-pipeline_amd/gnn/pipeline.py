"""Warm-start pipeline on the MI355X: C -> row features -> OneGNN u -> v = min(C - u) ->
lapjv_seeded(C, u, v), batched over independent cost matrices that stay resident in HBM.

Two entry points:
  * `GNNPredictor` -- the harness-facing object of the reference (scripts/gnn_benchmark.py:56-289):
    `GNNPredictor(model_path).predict(C) -> (u, v)` with float64 NumPy in/out;
  * `WarmStartPipeline` -- the batched device API: torch CUDA tensors
    in/out, everything enqueued on the current stream, no host round trips between stages.
"""
from __future__ import annotations

import ctypes as ct
from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch

from lap import _hip
from .features import ROW_FEATURE_DIM, min_trick_device, row_features_device
from .one_gnn import OneGNN

STATS_FIELDS = ("branch", "tight_edges", "free_rows", "arr_fired", "paths", "finds", "scan_steps",
                "scan_elems", "init_elems", "colred_elems", "transfer_rows", "arr_iters", "err",
                "r13", "r14", "r15")


def load_checkpoint(path, device="cpu") -> Tuple[OneGNN, dict]:
    """Build a OneGNN from a reference checkpoint.  Both schemas are accepted: flat
    (gnn/train_one_gnn.py:409-420) and nested under 'config'
    (gnn/train_progressive_clean.py:601-608); merged as scripts/gnn_benchmark.py:80-119 does.
    Files are read with weights_only=True (nothing in them is executed)."""
    ckpt = torch.load(str(path), map_location=device, weights_only=True)
    if isinstance(ckpt, dict) and "model_state_dict" in ckpt:
        state = ckpt["model_state_dict"]
        info = {
            "architecture": ckpt.get("architecture", "dual_gnn"),
            "hidden_dim": ckpt.get("hidden_dim", 128),
            "layers": ckpt.get("layers", 4),
            "dropout": ckpt.get("dropout", 0.1),
            "row_feat_dim": ckpt.get("row_feat_dim"),
        }
        cfg = ckpt.get("config")
        if isinstance(cfg, dict):
            for k in info:
                info[k] = cfg.get(k, info[k])
    else:
        raise ValueError("bare state dicts are legacy DualGNN checkpoints; only OneGNN is on the hot path")
    if info["architecture"] != "one_gnn":
        raise ValueError(f"architecture '{info['architecture']}' is not on the warm-start hot path (OneGNN only)")
    in_dim = info.get("row_feat_dim") or ROW_FEATURE_DIM
    if in_dim != ROW_FEATURE_DIM:
        raise ValueError(f"Checkpoint expects {in_dim} row features but the pipeline now uses {ROW_FEATURE_DIM}.")
    model = OneGNN(in_dim=in_dim, hidden=info["hidden_dim"], layers=info["layers"], dropout=info["dropout"])
    model.load_state_dict(state)
    model.to(device).eval()
    return model, info


class WarmStartPipeline:
    """Batched, device-resident warm-start solve."""

    def __init__(self, model: OneGNN, device="cuda:0", threads_hint: int = 0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("WarmStartPipeline runs on the MI355X only")
        self.lib = _hip.require_device()
        self.model = model.to(self.device).eval()
        self.threads_hint = int(threads_hint)
        self._ws = {}

    def _workspace(self, B, n, cold=False):
        # (cold solves carry the candidate lists of the row reduction: a larger block, cached separately)
        key = (B, n, cold)
        if key not in self._ws:
            if len(self._ws) >= 4:  # a handful of shapes at most: drop the oldest
                # kernels of an earlier call may still be running on it (any stream), and a captured
                # graph may hold its address: wait for the device before the block can be handed out again
                torch.cuda.synchronize(self.device)
                self._ws.pop(next(iter(self._ws)))
            nbytes = (self.lib.lapwarm_lapjv_workspace_bytes if cold else self.lib.lapwarm_seeded_workspace_bytes)(B, n)
            self._ws[key] = (torch.empty((nbytes,), dtype=torch.uint8, device=self.device), nbytes)
        ws = self._ws[key]
        # the caching allocator must not recycle the block while the stream that uses it now still runs
        ws[0].record_stream(torch.cuda.current_stream(self.device))
        return ws

    @torch.inference_mode()
    def predict_batch(self, C: torch.Tensor):
        """C (B,n,n) f64 CUDA -> u (B,n) f32 (the model output), v (B,n) f64 (fp64 min-trick on
        the float64-widened u, as the reference's CPU branch does)."""
        feat, topk = row_features_device(C)
        mask = torch.ones(feat.shape[:2], dtype=torch.bool, device=C.device)
        u = self.model(feat, mask=mask, topk_values=topk)["u"]
        v = min_trick_device(C, u)
        return u, v

    def seeded_batch(self, C: torch.Tensor, u: torch.Tensor, v: torch.Tensor, eps: float = 1e-12,
                     want_stats: bool = True):
        """Batched lapjv_seeded: x, y (B,n) int64, ret (B,) int32, stats (B,32) int64."""
        C = C.contiguous()
        B, n, _ = C.shape
        u = u.to(torch.float64).contiguous()
        v = v.to(torch.float64).contiguous()
        x = torch.empty((B, n), dtype=torch.int64, device=C.device)
        y = torch.empty((B, n), dtype=torch.int64, device=C.device)
        ret = torch.empty((B,), dtype=torch.int32, device=C.device)
        stats = torch.zeros((B, 32), dtype=torch.int64, device=C.device) if want_stats else None
        ws, nbytes = self._workspace(B, n)
        stream = torch.cuda.current_stream(C.device).cuda_stream
        rc = self.lib.lapwarm_seeded_batched(
            C.data_ptr(), B, n, u.data_ptr(), v.data_ptr(), float(eps), x.data_ptr(), y.data_ptr(),
            ret.data_ptr(), stats.data_ptr() if want_stats else None, ws.data_ptr(), nbytes,
            self.threads_hint, ct.c_void_p(stream))
        if _hip.check(rc, "seeded_batch") != 0:
            raise RuntimeError(f"lapwarm_seeded_batched failed (code {rc}): {_hip.last_error()}")
        return x, y, ret, stats

    def lapjv_batch(self, C: torch.Tensor, want_stats: bool = True):
        """Batched cold lapjv: x, y (B,n) int32, ret (B,), stats."""
        C = C.contiguous()
        B, n, _ = C.shape
        x = torch.empty((B, n), dtype=torch.int32, device=C.device)
        y = torch.empty((B, n), dtype=torch.int32, device=C.device)
        ret = torch.empty((B,), dtype=torch.int32, device=C.device)
        stats = torch.zeros((B, 32), dtype=torch.int64, device=C.device) if want_stats else None
        ws, nbytes = self._workspace(B, n, cold=True)
        stream = torch.cuda.current_stream(C.device).cuda_stream
        rc = self.lib.lapwarm_lapjv_batched(C.data_ptr(), B, n, x.data_ptr(), y.data_ptr(), ret.data_ptr(),
                                            stats.data_ptr() if want_stats else None, ws.data_ptr(), nbytes,
                                            self.threads_hint, ct.c_void_p(stream))
        if _hip.check(rc, "lapjv_batch") != 0:
            raise RuntimeError(f"lapwarm_lapjv_batched failed (code {rc}): {_hip.last_error()}")
        return x, y, ret, stats

    def optimal_duals_batch(self, C: torch.Tensor):
        """Cold JV + the optimal duals it ends with: x (B,n) int32, u, v (B,n) fp64, ret.
        u_i = C[i, x_i] - v[x_i]; (u, v) is feasible and tight on the optimal assignment."""
        C = C.contiguous()
        B, n, _ = C.shape
        x = torch.empty((B, n), dtype=torch.int32, device=C.device)
        y = torch.empty((B, n), dtype=torch.int32, device=C.device)
        u = torch.empty((B, n), dtype=torch.float64, device=C.device)
        v = torch.empty((B, n), dtype=torch.float64, device=C.device)
        ret = torch.empty((B,), dtype=torch.int32, device=C.device)
        ws, nbytes = self._workspace(B, n, cold=True)
        stream = torch.cuda.current_stream(C.device).cuda_stream
        rc = self.lib.lapwarm_lapjv_duals_batched(C.data_ptr(), B, n, x.data_ptr(), y.data_ptr(), u.data_ptr(),
                                                  v.data_ptr(), ret.data_ptr(), None, ws.data_ptr(), nbytes,
                                                  self.threads_hint, ct.c_void_p(stream))
        if _hip.check(rc, "optimal_duals_batch") != 0:
            raise RuntimeError(f"lapwarm_lapjv_duals_batched failed (code {rc}): {_hip.last_error()}")
        return x, u, v, ret

    @torch.inference_mode()
    def solve_batch(self, C: torch.Tensor, eps: float = 1e-12, want_stats: bool = True) -> dict:
        """The whole hot path for a resident batch."""
        u, v = self.predict_batch(C)
        x, y, ret, stats = self.seeded_batch(C, u, v, eps, want_stats)
        return {"x": x, "y": y, "ret": ret, "stats": stats, "u": u, "v": v}

    # ---- two-stream software pipeline: the dense sweeps + OneGNN of batch k+1 run beside the
    # per-instance solver of batch k.  The solver occupies one CU per instance (32 of the 256 at
    # K3), the sweeps want the rest; results are identical to solve_batch().
    def _streams(self):
        if not hasattr(self, "_s_pred"):
            self._s_pred = torch.cuda.Stream(self.device)
            self._s_solve = torch.cuda.Stream(self.device)
            self._pending = None
        return self._s_pred, self._s_solve

    @torch.inference_mode()
    def pipeline_submit(self, C: torch.Tensor, predict=None):
        """Enqueue features + OneGNN + min-trick for `C` on the prediction stream.  `predict(C) -> (u, v)`
        replaces the model-based prediction (K2: given duals, features + min-trick only)."""
        s_pred, _ = self._streams()
        s_pred.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s_pred):
            u, v = (predict or self.predict_batch)(C)
            ev = torch.cuda.Event()
            ev.record(s_pred)
        self._pending = (C, u, v, ev)
        self._predict = predict

    @torch.inference_mode()
    def pipeline_step(self, C_next: Optional[torch.Tensor] = None, eps: float = 1e-12,
                      want_stats: bool = True) -> dict:
        """Solve the batch submitted last (its duals are ready or being computed on the prediction
        stream), and submit `C_next` so that its dense stages overlap this solve.  The returned
        tensors are produced on the solver stream: synchronise (or wait on out["done"]) before
        reading them."""
        if self._pending is None:
            raise RuntimeError("pipeline_submit() first")
        _, s_solve = self._streams()
        C, u, v, ev = self._pending
        self._pending = None
        s_solve.wait_event(ev)
        with torch.cuda.stream(s_solve):
            x, y, ret, stats = self.seeded_batch(C, u, v, eps, want_stats)
            done = torch.cuda.Event()
            done.record(s_solve)
        for t in (C, u, v):
            t.record_stream(s_solve)
        # the results were allocated on the solver stream's pool and are consumed on the caller's
        for t in (x, y, ret) + ((stats,) if stats is not None else ()):
            t.record_stream(torch.cuda.current_stream(self.device))
        if C_next is not None:
            self.pipeline_submit(C_next, getattr(self, "_predict", None))
        return {"x": x, "y": y, "ret": ret, "stats": stats, "u": u, "v": v, "done": done}

    def pipeline_drain(self):
        s_pred, s_solve = self._streams()
        s_pred.synchronize()
        s_solve.synchronize()
        self._pending = None


class GNNPredictor:
    """`GNNPredictor(model_path).predict(C) -> (u, v)`, float64 NumPy, as the reference's
    harness expects (scripts/gnn_benchmark.py:213-289).  A ready OneGNN may be passed instead of
    a checkpoint path (`GNNPredictor(model=...)`)."""

    def __init__(self, model_path: Optional[str] = None, device: Optional[str] = None,
                 model: Optional[OneGNN] = None):
        self.model_path = model_path
        self.device = device or "cuda:0"
        if "cuda" not in str(self.device):
            raise RuntimeError("this GNNPredictor runs on the MI355X only; the CPU forward lives in the "
                               "reference / the test oracle")
        self.use_cuda = True
        self.row_only = True
        self.model_info = {}
        if model is None:
            if model_path is None:
                raise ValueError("model_path or model required")
            model, self.model_info = load_checkpoint(Path(model_path), self.device)
        self.model = model.to(self.device).eval()
        self._pipe = WarmStartPipeline(self.model, self.device)

    def predict(self, C: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        C = np.asarray(C, dtype=np.float64)
        Cd = torch.from_numpy(np.ascontiguousarray(C)).to(self.device).unsqueeze(0)
        u, v = self._pipe.predict_batch(Cd)
        torch.cuda.synchronize()
        return u[0].cpu().numpy().astype(np.float64), v[0].cpu().numpy().astype(np.float64)
