"""Plain PyTorch fp32 CPU restatement of OneGNN's inference forward -- TEST INFRASTRUCTURE.

Checker for the device forward (tolerance 1e-5 on u, BASELINE.json north_star);
never imported by the product package.  Pinned by tests/golden/onegnn_*.npz
(generated from the reference's own module by tests/golden/make_golden.py).

Restates gnn/one_gnn.py:89-160 (eval mode: every Dropout is the identity) as a
function of a state dict, and scripts/gnn_benchmark.py:226-289 (CPU branch of
GNNPredictor.predict) as `predict`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import features_np

TOPK = 16  # gnn/one_gnn.py:56 (never restored from checkpoints: SURVEY.md section 5)


def init_state_dict(hidden: int = 64, layers: int = 2, in_dim: int = 21, seed: int = 0) -> dict:
    """Deterministic random weights with the reference's state-dict keys and shapes
    (SURVEY.md Appendix C).  Not the reference's initialiser -- the golden fixtures
    carry the reference-initialised tensors; this is for synthetic benches."""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        bound = 1.0 / np.sqrt(i)
        w = (torch.rand((o, i), generator=g) * 2 - 1) * bound
        b = (torch.rand((o,), generator=g) * 2 - 1) * bound
        return w, b

    sd = {}
    sd["input_proj.0.weight"], sd["input_proj.0.bias"] = lin(hidden, in_dim)
    sd["input_proj.2.weight"], sd["input_proj.2.bias"] = torch.ones(hidden), torch.zeros(hidden)
    for l in range(layers):
        sd[f"blocks.{l}.fc1.weight"], sd[f"blocks.{l}.fc1.bias"] = lin(hidden, hidden)
        sd[f"blocks.{l}.fc2.weight"], sd[f"blocks.{l}.fc2.bias"] = lin(hidden, hidden)
        sd[f"blocks.{l}.norm.weight"], sd[f"blocks.{l}.norm.bias"] = torch.ones(hidden), torch.zeros(hidden)
    head = max(hidden // 2, 1)
    sd["pre_out.weight"], sd["pre_out.bias"] = lin(1, hidden)
    sd["row_out.0.weight"], sd["row_out.0.bias"] = lin(head, hidden)
    sd["row_out.3.weight"], sd["row_out.3.bias"] = lin(1, head)
    sd["edge_mlp.0.weight"], sd["edge_mlp.0.bias"] = lin(hidden, 1)
    sd["edge_mlp.2.weight"], sd["edge_mlp.2.bias"] = lin(hidden, hidden)
    sd["message_norm.weight"], sd["message_norm.bias"] = torch.ones(hidden), torch.zeros(hidden)
    return sd


def n_layers(sd: dict) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


@torch.no_grad()
def forward(sd: dict, row_feat: torch.Tensor, cost: torch.Tensor | None = None,
            mask: torch.Tensor | None = None, topk: int = TOPK) -> torch.Tensor:
    """row_feat (B,n,F) f32, cost (B,n,n) f32 or None, mask (B,n) bool or None -> u (B,n) f32."""
    if row_feat.ndim == 2:
        row_feat = row_feat.unsqueeze(0)
    H = sd["input_proj.0.weight"].shape[0]
    h = F.linear(row_feat, sd["input_proj.0.weight"], sd["input_proj.0.bias"])
    h = F.layer_norm(F.gelu(h), (H,), sd["input_proj.2.weight"], sd["input_proj.2.bias"], 1e-5)
    for l in range(n_layers(sd)):
        t = F.gelu(F.linear(h, sd[f"blocks.{l}.fc1.weight"], sd[f"blocks.{l}.fc1.bias"]))
        t = F.linear(t, sd[f"blocks.{l}.fc2.weight"], sd[f"blocks.{l}.fc2.bias"])
        h = F.layer_norm(h + t, (H,), sd[f"blocks.{l}.norm.weight"], sd[f"blocks.{l}.norm.bias"], 1e-5)
    u_pre = F.linear(h, sd["pre_out.weight"], sd["pre_out.bias"]).squeeze(-1)
    if cost is not None and h.shape[1] > 0:
        k = min(topk, cost.size(-1))
        if k > 0:
            red = cost - u_pre.unsqueeze(-1)
            if mask is not None:
                red = red.masked_fill(~mask.unsqueeze(-1), float("inf"))
            vals, _ = torch.topk(red, k=k, dim=-1, largest=False)
            ok = torch.isfinite(vals)
            w = torch.softmax(torch.where(ok, -vals, torch.full_like(vals, -float("inf"))), dim=-1)
            w = torch.where(ok, w, torch.zeros_like(w))
            e_in = torch.where(ok, vals, torch.zeros_like(vals)).unsqueeze(-1)
            e = F.gelu(F.linear(e_in, sd["edge_mlp.0.weight"], sd["edge_mlp.0.bias"]))
            e = F.linear(e, sd["edge_mlp.2.weight"], sd["edge_mlp.2.bias"])
            msg = (w.unsqueeze(-1) * e).sum(dim=-2)
            if mask is not None:
                msg = msg * mask.unsqueeze(-1)
            h = h + F.layer_norm(msg, (H,), sd["message_norm.weight"], sd["message_norm.bias"], 1e-5)
    t = F.gelu(F.linear(h, sd["row_out.0.weight"], sd["row_out.0.bias"]))
    u = F.linear(t, sd["row_out.3.weight"], sd["row_out.3.bias"]).squeeze(-1)
    u = u - u.mean(dim=-1, keepdim=True)
    if mask is not None:
        if mask.ndim == 1:
            mask = mask.unsqueeze(0)
        u = u.masked_fill(~mask, 0.0)
    return u


@torch.no_grad()
def predict(sd: dict, C: np.ndarray):
    """CPU branch of GNNPredictor.predict: (u f64[n], v f64[n])."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    feat = features_np.compute_row_features(C)
    row = torch.from_numpy(feat).float().unsqueeze(0)
    cost = torch.from_numpy(C).float().unsqueeze(0)
    mask = torch.ones((1, n), dtype=torch.bool)
    u32 = forward(sd, row, cost, mask).squeeze(0)[:n].numpy()
    v = np.min(C - u32[:, None], axis=0)
    return u32.astype(np.float64), v.astype(np.float64)
