"""OneGNN: per-row residual MLP + top-k refinement that predicts the row duals u.

Same constructor, parameter names / shapes (checkpoints load unchanged), call signature and
output dict as the reference module (gnn/one_gnn.py:18-160).  The forward is restructured for
the MI355X:

  * the 16 smallest reduced costs per row are not re-extracted from a (B, n, n) float32 copy of
    the cost matrix: topk(cost - u_pre) == topk(cost) - u_pre (x -> x - c is monotone in
    float32), and topk(cost) is a by-product of the fp64 row-feature sweep.  Callers that have
    it pass `topk_values=`; callers that only have `cost=` get a torch.topk on the device;
  * in eval mode the edge MLP's second (linear) layer is applied after the softmax-weighted sum
    instead of before it -- 16x fewer flops and no (B, n, 16, H) tensor in HBM.  The aggregation
    sum_k w_k GELU(w1 val_k + b1) is a hand-written HIP kernel (onegnn_refine.hip); the H x H
    GEMMs stay with PyTorch-ROCm (rocBLAS/hipBLASLt on the MFMA units);
  * training mode keeps the reference's op order (dropout placement).
"""
from __future__ import annotations

import ctypes as ct
from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F


class ResidualBlock(nn.Module):
    def __init__(self, hidden: int, dropout: float) -> None:
        super().__init__()
        self.fc1 = nn.Linear(hidden, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        self.norm = nn.LayerNorm(hidden)
        self.dropout = nn.Dropout(dropout)
        self.act = nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.dropout(self.act(self.fc1(x)))
        out = self.dropout(self.fc2(out))
        return self.norm(x + out)


class OneGNN(nn.Module):
    def __init__(self, in_dim: int, hidden: int = 64, layers: int = 2, dropout: float = 0.1,
                 topk: int = 16) -> None:
        super().__init__()
        if layers < 1:
            raise ValueError("layers must be >= 1")
        if hidden < 2:
            raise ValueError("hidden dimension must be >= 2 for head projection")
        self.input_proj = nn.Sequential(nn.Linear(in_dim, hidden), nn.GELU(), nn.LayerNorm(hidden))
        self.blocks = nn.ModuleList([ResidualBlock(hidden, dropout) for _ in range(layers)])
        head_hidden = max(hidden // 2, 1)
        self.pre_out = nn.Linear(hidden, 1)
        self.row_out = nn.Sequential(nn.Linear(hidden, head_hidden), nn.GELU(), nn.Dropout(dropout),
                                     nn.Linear(head_hidden, 1))
        self.topk = topk
        self.edge_mlp = nn.Sequential(nn.Linear(1, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self.message_norm = nn.LayerNorm(hidden)
        self.message_dropout = nn.Dropout(dropout)

    # ------------------------------------------------------------------ forward
    def forward(self, row_feat: torch.Tensor, *, cost: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None,
                topk_values: Optional[torch.Tensor] = None) -> dict:
        if row_feat.ndim == 2:
            row_feat = row_feat.unsqueeze(0)
        if row_feat.ndim != 3:
            raise ValueError("row_feat must have shape (batch, n, F)")
        h = self.input_proj(row_feat)
        for block in self.blocks:
            h = block(h)
        u_pre = self.pre_out(h).squeeze(-1)
        if cost is not None or topk_values is not None:
            h = h + self._sparse_refine(h, cost, u_pre, mask, topk_values)
        u = self.row_out(h).squeeze(-1)
        u = u - u.mean(dim=-1, keepdim=True)
        if mask is not None:
            if mask.ndim == 1:
                mask = mask.unsqueeze(0)
            u = u.masked_fill(~mask, 0.0)
        return {"u": u}

    def _sparse_refine(self, h, cost, u_pre, mask, topk_values):
        B, N, H = h.shape
        if N == 0:
            return torch.zeros_like(h)
        width = cost.size(-1) if cost is not None else topk_values.size(-1)
        k = min(self.topk, width)
        if k <= 0:
            return torch.zeros_like(h)
        mask_rows = mask.unsqueeze(-1) if mask is not None else None
        if topk_values is not None:
            # ascending smallest costs per row (float32, +inf padded): subtracting u_pre afterwards
            # gives exactly topk(cost - u_pre)
            base = topk_values[..., :k]
            if topk_values.ndim == 2:
                base = base.unsqueeze(0)
        else:
            base, _ = torch.topk(cost, k=k, dim=-1, largest=False)
        if self.training or not h.is_cuda or k != 16:
            return self._refine_reference_order(h, base, u_pre, mask_rows)
        return self._refine_fused(h, base.contiguous(), u_pre, mask_rows)

    def _refine_reference_order(self, h, base, u_pre, mask_rows):
        values = base - u_pre.unsqueeze(-1)
        if mask_rows is not None:
            values = values.masked_fill(~mask_rows, float("inf"))
        valid = torch.isfinite(values)
        neg = torch.where(valid, -values, torch.full_like(values, -float("inf")))
        w = torch.softmax(neg, dim=-1)
        w = torch.where(valid, w, torch.zeros_like(w))
        e_in = torch.where(valid, values, torch.zeros_like(values)).unsqueeze(-1)
        e = self.edge_mlp(e_in)
        msg = (w.unsqueeze(-1) * e).sum(dim=-2)
        if mask_rows is not None:
            msg = msg * mask_rows
        return self.message_norm(self.message_dropout(msg))

    def _refine_fused(self, h, base, u_pre, mask_rows):
        from lap import _hip
        B, N, H = h.shape
        rows = B * N
        up = u_pre.contiguous().view(rows)
        if mask_rows is not None:
            # masked rows: every value becomes +inf -> zero weights -> zero message
            base = base.masked_fill(~mask_rows, float("inf")).contiguous()
        agg = torch.empty((rows, H), dtype=torch.float32, device=h.device)
        wsum = torch.empty((rows,), dtype=torch.float32, device=h.device)
        lin1, lin2 = self.edge_mlp[0], self.edge_mlp[2]
        lib = _hip.require_device()
        stream = torch.cuda.current_stream(h.device).cuda_stream
        rc = lib.lapwarm_refine_aggregate_wsum(
            base.data_ptr(), up.data_ptr(), lin1.weight.contiguous().view(-1).data_ptr(),
            lin1.bias.contiguous().data_ptr(), agg.data_ptr(), wsum.data_ptr(), rows, H,
            ct.c_void_p(stream))
        if _hip.check(rc, "refine_aggregate") != 0:
            raise RuntimeError(f"refine_aggregate failed (code {rc})")
        msg = F.linear(agg, lin2.weight) + wsum.unsqueeze(-1) * lin2.bias
        msg = msg.view(B, N, H)
        if mask_rows is not None:
            msg = msg * mask_rows
        return self.message_norm(msg)
