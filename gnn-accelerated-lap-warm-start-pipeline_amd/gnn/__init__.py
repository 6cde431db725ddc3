"""`gnn` -- the OneGNN warm-start model and its row features on the MI355X.

Import surface of the hot path (reference: gnn/__init__.py:3-23): OneGNN, compute_row_features,
compute_row_features_torch, ROW_FEATURE_DIM.  DualGNN / compute_features (the O(n^2)-edge model)
are outside the hot path and not provided."""
from .features import compute_row_features, compute_row_features_torch, positional_encodings, ROW_FEATURE_DIM
from .one_gnn import OneGNN, ResidualBlock
from .pipeline import GNNPredictor, WarmStartPipeline, load_checkpoint

__all__ = ["OneGNN", "ResidualBlock", "compute_row_features", "compute_row_features_torch",
           "positional_encodings", "ROW_FEATURE_DIM", "GNNPredictor", "WarmStartPipeline", "load_checkpoint"]
