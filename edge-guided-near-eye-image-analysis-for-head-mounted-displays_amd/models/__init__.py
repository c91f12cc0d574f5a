"""Model variants of the hot path (RITnet_v2.DenseNet2D, RITnet_concat.DenseNet2D)."""
