"""Minimal stand-in for `torch_geometric` (absent from this image); test infrastructure only."""
