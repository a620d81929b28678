"""Decode loop around the HIP operators: model layer sequence, KV-cache engine, input
builder, scheduler glue and executors (the callers of the hot path)."""
