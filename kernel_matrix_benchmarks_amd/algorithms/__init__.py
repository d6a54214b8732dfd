"""Plugins in the kernel-matrix-benchmarks class API (base.py:7-167)."""
