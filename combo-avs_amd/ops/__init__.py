"""Autograd wrappers around the C-ABI kernels of libcombo_avs_hip.so (one module per kernel family)."""
