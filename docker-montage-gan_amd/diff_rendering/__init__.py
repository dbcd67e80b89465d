"""Learned differentiable renderer (reference diff_rendering/networks.py)."""
