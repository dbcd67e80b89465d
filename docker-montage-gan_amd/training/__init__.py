"""Callers of the plugin ops on the training-step path (networks, losses, phase loop)."""
