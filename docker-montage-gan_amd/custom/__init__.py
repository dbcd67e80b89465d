"""MontageGAN-specific callers (aio networks, global loss, STN placement, compositing)."""
