"""Mirror of the reference's ``montage_gan/torch_utils`` for the hot path (ops + the bits of misc on the step path)."""
