"""Layout glue between the local GANs, the STN and the global GAN (reference custom_utils/)."""
