"""The plugin API: upfirdn2d, bias_act, conv2d_resample, conv2d_gradfix, grid_sample_gradfix, fma
(reference montage_gan/torch_utils/ops/) plus the ops the hot path needed added:
alpha_composite, translate_sample, filtered_lrelu, modconv_epilogue."""
