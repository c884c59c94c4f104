"""Drop-in name for maua/GAN/wrappers/inference/ops.py:23-256 (the operator layer = the C-ABI boundary): re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.ops import (activate, bias_act, conv2d_resample, get_activation_defaults, modulated_conv2d, normalize_2nd_moment, setup_filter,  # noqa: F401
                          upfirdn2d, upsample2d)
