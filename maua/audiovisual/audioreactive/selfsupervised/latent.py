"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/latent.py:16-92: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.latent import (latent_patch, merge, multi_weighted, select_modulo, single_weighted, slerp,  # noqa: F401
                             slerp_loops, spline_loop_latents, spline_loops, tempo_loops)
