"""Drop-in name for maua/audiovisual/audioreactive/latent.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.latent import (copeerp, eerp, multi_weighted, select_modulo, single_weighted, slerp, slerp_loops,  # noqa: F401
                             spline_loop_latents, spline_loops, tempo_loops)
