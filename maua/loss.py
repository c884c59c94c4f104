"""Drop-in name for maua/loss.py:22-25: re-exports spherical_dist_loss (the one loss on the text-prompt guidance path)."""
from maua_amd.grad import spherical_dist_loss  # noqa: F401
