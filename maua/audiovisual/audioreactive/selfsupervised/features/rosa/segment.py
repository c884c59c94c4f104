"""Drop-in name for .../features/rosa/segment.py:7-267: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.segment import (differentiable_k_means, init_plus_plus, laplacian_segmentation,  # noqa: F401
                              laplacian_segmentation_rosa, recurrence_matrix, timelag_median_filter)
