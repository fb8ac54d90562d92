"""GPU side of the reference's srcs/transform filters that sit on the augmentation hot path."""
from .filters import (TransformConfig, analyze_color_regions, apply_blur_filter,  # noqa: F401
                      create_inclusive_mask, hsv_density_curves, hue_range_counts, leaf_hsv_histograms)
