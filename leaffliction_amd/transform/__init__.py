"""GPU side of the reference's srcs/transform filters that sit on the augmentation hot path."""
from .filters import TransformConfig, analyze_color_regions, apply_blur_filter, hue_range_counts  # noqa: F401
