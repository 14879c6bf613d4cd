"""Stand-in for distribution_extension.utils (absent)."""
from oracle.ref_dists import cat_distribution, stack_distribution

__all__ = ["cat_distribution", "stack_distribution"]
