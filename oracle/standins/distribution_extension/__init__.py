"""Stand-in for distribution_extension 1.0.7 (absent): re-exports the oracle's restatement."""
from oracle.ref_dists import Distribution, MultiOneHot, MultiOneHotFactory, kl_divergence

__all__ = ["Distribution", "MultiOneHot", "MultiOneHotFactory", "kl_divergence"]
