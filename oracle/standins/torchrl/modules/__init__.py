from oracle.ref_dists import MLP

__all__ = ["MLP"]
