"""Data side of the path: the rank-aware sampler of the reference and a synthetic paired dataset for
benchmarks / smoke training (the PNG/LMDB pipeline is SURVEY.md §8 row f3, not built yet)."""
from .data_sampler import EnlargedSampler  # noqa: F401
from .synthetic_dataset import SyntheticPairedDataset  # noqa: F401
