"""Data side of the path: the rank-aware sampler of the reference, the paired and LQ-only image datasets with the crop / flip
augmentation and prefetchers, and a synthetic paired dataset for benchmarks / smoke training."""
from .data_sampler import EnlargedSampler  # noqa: F401
from .paired_image_dataset import PairedImageDataset  # noqa: F401
from .device_pipeline import DevicePatchPipeline  # noqa: F401
from .prefetch_dataloader import (BackgroundIterator, CPUPrefetcher, CUDAPrefetcher, DeviceFeed, HostFeed,  # noqa: F401
                                  PrefetchDataLoader)
from .single_image_dataset import SingleImageDataset  # noqa: F401
from .synthetic_dataset import SyntheticPairedDataset  # noqa: F401
