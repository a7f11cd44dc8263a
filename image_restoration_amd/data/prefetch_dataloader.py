"""Feeds of the training loop: how the next batch reaches the step.

Behaviours of basicsr/data/prefetch_dataloader.py, own construction:

``BackgroundIterator`` / ``PrefetchDataLoader``  (reference :7-63) a daemon thread drains an iterator into a bounded queue so that
    collation overlaps the training step.
``HostFeed``    (reference CPUPrefetcher :66-82) batches straight from the loader.
``DeviceFeed``  (reference CUDAPrefetcher :84-125) the batch after the current one is already being copied to HBM on a private copy
    stream — and, with a ``DevicePatchPipeline``, augmented there by one HIP kernel — while the current step computes.

Both feeds speak ``reset()`` / ``next()`` (``None`` at the end of an epoch), which is all train.py uses; ``CPUPrefetcher`` and
``CUDAPrefetcher`` remain as the names the option ``prefetch_mode: cpu | cuda`` selects."""
import queue
import threading

import torch
from torch.utils.data import DataLoader

_END = object()


class BackgroundIterator(threading.Thread):
    """Iterates ``source`` on a daemon thread, at most ``depth`` items ahead of the consumer."""

    def __init__(self, source, depth):
        super().__init__(daemon=True)
        self._items = queue.Queue(maxsize=max(1, int(depth)))
        self._source = source
        self._failure = None
        self.start()

    def run(self):
        try:
            for item in self._source:
                self._items.put(item)
        except BaseException as exc:  # noqa: BLE001 - re-raised in the consumer
            self._failure = exc
        self._items.put(_END)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._items.get()
        if item is _END:
            if self._failure is not None:
                raise self._failure
            raise StopIteration
        return item


PrefetchGenerator = BackgroundIterator


class PrefetchDataLoader(DataLoader):
    """A DataLoader whose iteration runs ``num_prefetch_queue`` batches ahead on a background thread."""

    def __init__(self, num_prefetch_queue, **kwargs):
        self.num_prefetch_queue = num_prefetch_queue
        super().__init__(**kwargs)

    def __iter__(self):
        return BackgroundIterator(super().__iter__(), self.num_prefetch_queue)


class HostFeed:

    def __init__(self, loader):
        self._loader = loader
        self._it = None
        self.reset()

    def reset(self):
        self._it = iter(self._loader)

    def next(self):
        return next(self._it, None)


class DeviceFeed:
    """Double buffering across streams: ``next()`` hands out batch k (making the caller's stream wait for its copy / augmentation)
    and immediately starts staging batch k+1 on the copy stream.  ``pipeline`` (optional) maps the staged batch on the copy stream,
    e.g. the uint8 -> float augmentation kernel."""

    def __init__(self, loader, opt, pipeline=None):
        self._loader = loader
        self._pipeline = pipeline
        self.device = torch.device('cpu' if opt['num_gpu'] == 0 else 'cuda')
        self._copy_stream = torch.cuda.Stream() if self.device.type == 'cuda' else None
        self._it = None
        self._staged = None
        self.reset()

    def reset(self):
        self._it = iter(self._loader)
        self._stage()

    def _stage(self):
        batch = next(self._it, None)
        if batch is not None and self._copy_stream is not None:
            with torch.cuda.stream(self._copy_stream):
                batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                if self._pipeline is not None:
                    batch = self._pipeline(batch)
        elif batch is not None and self._pipeline is not None:
            batch = self._pipeline(batch)
        self._staged = batch

    def next(self):
        ready = self._staged
        if self._copy_stream is not None:
            consumer = torch.cuda.current_stream()
            consumer.wait_stream(self._copy_stream)
            if ready is not None:
                for v in ready.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(consumer)   # allocated on the copy stream, used on the consumer's
        self._stage()
        return ready


CPUPrefetcher = HostFeed
CUDAPrefetcher = DeviceFeed
