"""Batch prefetchers of the training loop (basicsr/data/prefetch_dataloader.py:66-125): CPUPrefetcher is a plain iterator
wrapper, CUDAPrefetcher copies the next batch to the HIP device on a side stream while the current step computes (the
kernels of libsr_hip.so run on torch's current stream, so ``wait_stream`` orders them after the copy)."""
import torch


class CPUPrefetcher:

    def __init__(self, loader):
        self.ori_loader = loader
        self.loader = iter(loader)

    def next(self):
        try:
            return next(self.loader)
        except StopIteration:
            return None

    def reset(self):
        self.loader = iter(self.ori_loader)


class CUDAPrefetcher:

    def __init__(self, loader, opt):
        self.ori_loader = loader
        self.loader = iter(loader)
        self.opt = opt
        self.device = torch.device('cuda' if opt['num_gpu'] != 0 else 'cpu')
        self.stream = torch.cuda.Stream() if self.device.type == 'cuda' else None
        self.preload()

    def preload(self):
        try:
            self.batch = next(self.loader)
        except StopIteration:
            self.batch = None
            return
        if self.stream is None:
            return
        with torch.cuda.stream(self.stream):
            for k, v in self.batch.items():
                if torch.is_tensor(v):
                    self.batch[k] = v.to(device=self.device, non_blocking=True)

    def next(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        batch = self.batch
        if batch is not None and self.stream is not None:
            for v in batch.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream())  # the consumer stream now owns the memory
        self.preload()
        return batch

    def reset(self):
        self.loader = iter(self.ori_loader)
        self.preload()
