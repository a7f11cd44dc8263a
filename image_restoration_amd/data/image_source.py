"""What the image datasets share: a lazily opened byte source, decoding to the reference's in-memory convention, tensor conversion
and per-channel normalisation.

The reference repeats these steps inside each dataset class (basicsr/data/paired_image_dataset.py:67-98,
single_image_dataset.py:43-65); here they live in one mix-in.  Convention kept: decoded images are HWC **BGR** float32 in [0, 1]
(``imfrombytes(float32=True)``), tensors are CHW **RGB** float32 (``img2tensor(bgr2rgb=True)``), ``mean`` / ``std`` options normalise
in place like torchvision's ``normalize``."""
import torch

from ..utils.img_util import imfrombytes, img2tensor
from .file_client import FileClient


class ImageSource:

    def _init_source(self, opt, db_paths, client_keys):
        self.opt = opt
        self._client = None
        self._backend = dict(opt['io_backend'])
        if self._backend['type'] == 'lmdb':
            self._backend.update(db_paths=list(db_paths), client_keys=list(client_keys))
        self._mean, self._std = opt.get('mean'), opt.get('std')

    @property
    def io_backend_opt(self):
        return self._backend

    def _bytes(self, path, key):
        if self._client is None:       # opened on first use, i.e. inside the DataLoader worker that owns it
            settings = dict(self._backend)
            self._client = FileClient(settings.pop('type'), **settings)
        return self._client.get(path, key)

    def decode(self, path, key, as_float=True):
        """HWC BGR; float32 in [0, 1] by default, the raw uint8 pixels for the device-side pipeline."""
        return imfrombytes(self._bytes(path, key), float32=as_float)

    def to_tensors(self, *images):
        tensors = img2tensor(list(images), bgr2rgb=True, float32=True)
        if self._mean is not None or self._std is not None:
            channels = tensors[0].size(0)
            shift = torch.tensor(self._mean if self._mean is not None else [0.] * channels, dtype=torch.float32).view(-1, 1, 1)
            spread = torch.tensor(self._std if self._std is not None else [1.] * channels, dtype=torch.float32).view(-1, 1, 1)
            for t in tensors:
                t.sub_(shift).div_(spread)
        return tensors
