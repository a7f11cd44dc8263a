"""Pre/post-processing convention of the path, restated with numpy (the reference's
``basicsr/utils/img_util.py`` needs cv2/torchvision, absent here).

``img2tensor`` (:9-35): HWC (BGR) -> CHW (RGB) float tensor; the caller scales to [0,1]
(inference.py:68, imfrombytes(float32=True) :128-132).  ``tensor2img`` (:38-94): squeeze, float,
clamp to ``min_max``, normalise, CHW -> HWC, RGB -> BGR, ``(x*255).round()`` -> uint8.
"""
import numpy as np
import torch


def img2tensor(imgs, bgr2rgb=True, float32=True):
    def _one(img):
        if img.ndim == 3 and img.shape[2] == 3 and bgr2rgb:
            img = img[:, :, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1) if img.ndim == 3 else img[None]))
        return t.float() if float32 else t

    if isinstance(imgs, list):
        return [_one(i) for i in imgs]
    return _one(imgs)


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    """Accepts a tensor [1,C,H,W], [C,H,W] or [H,W] or a list of them.  4-D batches with N>1 are
    not gridded (the reference's make_grid path needs torchvision and is unused on this path)."""
    if not (torch.is_tensor(tensor) or (isinstance(tensor, list) and all(torch.is_tensor(t) for t in tensor))):
        raise TypeError(f'tensor or list of tensors expected, got {type(tensor)}')
    single = torch.is_tensor(tensor)
    result = []
    for t in ([tensor] if single else tensor):
        t = t.squeeze(0).float().detach().cpu().clamp_(*min_max)
        t = (t - min_max[0]) / (min_max[1] - min_max[0])
        if t.dim() == 3:
            img = t.numpy().transpose(1, 2, 0)
            if img.shape[2] == 1:
                img = np.squeeze(img, axis=2)
            elif rgb2bgr:
                img = img[:, :, ::-1]
        elif t.dim() == 2:
            img = t.numpy()
        else:
            raise TypeError(f'Only support 4D (N=1), 3D and 2D tensor. But received with dimension: {t.dim()}')
        if out_type == np.uint8:
            img = (img * 255.0).round()
        result.append(np.ascontiguousarray(img).astype(out_type))
    return result[0] if len(result) == 1 else result  # a one-element list unwraps, like the reference (:92-94)


def imfrombytes(content, flag='color', float32=False):
    """Decode image bytes to an HWC **BGR** ndarray like cv2.imdecode (img_util.py:114-135): 'color' -> 3 channels,
    'grayscale' -> 2-D, 'unchanged' -> as stored (alpha kept, channel order BGR[A]).  Decoded with PIL (cv2 is absent)."""
    import io
    from PIL import Image
    img = Image.open(io.BytesIO(content))
    if flag == 'color':
        arr = np.asarray(img.convert('RGB'))[:, :, ::-1]
    elif flag == 'grayscale':
        arr = np.asarray(img.convert('L'))
    elif flag == 'unchanged':
        arr = np.asarray(img)
        if arr.ndim == 3 and arr.shape[2] >= 3:
            arr = np.concatenate([arr[:, :, 2::-1], arr[:, :, 3:]], axis=2)
    else:
        raise KeyError(flag)
    arr = np.ascontiguousarray(arr)
    return arr.astype(np.float32) / 255. if float32 else arr
