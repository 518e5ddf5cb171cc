"""nd_amd/_device.py -- host <-> device plumbing (numpy arrays in, torch ROCm tensors on the GPU)."""
import warnings

import numpy as np
import torch


def is_tensor(a):
    return isinstance(a, torch.Tensor)


def device_of(*arrays, device=None):
    """The device to compute on: an explicit one, else that of the first CUDA tensor among the
    inputs, else the current ROCm device.  There is no CPU path."""
    if device is not None:
        return torch.device(device)
    for a in arrays:
        if is_tensor(a) and a.is_cuda:
            return a.device
    if not torch.cuda.is_available():
        raise RuntimeError('nd_amd needs a ROCm GPU: the compute path is HIP only '
                           '(torch.cuda.is_available() is False)')
    return torch.device('cuda', torch.cuda.current_device())


def to_device(a, device):
    if is_tensor(a):
        return a if a.device == device else a.to(device)
    a = np.asarray(a)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    with warnings.catch_warnings():
        # read-only sources (memory maps, xarray views) are only read
        warnings.filterwarnings('ignore', message='The given NumPy array is not writable')
        return torch.from_numpy(a).to(device)


def write_back(result, output):
    """Store a device tensor into `output` in place (numpy array view or torch tensor)."""
    if is_tensor(output):
        output.copy_(result)
    elif (result.is_cuda and output.flags.c_contiguous and output.flags.writeable
          and result.numel() * result.element_size() >= (32 << 20)
          and output.dtype == np_dtype(result)):
        # large host target: download into page-locked memory, then a multi-threaded host copy
        host = torch.empty(result.shape, dtype=result.dtype, pin_memory=True)
        host.copy_(result)
        torch.from_numpy(output).copy_(host.reshape(output.shape))
    else:
        output[...] = to_host(result)
    return output


def to_host(t):
    """Device tensor -> numpy array.  The download goes into page-locked memory (several times
    faster than into pageable memory) and the returned array is a view of that buffer."""
    if not t.is_cuda:
        return t.numpy()
    if t.numel() * t.element_size() < (32 << 20):
        return t.cpu().numpy()              # small: a page-locked allocation costs more than it saves
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t)
    return host.numpy()


_TORCH2NP = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16,
             torch.complex64: np.complex64, torch.complex128: np.complex128,
             torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8,
             torch.int16: np.int16, torch.bool: np.bool_}


def np_dtype(a):
    if is_tensor(a):
        return np.dtype(_TORCH2NP[a.dtype])
    return np.asarray(a).dtype
