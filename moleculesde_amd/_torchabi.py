"""The three conversions every wrapper needs: torch's current HIP stream, a tensor's device address, fp32 / int32 contiguity
(no CPU path: a CPU tensor raises)."""
import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    if not t.is_cuda:
        raise _lib.MsdeHipError("moleculesde_amd kernels need tensors on the HIP device (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _i32(t):
    if not t.is_cuda:
        raise _lib.MsdeHipError("moleculesde_amd kernels need tensors on the HIP device (no CPU fallback)")
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()
