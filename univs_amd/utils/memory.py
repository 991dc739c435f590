"""Out-of-memory retry around the clip loops' full-resolution resizes (the reference wraps exactly these `F.interpolate` calls in
detectron2's `retry_if_cuda_oom`: mask2former_video/utils/memory.py:27-80, used at inference_video_entity.py:933, :978, :1104 --
at BASELINE config 5 one `pred_masks` tensor is 1.04 GB and the 40-frame anchor peaks at 17 GB).

Same ladder as the reference's wrapper -- the call as given; again after the caching allocator has returned its free blocks; then on
the host in float32 -- with two differences a 288-GB device suggests: an allocation failure is recognised by its exception TYPE
(`torch.OutOfMemoryError`; the message text as a fallback for older runtimes, HIP wording included), and a result that was
computed on the host is copied BACK to the inputs' device when it fits (`back=True`, the default), so callers never see a device
change they did not ask for.  `func` must be stateless (it may run up to three times)."""
import functools
import logging

import torch

__all__ = ["retry_if_oom", "OOM_EVENTS"]

OOM_EVENTS = {"empty_cache": 0, "host": 0}      # how often each rung was taken (tests, logs)
_OOM_TYPES = tuple(t for t in (getattr(torch, "OutOfMemoryError", None), getattr(torch.cuda, "OutOfMemoryError", None)) if t is not None)


def _is_oom(e):
    if _OOM_TYPES and isinstance(e, _OOM_TYPES):
        return True
    msg = str(e)
    return isinstance(e, RuntimeError) and ("out of memory" in msg.lower())


def _device_of(args, kwargs):
    for x in list(args) + list(kwargs.values()):
        if isinstance(x, torch.Tensor) and x.device.type != "cpu":
            return x.device
    return None


def _to_host(x):
    if isinstance(x, torch.Tensor) and x.device.type != "cpu":
        x = x.cpu()
        return x.float() if x.dtype in (torch.float16, torch.bfloat16) else x
    return x


def retry_if_oom(func, back=True):
    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        try:
            return func(*args, **kwargs)
        except Exception as e:        # noqa: BLE001 -- re-raised unless it is an allocation failure
            if not _is_oom(e):
                raise
        dev = _device_of(args, kwargs)
        if dev is not None and dev.type == "cuda":
            torch.cuda.empty_cache()
        OOM_EVENTS["empty_cache"] += 1
        try:
            return func(*args, **kwargs)
        except Exception as e:        # noqa: BLE001
            if not _is_oom(e):
                raise
        OOM_EVENTS["host"] += 1
        logging.getLogger("univs_amd").info("%s: out of device memory twice -- running on the host in float32", getattr(func, "__name__", func))
        with torch.autocast("cuda", enabled=False):
            out = func(*[_to_host(a) for a in args], **{k: _to_host(v) for k, v in kwargs.items()})
        if back and dev is not None and isinstance(out, torch.Tensor):
            try:
                return out.to(dev)
            except Exception as e:    # noqa: BLE001
                if not _is_oom(e):
                    raise
        return out
    return wrapped
