"""CPU: the out-of-memory ladder around the clip loops' resizes (univs_amd/utils/memory.py; reference:
mask2former_video/utils/memory.py:27-80): as given -> after the allocator's cache is emptied -> on the host; other errors pass."""
import pytest
import torch

from univs_amd.utils import memory


def _oom():
    cls = getattr(torch, "OutOfMemoryError", None)
    return cls("HIP out of memory. Tried to allocate 1.04 GiB") if cls is not None else RuntimeError("HIP out of memory. Tried to allocate")


def test_ladder_and_pass_through():
    calls = []

    def f(x, scale=1.0):
        calls.append(x.device.type)
        if len(calls) <= fails[0]:
            raise _oom()
        return x * scale
    x = torch.arange(4.0)
    for n_fail, events in ((0, (0, 0)), (1, (1, 0)), (2, (1, 1))):
        calls.clear()
        fails = [n_fail]
        memory.OOM_EVENTS.update(empty_cache=0, host=0)
        y = memory.retry_if_oom(f)(x, scale=2.0)
        assert torch.equal(y, x * 2) and len(calls) == n_fail + 1
        assert (memory.OOM_EVENTS["empty_cache"], memory.OOM_EVENTS["host"]) == events
    fails = [3]
    calls.clear()
    with pytest.raises(Exception) as ei:           # the host attempt fails too: the caller sees the allocation failure
        memory.retry_if_oom(f)(x)
    assert "out of memory" in str(ei.value)

    def g(x):
        raise RuntimeError("shape mismatch")
    with pytest.raises(RuntimeError, match="shape mismatch"):
        memory.retry_if_oom(g)(x)


def test_the_clip_loops_resize_through_it(monkeypatch):
    import torch.nn.functional as F
    from univs_amd.inference import video_entity, video_vos
    seen = []
    real = F.interpolate

    def flaky(*a, **k):
        seen.append(1)
        if len(seen) == 1:
            raise _oom()
        return real(*a, **k)
    monkeypatch.setattr(F, "interpolate", flaky)
    m = torch.rand(2, 3, 4, 6)
    out = video_entity._resize(m, (8, 12))
    assert out.shape == (2, 3, 8, 12) and len(seen) == 2
    assert torch.equal(out, real(m, (8, 12), mode="bilinear", align_corners=False))
    assert video_vos._resize(m, (8, 12)).shape == (2, 3, 8, 12)
