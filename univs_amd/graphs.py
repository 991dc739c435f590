"""hipGraph replay of the shape-static parts of a clip (backbone, pixel decoder).

A config-2 clip is ~750 kernel launches; the host needs ~18 ms to enqueue them against ~25 ms of GPU time, and a prompted
clip (1 800 launches) is host-bound outright (profiles/r03_prompted_clip_breakdown_v0.txt).  The backbone and the pixel
decoder are pure functions of their input tensors with no host synchronisation (the C ABI is capture-legal:
include/univs_hip.h), so their launch sequences are captured once per input signature (shapes, dtypes, device) and
replayed: one `hipGraphLaunch` instead of ~450 launches.  Opt-in (`SWITCHES.graphs` / `UNIVS_GRAPHS=1`): capture takes a
private memory pool per signature (the activations of one forward) and two eager warm-up calls.

Semantics are those of the eager call: inputs are copied into the graph's static buffers, outputs are CLONED out of them
(fresh tensors, as eager returns: a caller may keep the features of one window while computing the next -- the reference's
long-video path does), the arithmetic is the very same kernels, so results are bit-identical (tests/test_graphs_gpu.py).
"""
import collections

import torch


def _flatten(obj, out):
    """Tensors of a nested tuple / list / dict structure, in a fixed order; returns a spec to rebuild it."""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return ("t",)
    if isinstance(obj, (list, tuple)):
        return ("l" if isinstance(obj, list) else "u", [_flatten(o, out) for o in obj])
    if isinstance(obj, dict):
        return ("d", [(k, _flatten(v, out)) for k, v in obj.items()])
    return ("c", obj)            # a constant (None, int, ...): part of the signature


def _unflatten(spec, it):
    kind = spec[0]
    if kind == "t":
        return next(it)
    if kind in ("l", "u"):
        seq = [_unflatten(s, it) for s in spec[1]]
        return seq if kind == "l" else tuple(seq)
    if kind == "d":
        return {k: _unflatten(s, it) for k, s in spec[1]}
    return spec[1]


def _spec_key(spec):
    kind = spec[0]
    if kind == "t":
        return "t"
    if kind in ("l", "u"):
        return (kind, tuple(_spec_key(s) for s in spec[1]))
    if kind == "d":
        return ("d", tuple((k, _spec_key(s)) for k, s in spec[1]))
    return ("c", repr(spec[1]))


class GraphedCallable:
    """`fn(*args)` with tensors / nested containers of tensors in and out, replayed as a hipGraph per input signature."""

    def __init__(self, fn, max_graphs=4, warmup=2, module=None):
        """`module`: the nn.Module whose parameters `fn` reads.  A captured graph bakes in pointers to tensors DERIVED from the
        parameters (pre-split weights, permuted / merged projection weights, bias tables): the fingerprint of the module's
        parameters and buffers (address + version of each) is part of the graph key, so `load_state_dict`, an optimizer step or
        `.to()` re-captures instead of replaying against stale or freed memory.  Writes through `.data` bump no version
        counter: call `reset()` after those (as `ops.invalidate_presplit`)."""
        self.fn, self.max_graphs, self.warmup = fn, max_graphs, warmup
        self.module = module if module is not None else getattr(fn, "__self__", None)
        self.entries = collections.OrderedDict()
        self.replays = 0

    def reset(self):
        """Drop every captured graph (the next call captures again)."""
        self.entries.clear()

    def _fingerprint(self):
        m = self.module
        if not isinstance(m, torch.nn.Module):
            return ()
        return tuple((t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))

    def eligible(self, flat):
        return (bool(flat) and all(t.is_cuda for t in flat) and not torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing())

    def __call__(self, *args):
        flat, spec = [], None
        spec = _flatten(args, flat)
        if not self.eligible(flat):
            return self.fn(*args)
        key = (_spec_key(spec), tuple((tuple(t.shape), t.dtype, str(t.device)) for t in flat), self._fingerprint())
        e = self.entries.get(key)
        if e is None:
            e = self._capture(flat, spec)
            self.entries[key] = e
            while len(self.entries) > self.max_graphs:
                self.entries.popitem(last=False)
        else:
            self.entries.move_to_end(key)
        graph, static_in, static_out, out_spec = e
        for s, t in zip(static_in, flat):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        graph.replay()
        self.replays += 1
        return _unflatten(out_spec, iter([o.clone() for o in static_out]))

    def _capture(self, flat, spec):
        static_in = [t.clone() for t in flat]
        args = _unflatten(spec, iter(static_in))
        # eager warm-up on a side stream (allocator, library workspaces, one-time table uploads of the native operators)
        side = torch.cuda.Stream(device=flat[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn(*args)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.fn(*args)
        static_out = []
        out_spec = _flatten(out, static_out)
        return graph, static_in, static_out, out_spec
