"""Frame sharding of one clip over the GPUs of a node.

Everything on the hot path is per-frame (backbone, pixel decoder incl. every MSDeformAttn call, masked
cross-attention, ProCA, FFN, mask decode) except the spatio-temporal self-attention over the Q'*T query
tokens and three tiny means over T (class logits, grounding re-id).  So rank r owns a contiguous block of
frames, keeps its feature maps local (80 MB/frame at 720p never moves) and, once per decoder layer,
all-gathers the query states `[Q', T_loc, 256]` (~0.6 MB per rank at Q'=120, T_loc=5) over RCCL/xGMI --
latency-bound, no ring needed.  The reference has no such mode (it has no model-side collective at all on
the inference path, SURVEY.md section 2 "Parallelism"); semantics are fixed by the single-process result,
which `tests/test_distributed_cpu.py` checks with a 2-rank gloo group.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import collections

import torch
import torch.distributed as dist


class FrameShard:
    """Equal contiguous blocks of frames per rank (T_total = world * T_loc)."""

    def __init__(self, group=None, always_collective=False):
        """`always_collective`: issue the collectives even in a one-rank group (bench.py's N = 1 anchor runs the very code
        path of N ranks, RCCL calls included; by default a one-rank shard returns its inputs)."""
        assert dist.is_available() and dist.is_initialized(), "init torch.distributed first"
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.always_collective = always_collective
        # bytes this rank RECEIVED per kind of collective ("all_gather", "all_reduce", "broadcast"; "result:*" = the clip loop's
        # gathers of mask logits / embeddings): tests and DESIGN.md section 6 read them; a ClipShard adds to its parent's counter
        self.bytes = collections.Counter()
        self._subgroups = {}

    def global_rank(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def subgroup(self, ranks):
        """The process group of the group ranks `ranks` (sorted, distinct), made once per set.  `dist.new_group` is a collective over
        the WHOLE job: every rank of this shard must ask for the same sets in the same order (the clip loops do: every rank walks
        every clip)."""
        key = tuple(int(r) for r in ranks)
        g = self._subgroups.get(key)
        if g is None:
            kw = {"timeout": self.timeout} if getattr(self, "timeout", None) is not None else {}
            g = self._subgroups[key] = dist.new_group(ranks=[self.global_rank(r) for r in key], **kw)
        return g

    def broadcast_state(self, src: int, tensors, device):
        """`tensors` ({name: tensor} on the source rank `src` (group rank), anything elsewhere) -> the same dict on every rank: one
        small object broadcast (names, shapes, dtypes) and one broadcast per tensor.  CPU tensors travel through `device` when the
        backend needs device memory (RCCL)."""
        gsrc = self.global_rank(src)
        meta = [[(k, tuple(v.shape), v.dtype, v.device.type) for k, v in tensors.items()] if self.rank == src else None]
        dist.broadcast_object_list(meta, src=gsrc, group=self.group)
        via_device = dist.get_backend(self.group) != "gloo"
        out = {}
        for k, shape, dtype, dev_type in meta[0]:
            if self.rank == src:
                t = tensors[k].contiguous()
                t = t.to(device) if (via_device and dev_type == "cpu") else t
            else:
                t = torch.empty(shape, dtype=dtype, device=device if (via_device or dev_type != "cpu") else "cpu")
            if t.numel():
                dist.broadcast(t, src=gsrc, group=self.group)
                if self.rank != src:
                    self.bytes["broadcast"] += t.numel() * t.element_size()
            out[k] = t.cpu() if dev_type == "cpu" and t.device.type != "cpu" else t
        return out

    def total(self, t_local: int) -> int:
        return t_local * self.world

    def local_slice(self, t_local: int) -> slice:
        return slice(self.rank * t_local, (self.rank + 1) * t_local)

    def all_gather_frames(self, x: torch.Tensor, dim: int) -> torch.Tensor:
        """Concatenate every rank's block along `dim` in rank order: ONE collective into ONE preallocated tensor
        (`all_gather_into_tensor`; no per-rank temporaries, no concatenation pass when dim == 0)."""
        dim = dim % x.dim()            # negative axes count from the end
        if self.world == 1 and not self.always_collective:
            return x
        x = x.contiguous()
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)   # rank-major along dim 0 (the form both RCCL and gloo take)
        self.bytes["all_gather"] += (out.numel() - x.numel()) * x.element_size()
        if dim == 0:
            return out
        # [world, ..., n_dim, ...] -> [..., world * n_dim, ...]
        out = out.view((self.world,) + tuple(x.shape))
        return out.movedim(0, dim).reshape(tuple(x.shape[:dim]) + (self.world * x.shape[dim],) + tuple(x.shape[dim + 1:]))

    def all_reduce_max(self, x: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.always_collective:
            return x
        x = x.contiguous()
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.group)
        self.bytes["all_reduce"] += x.numel() * x.element_size()
        return x

    def all_reduce_sum(self, x: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.always_collective:
            return x
        x = x.contiguous()
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes["all_reduce"] += x.numel() * x.element_size()
        return x


class ClipShard(FrameShard):
    """The frames of ONE clip held by the ranks of a group in ARBITRARY ownership: `owners[p]` = group rank that holds clip position p.
    What the sliding clip loop needs (inference/video_entity.py: frames are owned by absolute index, `f % world`, so a frame computed
    once for a window serves every overlapping clip on the rank that made it): a rank's positions need not be contiguous nor equally
    many.  Every rank of the group must own at least one position.  The interface is FrameShard's:
      total(t_local) -> T; local_slice(t_local) -> a slice when the rank's positions are contiguous, else a list of positions;
      all_gather_frames(x, dim): pads every rank's block to the largest count, ONE all_gather_into_tensor, then a gather of the valid
      rows into clip order."""

    def __init__(self, owners, group=None, always_collective=False, counter=None):
        super().__init__(group=group, always_collective=always_collective)
        if counter is not None:
            self.bytes = counter                                   # the loop's shard keeps the totals
        self.owners = [int(o) for o in owners]
        self.t_total = len(self.owners)
        assert self.t_total > 0 and all(0 <= o < self.world for o in self.owners), (self.owners, self.world)
        self.positions = [[p for p, o in enumerate(self.owners) if o == r] for r in range(self.world)]
        assert all(len(ps) > 0 for ps in self.positions), f"every rank of the group must own a frame of the clip (owners {self.owners})"
        self.counts = [len(ps) for ps in self.positions]
        self.max_count = max(self.counts)
        self.local_positions = self.positions[self.rank]
        # clip position p -> row of the padded rank-major gather
        self._order = [self.owners[p] * self.max_count + self.positions[self.owners[p]].index(p) for p in range(self.t_total)]
        self._order_cache = {}

    def total(self, t_local: int) -> int:
        assert t_local == len(self.local_positions), (t_local, self.local_positions)
        return self.t_total

    def local_slice(self, t_local: int):
        ps = self.local_positions
        assert t_local == len(ps), (t_local, ps)
        if ps == list(range(ps[0], ps[0] + len(ps))):
            return slice(ps[0], ps[0] + len(ps))
        return list(ps)

    def all_gather_frames(self, x: torch.Tensor, dim: int) -> torch.Tensor:
        dim = dim % x.dim()
        if self.world == 1 and not self.always_collective:
            return x
        n = x.shape[dim]
        assert n == len(self.local_positions), (n, self.local_positions)
        xm = x.movedim(dim, 0)
        if n < self.max_count:
            xm = torch.cat([xm, xm.new_zeros((self.max_count - n,) + tuple(xm.shape[1:]))], 0)
        xm = xm.contiguous()
        out = torch.empty((self.world * self.max_count,) + tuple(xm.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, xm, group=self.group)
        self.bytes["all_gather"] += (out.numel() - xm.numel()) * x.element_size()
        idx = self._order_cache.get(x.device)
        if idx is None:
            idx = self._order_cache[x.device] = torch.tensor(self._order, dtype=torch.long).to(x.device, non_blocking=True)
        return out.index_select(0, idx).movedim(0, dim)


def cyclic_owners(first_frame: int, num_frames: int, world: int):
    """Owner (group rank) of every position of the clip [first_frame, first_frame + num_frames) when frame f of the video belongs to
    rank f % world: fixed per absolute frame, so overlapping clips and the memory pool agree on who holds a frame's features."""
    return [(first_frame + p) % world for p in range(num_frames)]


def shard_frames(tensor_or_dict, shard: FrameShard, t_total: int):
    """Slice dim 0 (frames) of a tensor / every tensor of a dict to this rank's block."""
    assert t_total % shard.world == 0, "frames must divide evenly over the ranks"
    sl = shard.local_slice(t_total // shard.world)
    if isinstance(tensor_or_dict, dict):
        return {k: v[sl] for k, v in tensor_or_dict.items()}
    return tensor_or_dict[sl]
