"""Branch farm: the transition tree sharded over the GPUs of one node.

One process per GPU (``torchrun``), ``torch.distributed`` with backend ``nccl`` (= RCCL over xGMI
on MI355X) or ``gloo`` (CPU tests).  The program is SPMD: every rank runs the same
``BlendingEngine`` and takes the same decisions from the same data, so no control messages exist.
What moves:

* C1, native fast path: the two anchor trajectories are a B = 2 batch that every rank computes itself inside the
  same wavefront as its share of mid branches (redundant work instead of an idle phase on N-1 ranks: the mid
  branches need the anchors' step i-1 latents at every step), then ``share_anchor_pair`` — ONE ``broadcast`` of
  rank 0's packed [2 x (latent stack | frame)] (1.75 MiB at 512^2) so that the STORED anchors are bit-identical
  on every rank whatever batch width each rank computed them in;
* C1 (generic pipes, recycled / crossfed anchors): ``share_trajectory`` — ONE ``broadcast`` of the
  [steps,4,L,L] fp16 stack from the rank that denoised it (128 KiB at 512^2);
* C2/C3 per speculative round: ``exchange_branches`` — ONE all-gather of a packed byte slot per branch
  (latent stack from the injection step on + decoded uint8 frame; 832 KiB at 512^2) and
  ``exchange_scalars`` — ONE all-gather of the two float64 neighbour distances each rank measured for
  the branches it owns (LPIPS is sharded, its features are computed on demand and cached per frame).

There is no all-reduce on this path and every message is small (<= a few MiB per rank), i.e. latency-
not bandwidth-bound on xGMI: two collectives per round, never one per branch.  A branch is a pure
function of (parent stacks, conditionings, fraction), so any rank may evaluate any branch.
``check_consistent`` lets the engine verify (cheaply, once per transition) that every rank derived
the same branching plan before any collective whose shape depends on it is issued.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class BranchFarm:
    def __init__(self, group=None, device: Optional[torch.device] = None):
        assert dist.is_initialized(), "init torch.distributed first (torchrun + init_process_group)"
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.bytes_moved = 0
        self.collectives = 0
        self._send = {}             # (slots, slot bytes) -> preallocated packed send buffer of exchange_branches (reused round after round)
        self._gather_into = True    # dist.all_gather_into_tensor (one contiguous receive buffer); falls back to the list form once if the backend lacks it
        # LB_FARM_TRACE=1: wall-clock split of every exchange_branches / exchange_scalars / share_* call (pack / collective / unpack, the
        # device drained at each boundary - so tracing costs the overlap it measures: diagnostics only).  bench.py prints it in its
        # `farm` block, so that the first run on real xGMI explains itself.
        self.trace_on = os.environ.get("LB_FARM_TRACE", "0") not in ("", "0")
        self.trace: List[dict] = []

    # -- helpers ------------------------------------------------------------------------------
    def _tick(self) -> float:
        if self.trace_on and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def _trace(self, what: str, t0: float, t1: float, t2: float, t3: float, nbytes: int) -> None:
        if self.trace_on:
            self.trace.append({"op": what, "pack_ms": (t1 - t0) * 1e3, "collective_ms": (t2 - t1) * 1e3, "unpack_ms": (t3 - t2) * 1e3,
                               "bytes_per_rank": int(nbytes)})

    def trace_summary(self) -> dict:
        """Per operation: calls and the mean pack / collective / unpack milliseconds (empty without LB_FARM_TRACE)."""
        out = {}
        for e in self.trace:
            o = out.setdefault(e["op"], {"calls": 0, "pack_ms": 0.0, "collective_ms": 0.0, "unpack_ms": 0.0, "bytes_per_rank": e["bytes_per_rank"]})
            o["calls"] += 1
            for k in ("pack_ms", "collective_ms", "unpack_ms"):
                o[k] += e[k]
        for o in out.values():
            for k in ("pack_ms", "collective_ms", "unpack_ms"):
                o[k] = round(o[k] / o["calls"], 4)
        return out

    def _all_gather_packed(self, buf: torch.Tensor) -> torch.Tensor:
        """``buf`` [slots, slot] uint8 of this rank -> [world, slots, slot] of every rank: ONE collective into ONE receive buffer."""
        recv = torch.empty((self.world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
        if self._gather_into:
            try:
                dist.all_gather_into_tensor(recv, buf, group=self.group)
            except (RuntimeError, NotImplementedError, AttributeError):
                self._gather_into = False
        if not self._gather_into:
            dist.all_gather([recv[r] for r in range(self.world)], buf, group=self.group)
        self.bytes_moved += buf.numel() * buf.element_size() * (self.world - 1)
        self.collectives += 1
        return recv

    def _all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        t = t.to(self.device).contiguous()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        self.bytes_moved += t.numel() * t.element_size() * (self.world - 1)
        self.collectives += 1
        return out

    def _broadcast(self, t: torch.Tensor, src: int) -> torch.Tensor:
        t = t.to(self.device).contiguous()
        dist.broadcast(t, src=src if self.group is None else dist.get_global_rank(self.group, src), group=self.group)
        self.bytes_moved += t.numel() * t.element_size()
        self.collectives += 1
        return t

    def owner_of(self, index: int) -> int:
        """Round-robin ownership of the ``index``-th branch of a round."""
        return index % self.world

    def my_indices(self, n_total: int) -> List[int]:
        return list(range(self.rank, n_total, self.world))

    # -- plan agreement -------------------------------------------------------------------------
    def check_consistent(self, values: Sequence[float], what: str = "branching plan") -> None:
        """Every rank contributes the same vector or the run stops HERE with a clear error instead of hanging in
        a later collective with mismatched shapes (e.g. a time-budget plan derived from rank-local timings)."""
        v = torch.tensor([float(x) for x in values], dtype=torch.float64)
        got = self._all_gather(v)
        for r, other in enumerate(got):
            if other.shape != got[0].shape or not torch.equal(other.cpu(), got[0].cpu()):
                raise RuntimeError(f"BranchFarm: rank {r} derived a different {what} than rank 0: "
                                   f"{other.tolist()} vs {got[0].tolist()}")

    def broadcast_floats(self, values: Sequence[float], src: int = 0) -> List[float]:
        """Rank ``src``'s values on every rank (timings that feed the planner, noise seeds ...)."""
        t = torch.tensor([float(x) for x in values], dtype=torch.float64)
        return [float(x) for x in self._broadcast(t, src).cpu().tolist()]

    # -- C1: anchors --------------------------------------------------------------------------
    def share_trajectory(self, traj: Optional[Sequence[torch.Tensor]], owner: int, steps: int,
                         shape: Optional[Tuple[int, int, int, int]] = None):
        """Every rank gets the owner's full trajectory (list of ``steps`` latents): one broadcast.
        ``shape`` = (b, c, h, w) of one latent (every rank knows it from its own pipe)."""
        if shape is None:
            meta = torch.zeros(4, dtype=torch.int64)
            if self.rank == owner:
                meta = torch.tensor(list(traj[0].shape), dtype=torch.int64)
            shape = tuple(int(v) for v in self._broadcast(meta, owner).cpu().tolist())
        b, c, h, w = shape
        if self.rank == owner:
            stack = torch.stack([t.reshape(b, c, h, w) for t in traj]).to(self.device, torch.float16)
        else:
            stack = torch.zeros(steps, b, c, h, w, dtype=torch.float16, device=self.device)
        got = self._broadcast(stack, owner)
        return [got[i].clone() for i in range(steps)]

    def share_anchor_pair(self, trajs: Sequence[Sequence[torch.Tensor]], frames: Sequence[object], owner: int, steps: int,
                          make_frame: Callable[[torch.Tensor], object], lat_shape: Tuple[int, int, int],
                          frame_hw: Tuple[int, int]):
        """C1 on the native fast path: every rank has denoised both anchors itself (inside batches of different width, so
        the last bits may differ between ranks); ONE broadcast of the owner's packed [2 x (latent stack | frame)] makes the
        stored anchors - parents of every later branch, first / last frame of the transition - bit-identical everywhere.
        Returns ([trajectory 1, trajectory 2], [frame 1, frame 2]); the owner keeps its own objects."""
        c, h, w = lat_shape
        fh, fw = frame_hw
        lat_bytes, frm_bytes = steps * c * h * w * 2, fh * fw * 3
        slot = (lat_bytes + frm_bytes + 15) // 16 * 16
        buf = torch.zeros(2, slot, dtype=torch.uint8, device=self.device)
        if self.rank == owner:
            for k in range(2):
                lat = torch.stack([t.reshape(c, h, w) for t in trajs[k]]).to(self.device, torch.float16).contiguous()
                buf[k, :lat_bytes] = lat.view(torch.uint8).reshape(-1)
                buf[k, lat_bytes:lat_bytes + frm_bytes] = self._frame_u8(frames[k]).to(self.device).reshape(-1)
        got = self._broadcast(buf, owner)
        if self.rank == owner:
            return [list(trajs[0]), list(trajs[1])], [frames[0], frames[1]]
        out_t, out_f = [], []
        for k in range(2):
            lat = got[k, :lat_bytes].clone().view(torch.float16).view(steps, 1, c, h, w)
            out_t.append([lat[i] for i in range(steps)])
            out_f.append(make_frame(got[k, lat_bytes:lat_bytes + frm_bytes].clone().view(fh, fw, 3)))
        return out_t, out_f

    # -- C2/C3: one speculative round ------------------------------------------------------------
    def exchange_branches(self, mine: List[Tuple[list, object]], n_total: int, active_steps: int, total_steps: int,
                          make_frame: Callable[[torch.Tensor], object], lat_shape: Tuple[int, int, int],
                          frame_hw: Tuple[int, int]):
        """``mine``: (trajectory, frame) of the branches ``my_indices(n_total)`` in that order.
        Returns (trajectory, frame) of ALL ``n_total`` branches, in branch order, on every rank.
        One all-gather of a packed uint8 slot per branch: [active_steps latents fp16 | frame u8]; trajectories
        travel from the injection step on and are re-padded with ``None``.  The payload shapes are arguments
        (every rank knows them from its own pipe), so a rank that owns no branch this round needs no metadata."""
        c, h, w = lat_shape
        fh, fw = frame_hw
        lat_bytes = active_steps * c * h * w * 2
        frm_bytes = fh * fw * 3
        slot = (lat_bytes + frm_bytes + 15) // 16 * 16
        slots = (n_total + self.world - 1) // self.world
        assert len(mine) == len(self.my_indices(n_total)), (len(mine), n_total, self.rank, self.world)
        t0 = self._tick()
        # the packed send buffer lives as long as the farm (round 5 allocated and zero-filled one per round, and staged every
        # latent stack through a torch.stack temporary): latents and frame are copied straight into their slot
        buf = self._send.get((slots, slot))
        if buf is None:
            if len(self._send) >= 8:
                self._send.clear()
            buf = self._send[(slots, slot)] = torch.zeros(slots, slot, dtype=torch.uint8, device=self.device)
        for k, (traj, frame) in enumerate(mine):
            live = [t for t in traj if t is not None]
            assert len(live) == active_steps, (len(live), active_steps)
            dst = buf[k, :lat_bytes].view(torch.float16).view(active_steps, c, h, w)
            for i, t in enumerate(live):
                dst[i].copy_(t.reshape(c, h, w), non_blocking=True)
            buf[k, lat_bytes:lat_bytes + frm_bytes].copy_(self._frame_u8(frame).reshape(-1), non_blocking=True)
        t1 = self._tick()
        got = self._all_gather_packed(buf)              # [world, slots, slot]: a fresh buffer, the views below keep it alive
        t2 = self._tick()
        out = []
        for idx in range(n_total):
            r, k = idx % self.world, idx // self.world
            row = got[r, k]
            lat = row[:lat_bytes].view(torch.float16).view(active_steps, 1, c, h, w)
            traj = [None] * (total_steps - active_steps) + [lat[i] for i in range(active_steps)]
            out.append((traj, make_frame(row[lat_bytes:lat_bytes + frm_bytes].view(fh, fw, 3))))
        self._trace("exchange_branches", t0, t1, t2, self._tick(), slots * slot)
        return out

    def exchange_scalars(self, mine: Sequence[Sequence[float]], n_total: int, width: int = 2) -> List[List[float]]:
        """``mine``: ``width`` float64 values for each of ``my_indices(n_total)``; returns the values of all
        ``n_total`` branches in branch order, bit-identical on every rank (they are gathered, never recomputed)."""
        slots = (n_total + self.world - 1) // self.world
        t0 = self._tick()
        rows = [[float(v) for v in vals] for vals in mine] + [[0.0] * width] * (slots - len(mine))
        t = torch.tensor(rows, dtype=torch.float64).reshape(slots, width)         # ONE host tensor, one upload
        t1 = self._tick()
        got = self._all_gather_packed(t.to(self.device)).cpu()
        t2 = self._tick()
        vals = got.tolist()
        out = [vals[idx % self.world][idx // self.world] for idx in range(n_total)]
        self._trace("exchange_scalars", t0, t1, t2, self._tick(), slots * width * 8)
        return out

    @staticmethod
    def _frame_u8(frame) -> torch.Tensor:
        dev = getattr(frame, "_lb_u8", None)
        if dev is not None:
            return dev
        return torch.from_numpy(np.ascontiguousarray(np.asarray(frame, dtype=np.uint8)))
