"""Branch farm: the transition tree sharded over the GPUs of one node.

One process per GPU (``torchrun``), ``torch.distributed`` with backend ``nccl`` (= RCCL over xGMI
on MI355X) or ``gloo`` (CPU tests).  The program is SPMD: every rank runs the same
``BlendingEngine`` and takes the same decisions from the same data, so no control messages exist.
Only two kinds of payload ever move:

* C1  the two anchor latent stacks, from the rank that denoised them to everybody
      (``share_trajectory``: one all-gather of a [steps,4,L,L] fp16 stack, 128 KiB at 512^2);
* C2/C3  per speculative round, the branches each rank evaluated: latent stack from the injection
      step on, decoded uint8 frame, and the two neighbour distances (``exchange_branches``: three
      all-gathers of fixed-size slots).

There is no all-reduce on this path and every message is small (<= a few MiB), i.e. latency- not
bandwidth-bound on xGMI: one collective per payload kind per round, never per branch.
A branch is a pure function of (parent stacks, conditionings, fraction), so any rank may evaluate
any branch and a lost branch can simply be recomputed.
"""
from __future__ import annotations

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

    # -- helpers ------------------------------------------------------------------------------
    def _all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        t = t.to(self.device).contiguous()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        self.bytes_moved += t.numel() * t.element_size() * (self.world - 1)
        self.collectives += 1
        return out

    # -- C1: anchors --------------------------------------------------------------------------
    def share_trajectory(self, traj: Optional[Sequence[torch.Tensor]], owner: int, steps: int, like=None):
        """Every rank gets the owner's full trajectory (list of ``steps`` latents)."""
        meta = torch.zeros(5, dtype=torch.int64)
        if self.rank == owner:
            z = traj[0]
            meta = torch.tensor([z.shape[0], z.shape[1], z.shape[2], z.shape[3], 0], dtype=torch.int64)
        metas = self._all_gather(meta)
        b, c, h, w, _ = [int(v) for v in metas[owner].tolist()]
        if self.rank == owner:
            stack = torch.stack([t.reshape(b, c, h, w) for t in traj]).to(self.device, torch.float16)
        else:
            stack = torch.zeros(steps, b, c, h, w, dtype=torch.float16, device=self.device)
        got = self._all_gather(stack)[owner]
        return [got[i].clone() for i in range(steps)]

    # -- C2/C3: one speculative round ------------------------------------------------------------
    def exchange_branches(self, mine: List[Tuple[list, object, float, float]], n_total: int, active_steps: int,
                          total_steps: int, make_frame: Callable[[torch.Tensor], object]):
        """``mine``: results (trajectory, frame, sim_left, sim_right) of specs rank, rank+world, ...
        Returns the results of ALL ``n_total`` specs, in spec order, on every rank.  Trajectories
        travel from the injection step on (``active_steps`` latents) and are re-padded with ``None``."""
        slots = (n_total + self.world - 1) // self.world
        shape = torch.zeros(6, dtype=torch.int64)          # every rank learns the payload shapes
        if mine:
            z, f = mine[0][0][-1], self._frame_u8(mine[0][1])
            shape = torch.tensor([z.shape[-3], z.shape[-2], z.shape[-1], f.shape[0], f.shape[1], 1], dtype=torch.int64)
        ref = next(s for s in self._all_gather(shape) if int(s[5]) == 1)
        c, h, w, fh, fw, _ = [int(v) for v in ref.tolist()]
        lat = torch.zeros(slots, active_steps, c, h, w, dtype=torch.float16, device=self.device)
        frm = torch.zeros(slots, fh, fw, 3, dtype=torch.uint8, device=self.device)
        sim = torch.zeros(slots, 2, dtype=torch.float64, device=self.device)
        for k, (traj, frame, sl, sr) in enumerate(mine):
            live = [t for t in traj if t is not None]
            assert len(live) == active_steps, (len(live), active_steps)
            lat[k] = torch.stack([t.reshape(c, h, w) for t in live]).to(self.device, torch.float16)
            frm[k] = self._frame_u8(frame).to(self.device)
            sim[k, 0], sim[k, 1] = float(sl), float(sr)
        lats, frms, sims = self._all_gather(lat), self._all_gather(frm), self._all_gather(sim)
        sims = [x.cpu() for x in sims]
        out = []
        for idx in range(n_total):
            r, k = idx % self.world, idx // self.world
            traj = [None] * (total_steps - active_steps) + [lats[r][k, i].unsqueeze(0).clone() for i in range(active_steps)]
            out.append((traj, make_frame(frms[r][k].clone()), float(sims[r][k, 0]), float(sims[r][k, 1])))
        return out

    @staticmethod
    def _frame_u8(frame) -> torch.Tensor:
        dev = getattr(frame, "_lb_u8", None)
        if dev is not None:
            return dev
        return torch.from_numpy(np.ascontiguousarray(np.asarray(frame, dtype=np.uint8)))
