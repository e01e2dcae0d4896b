"""Row-band sharding of one image across ranks, and the gather of rendered bands to rank 0.

Pixels are independent (the reference already renders 4x4 tiles independently,
src/software_raytracer.cpp:205-240), so a frame shards by rows with no exchange while
rendering; the only collective is the final gather of packed 0xAARRGGBB bands.  One process per
GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def band_bounds(height: int, world: int) -> list[int]:
    """world+1 row indices; rank r renders rows [b[r], b[r+1]).  Bands differ by at most one row."""
    return [height * r // world for r in range(world + 1)]


def band_of(height: int, world: int, rank: int) -> tuple[int, int]:
    b = band_bounds(height, world)
    return b[rank], b[rank + 1] - b[rank]


class BandGather:
    """Preallocated gather of per-rank bands into a full image on rank 0.

    The collective is dist.gather on equally sized (padded) int32 tensors: rank 0 receives from
    every peer over its direct xGMI links at once; payload is 4 bytes per pixel."""

    def __init__(self, width: int, height: int, device: torch.device, dst: int = 0):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst = dst
        self.width, self.height = width, height
        self.bounds = band_bounds(height, self.world)
        self.max_rows = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        self.row_begin, self.rows = self.bounds[self.rank], self.bounds[self.rank + 1] - self.bounds[self.rank]
        # local band buffer is padded to max_rows so every rank contributes the same shape
        self.local = torch.zeros((self.max_rows, width), dtype=torch.int32, device=device)
        self.even = all(self.bounds[r + 1] - self.bounds[r] == self.max_rows for r in range(self.world))
        if self.rank == dst:
            self.image = torch.zeros((height, width), dtype=torch.int32, device=device)
            if self.even:
                # equal bands: receive straight into the rows of the final image, no assembly copy
                self.parts = [self.image[self.bounds[r]:self.bounds[r + 1]] for r in range(self.world)]
            else:
                self.parts = [torch.zeros((self.max_rows, width), dtype=torch.int32, device=device) for _ in range(self.world)]
        else:
            self.parts = None
            self.image = None

    def gather(self) -> torch.Tensor | None:
        """Collects every rank's band; returns the assembled image on dst, None elsewhere."""
        if self.world == 1:
            self.image[: self.rows].copy_(self.local[: self.rows])
            return self.image
        dist.gather(self.local, self.parts if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        if self.even:
            return self.image
        for r in range(self.world):
            r0, r1 = self.bounds[r], self.bounds[r + 1]
            self.image[r0:r1].copy_(self.parts[r][: r1 - r0])
        return self.image
