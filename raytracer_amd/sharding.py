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
    every peer over its direct xGMI links at once; payload is 4 bytes per pixel.

    Two local band buffers alternate so that the gather of frame i (on the collective's own
    stream) overlaps the render of frame i+1 (on the compute stream): `begin(i)` returns the
    buffer to render frame i into, `submit(i)` starts its gather without waiting, `finish()`
    waits for everything submitted.  `gather()` is the simple blocking form on buffer 0."""

    def __init__(self, width: int, height: int, device: torch.device, dst: int = 0):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst = dst
        self.width, self.height = width, height
        self.bounds = band_bounds(height, self.world)
        self.max_rows = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        self.row_begin, self.rows = self.bounds[self.rank], self.bounds[self.rank + 1] - self.bounds[self.rank]
        # local band buffers are padded to max_rows so every rank contributes the same shape
        self.locals = [torch.zeros((self.max_rows, width), dtype=torch.int32, device=device) for _ in range(2)]
        self.local = self.locals[0]
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
        self._pending = [None, None]

    # ------------------------------------------------------------------ blocking form

    def gather(self) -> torch.Tensor | None:
        """Collects every rank's band from buffer 0; returns the assembled image on dst, None elsewhere."""
        self._start(0, async_op=False)
        return self._assemble()

    # ------------------------------------------------------------------ overlapped form

    def begin(self, frame: int) -> torch.Tensor:
        """Buffer to render `frame` into.  Waits (stream-side) for the gather that last read it."""
        slot = frame & 1
        if self._pending[slot] is not None:
            self._pending[slot].wait()
            self._pending[slot] = None
        return self.locals[slot]

    def submit(self, frame: int) -> None:
        """Starts the gather of the band rendered into begin(frame)'s buffer and returns at once.
        At most one gather is in flight: the previous frame's gather, which had that frame's
        successor render to hide behind, is completed first, so frames land in the destination
        image in order on any backend."""
        slot = frame & 1
        other = slot ^ 1
        if self._pending[other] is not None:
            self._pending[other].wait()
            self._pending[other] = None
        self._pending[slot] = self._start(slot, async_op=True)

    def finish(self) -> torch.Tensor | None:
        for slot in (0, 1):
            if self._pending[slot] is not None:
                self._pending[slot].wait()
                self._pending[slot] = None
        return self._assemble()

    # ------------------------------------------------------------------ internals

    def _start(self, slot: int, async_op: bool):
        src = self.locals[slot]
        if self.world == 1:
            self.image[: self.rows].copy_(src[: self.rows])
            return None
        return dist.gather(src, self.parts if self.rank == self.dst else None, dst=self.dst, async_op=async_op)

    def _assemble(self) -> torch.Tensor | None:
        if self.rank != self.dst:
            return None
        if self.world == 1 or self.even:
            return self.image
        for r in range(self.world):
            r0, r1 = self.bounds[r], self.bounds[r + 1]
            self.image[r0:r1].copy_(self.parts[r][: r1 - r0])
        return self.image
