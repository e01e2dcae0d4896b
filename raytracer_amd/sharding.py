"""Sharding of one image across ranks by rows, and the gather of rendered bands to rank 0.

Pixels are independent (the reference already renders 4x4 tiles independently,
src/software_raytracer.cpp:205-240), so a frame shards by rows with no exchange while
rendering; the only collective is the final gather of packed 0xAARRGGBB bands.  One process per
GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Rows cost very different amounts (a row of sky against a row across the sphere), so a rank's
band is not a contiguous block but row tiles dealt round-robin: tile k (rows k*T .. k*T+T-1)
belongs to rank k mod N.  Every rank then renders the same mix of rows.  T defaults to 4, the
height of the reference's tiles (src/raytracer.h:28) and of the strip one wavefront renders.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def band_bounds(height: int, world: int) -> list[int]:
    """Contiguous split: world+1 row indices; band r = rows [b[r], b[r+1]).  Sizes differ by at most one row."""
    return [height * r // world for r in range(world + 1)]


def band_of(height: int, world: int, rank: int) -> tuple[int, int]:
    b = band_bounds(height, world)
    return b[rank], b[rank + 1] - b[rank]


def tiled_band(height: int, world: int, rank: int, tile_rows: int) -> dict:
    """The band of `rank` when tiles of tile_rows rows are dealt round-robin to `world` ranks, as
    the aehip_params fields that describe it (row_begin, row_count, tile_rows, tile_stride).
    tile_rows 0, or a single rank, gives the contiguous split of band_bounds."""
    if tile_rows == 0 or world == 1:
        r0, rows = band_of(height, world, rank)
        return dict(row_begin=r0, row_count=rows, tile_rows=0, tile_stride=0)
    stride = tile_rows * world
    span = max(0, height - rank * tile_rows)
    rows = span // stride * tile_rows + min(span % stride, tile_rows)
    return dict(row_begin=rank * tile_rows, row_count=rows, tile_rows=tile_rows, tile_stride=stride)


def band_image_rows(band: dict) -> list[int]:
    """Image row of every band row, in band order."""
    r0, n, t, s = band["row_begin"], band["row_count"], band["tile_rows"], band["tile_stride"]
    if t == 0:
        return list(range(r0, r0 + n))
    return [r0 + (j // t) * s + j % t for j in range(n)]


class BandGather:
    """Preallocated gather of per-rank bands into a full image on rank 0.

    The collective is dist.gather on equally sized (padded) int32 tensors: rank 0 receives from
    every peer over its direct xGMI links at once; payload is 4 bytes per pixel.  Bands arrive in
    band order; one row-gather kernel on rank 0 puts the rows in image order (contiguous equal
    bands are received straight into the image instead).

    Two local band buffers alternate so that the gather of frame i (on the collective's own
    stream) overlaps the render of frame i+1 (on the compute stream): `begin(i)` returns the
    buffer to render frame i into, `submit(i)` starts its gather without waiting, `finish()`
    waits for everything submitted.  `gather()` is the simple blocking form on buffer 0.

    On a GPU every call orders its work on the stream that is current when it is made, and
    begin()/submit() of consecutive frames may be made on two different streams (frame i on
    stream i & 1): frame i+1 then starts while the last workgroups of frame i drain, which is
    worth 2 % of a whole 4K frame and 8-15 % of an eighth of one.  A buffer is handed out again
    only after the gather that read it, wherever that gather was waited for.

    `set_columns(c0, c1)` narrows what is sent to the columns that hold more than the row's
    background (see there); the image on dst is whole either way.

    `band` holds the aehip_params fields of this rank's band (pass as **band to setup()).
    `always_collective` runs the collective and the stacked-bands path even for a single rank
    (tests of the protocol on one GPU)."""

    def __init__(self, width: int, height: int, device: torch.device, dst: int = 0, tile_rows: int = 4,
                 always_collective: bool = False):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst = dst
        self.width, self.height = width, height
        self.on_gpu = torch.device(device).type == "cuda"
        self.collective = self.world > 1 or always_collective
        self.bands = [tiled_band(height, self.world, r, tile_rows) for r in range(self.world)]
        self.band = self.bands[self.rank]
        self.row_begin, self.rows = self.band["row_begin"], self.band["row_count"]
        self.max_rows = max(b["row_count"] for b in self.bands)
        self.tiled = self.band["tile_rows"] != 0
        # local band buffers are padded to max_rows so every rank contributes the same shape
        self.locals = [torch.zeros((self.max_rows, width), dtype=torch.int32, device=device) for _ in range(2)]
        self.local = self.locals[0]
        self.even = all(b["row_count"] == self.max_rows for b in self.bands)
        self.parts = None
        self.image = None
        self._row_source = None
        # whether the destination receives the bands stacked and reorders rows (known to every rank)
        self._stacked = not (self.even and not self.tiled and not always_collective)
        if self.rank == dst:
            self.image = torch.zeros((height, width), dtype=torch.int32, device=device)
            if self.even and not self.tiled and not always_collective:
                # equal contiguous bands: receive straight into the rows of the final image
                self.parts = [self.image[b["row_begin"]:b["row_begin"] + b["row_count"]] for b in self.bands]
            else:
                # two receive buffers in turn (one per local band buffer): the rows of frame i are
                # put in image order from one while frame i+1 arrives in the other
                self._recvs = [torch.zeros((self.world, self.max_rows, width), dtype=torch.int32, device=device)
                               for _ in range(2 if self.on_gpu else 1)]
                self._recv = self._recvs[0]
                self.parts = [self._recv[r] for r in range(self.world)]
                # per receive buffer, made once (a frame at 8 ranks lasts 50 us; slicing eight views a frame is a third of that)
                self._recv_parts = [[recv[r] for r in range(self.world)] for recv in self._recvs]
                self._recv_rows = [recv.view(self.world * self.max_rows, width) for recv in self._recvs]
                # image row y comes from row _row_source[y] of the stacked received bands
                src = torch.zeros(height, dtype=torch.int64)
                for r, b in enumerate(self.bands):
                    rows = band_image_rows(b)
                    src[torch.tensor(rows, dtype=torch.int64)] = torch.arange(len(rows), dtype=torch.int64) + r * self.max_rows
                self._row_source = src.to(device)
        if not self.collective:
            # one rank: the band is the image, rendered in place; nothing to gather or copy
            self.locals = [self.image[: self.max_rows], self.image[: self.max_rows]]
            self.local = self.locals[0]
        self._pending = [None, None]
        self._narrow = None             # set_columns(): per slot, the kept columns of the band behind a background pixel a row
        self._columns = None
        self._handle = None             # inside submit(): the communication stream's hipStream_t
        self._prepacked = False         # inside submit(packed=True): the caller packed the kept columns already
        # on a GPU the library's kernels pack and expand (RAYTRACER_AMD_TORCH_COLUMNS=1: torch's gathers, for comparison)
        self._use_lib = self.on_gpu and os.environ.get("RAYTRACER_AMD_TORCH_COLUMNS") != "1"
        # per buffer: event after which the buffer may be rendered into again (an event is
        # re-recorded freely: a wait holds the record it saw)
        self._events = [torch.cuda.Event() for _ in range(2)] if self.on_gpu else None
        self._free = [None, None]
        # On a GPU the row reorder runs on a stream of its own, after the gather it belongs to: it is
        # memory traffic (33 MB read and written for a 4K frame) that fits beside the next frame's
        # arithmetic, and on a render stream it would delay that stream's next frame instead.
        # RCCL on a GPU: the gather is ordered by streams, not by the host.  Every rank has a
        # communication stream; submit() makes it wait for the band's render, issues the gather on it
        # in the blocking form -- which blocks streams, not the host: the collective runs on the
        # process group's stream after the current one and the current one waits for it -- and, on
        # the destination, the row reorder behind it.  (The asynchronous form with a later wait()
        # costs the host 44 us a frame against 12 us; at 8 ranks a frame lasts 50 us.)
        self._stream_ordered = bool(self.on_gpu and self.collective and dist.is_initialized() and dist.get_backend() == "nccl"
                                    and os.environ.get("RAYTRACER_AMD_ASYNC_GATHER") != "1")       # (1: the asynchronous form, for comparison)
        self._comm_stream = torch.cuda.Stream(device=device) if self._stream_ordered else None
        self._comm_handle = self._comm_stream.cuda_stream if self._stream_ordered else None
        self._rendered = [torch.cuda.Event() for _ in range(2)] if self._stream_ordered else None
        self._reorder_stream = torch.cuda.Stream(device=device) if (self.on_gpu and self._row_source is not None and self.collective
                                                                    and not self._stream_ordered) else None

    # ------------------------------------------------------------------ fewer bytes on the wire

    def set_columns(self, col_begin: int, col_end: int) -> bool:
        """Every band pixel outside columns [col_begin, col_end) holds its row's background colour
        (HipRaytracer.background_columns(); the same two numbers on every rank): from now on a rank
        sends those columns and one background pixel a row, and the destination's row reorder writes
        the background around them.  A peer's band crosses one xGMI link (about 50-75 GB/s one way):
        at a fraction of a millisecond a frame the bytes on that link, not the GPUs, bound the
        frame rate, and on the reference's scene the sphere's columns are 40 % of the frame's.
        Returns whether the narrow form is in use ((0, width), or columns that save less than a
        quarter, switch it off).  Call between frames (nothing submitted and not finished)."""
        width = self.width
        col_begin, col_end = max(0, int(col_begin)), min(width, int(col_end))
        kept = col_end - col_begin + 1
        narrow = bool(self.collective and self._stacked and col_end > col_begin and kept * 4 <= width * 3
                      and os.environ.get("RAYTRACER_AMD_WHOLE_ROWS") != "1")       # (1: whole rows on the wire, for comparison)
        self._agree_on(col_begin, col_end, narrow)
        if self._narrow is not None and narrow and self._columns == (col_begin, col_end):
            return True                 # as it is: the buffers and the element map stay
        # The buffers of the previous columns may still be read by the communication or the reorder stream
        # (a frame submitted and not finished): wait for both before they go back to the allocator.
        if self._narrow is not None:
            self.finish()
        self._narrow = None
        self._columns = (col_begin, col_end)
        device = self.locals[0].device
        if not narrow:
            return False
        # a background pixel of each row: the first column if it is outside the kept ones, else the last
        bg = 0 if col_begin > 0 else width - 1
        self._col_index = torch.cat([torch.tensor([bg], dtype=torch.int64), torch.arange(col_begin, col_end, dtype=torch.int64)]).to(device)
        self._narrow = [torch.zeros((self.max_rows, kept), dtype=torch.int32, device=device) for _ in range(2)]
        if self.rank == self.dst:
            self._nrecvs = [torch.zeros((self.world, self.max_rows, kept), dtype=torch.int32, device=device)
                            for _ in range(2 if self.on_gpu else 1)]
            self._nrecv_parts = [[recv[r] for r in range(self.world)] for recv in self._nrecvs]
            self._nrecv_flat = [recv.view(-1) for recv in self._nrecvs]
            if not self._use_lib:
                # image pixel (y, x) comes from word _row_source[y] * kept + (x - col_begin + 1 inside the kept columns, 0 outside)
                # (on a GPU the library's expand kernel works this out per pixel: a 4K frame's map is 33 MB to read a frame,
                # and torch's gather through it took 68 us where the kernel takes a fraction of that)
                cols = torch.zeros(width, dtype=torch.int64)
                cols[col_begin:col_end] = torch.arange(1, kept, dtype=torch.int64)
                source = self._row_source.cpu().view(-1, 1) * kept + cols.view(1, -1)
                self._elem_source = source.view(-1).to(torch.int32).to(device)
                self._image_flat = self.image.view(-1)
        self._bg_col = bg
        if self._use_lib:
            import raytracer_amd
            self._lib = raytracer_amd.load()
        return True

    def _agree_on(self, col_begin: int, col_end: int, narrow: bool) -> None:
        """Every rank must send the same shape: a rank with other columns, or one that kept whole rows, would
        leave the gather waiting for bytes that never come.  One small all-reduce per call of set_columns
        (between frames), min and max of the three numbers in one tensor."""
        if not (dist.is_initialized() and self.world > 1):
            return
        mine = [col_begin, col_end, int(narrow)]
        on = self.locals[0].device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(mine + [-v for v in mine], dtype=torch.int64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        lo, hi = t[:3].tolist(), [-v for v in t[3:].tolist()]
        if lo != hi:
            raise RuntimeError(f"set_columns: ranks disagree on what they send (rank {self.rank}: columns {col_begin}..{col_end}, "
                               f"narrow {narrow}; over ranks min {lo}, max {hi})")

    def _stream_handle(self) -> int:
        """hipStream_t of the stream the gather's own kernels go to: the communication stream inside submit(), else the current one."""
        return self._handle if self._handle is not None else torch.cuda.current_stream().cuda_stream

    def _pack(self, slot: int) -> torch.Tensor:
        """The tensor this rank contributes for `slot`: the band, or its kept columns behind a background pixel a row."""
        if self._narrow is None:
            return self.locals[slot]
        if self._prepacked:
            return self._narrow[slot]
        if self._use_lib:
            c0, c1 = self._columns
            rc = self._lib.aehip_band_pack_columns(self.locals[slot].data_ptr(), self.max_rows, self.width, c0, c1, self._bg_col,
                                                   self._narrow[slot].data_ptr(), self._stream_handle())
            if rc != 0:
                raise RuntimeError(f"aehip_band_pack_columns failed ({rc}): {self._lib.aehip_last_error().decode()}")
            return self._narrow[slot]
        return torch.index_select(self.locals[slot], 1, self._col_index, out=self._narrow[slot])

    def _receive_parts(self, slot: int):
        if self._narrow is not None:
            return self._nrecv_parts[slot if len(self._nrecvs) > 1 else 0]
        if self._row_source is not None and len(self._recvs) > 1:
            return self._recv_parts[slot]
        return self.parts

    # ------------------------------------------------------------------ blocking form

    def gather(self) -> torch.Tensor | None:
        """Collects every rank's band from buffer 0; returns the assembled image on dst, None elsewhere."""
        self._prepacked = False
        self._start(0, async_op=False)
        return self._assemble()

    # ------------------------------------------------------------------ overlapped form

    def begin(self, frame: int, stream: "torch.cuda.Stream | None" = None) -> torch.Tensor:
        """Buffer to render `frame` into.  Waits (stream-side) for the gather that last read it.
        `stream`: the current stream, for callers that have it at hand (looking it up costs 3 us)."""
        slot = frame & 1
        self._complete(slot)
        if self._free[slot] is not None:
            (stream if stream is not None else torch.cuda.current_stream()).wait_event(self._free[slot])
        return self.locals[slot]

    def pack(self, slot: int, stream_handle: int) -> None:
        """Packs buffer `slot`'s kept columns on the caller's stream (a hipStream_t), for callers that queue the
        render and the pack together -- a HIP graph of the two, say -- and then submit(..., packed=True).
        Nothing to do while whole rows are sent."""
        if self._narrow is None:
            return
        if not self._use_lib:
            raise RuntimeError("BandGather.pack needs the library's pack kernel (a GPU, RAYTRACER_AMD_TORCH_COLUMNS unset)")
        c0, c1 = self._columns
        rc = self._lib.aehip_band_pack_columns(self.locals[slot].data_ptr(), self.max_rows, self.width, c0, c1, self._bg_col,
                                               self._narrow[slot].data_ptr(), stream_handle)
        if rc != 0:
            raise RuntimeError(f"aehip_band_pack_columns failed ({rc}): {self._lib.aehip_last_error().decode()}")

    def submit(self, frame: int, stream: "torch.cuda.Stream | None" = None, packed: bool = False) -> None:
        """Starts the gather of the band rendered into begin(frame)'s buffer and returns at once.
        At most one gather is in flight: the previous frame's gather, which had that frame's
        successor render to hide behind, is completed (and its rows put in image order) first, so
        frames land in the destination image in order on any backend.
        `stream`: the current stream (the one the band was rendered on), as for begin().
        `packed`: the caller has packed the buffer's kept columns itself, behind the render (pack())."""
        slot = frame & 1
        self._prepacked = packed and self._narrow is not None
        if self._stream_ordered:
            render_stream = stream if stream is not None else torch.cuda.current_stream()
            self._rendered[slot].record(render_stream)
            torch.cuda.set_stream(self._comm_stream)        # (set_stream twice: 1 us; the context manager: 7)
            self._handle = self._comm_handle
            try:
                self._comm_stream.wait_event(self._rendered[slot])
                dist.gather(self._pack(slot), self._receive_parts(slot) if self.rank == self.dst else None, dst=self.dst)
                self._assemble(slot)
                self._free[slot] = self._events[slot]       # the band buffer (and this slot's receive buffer) may be reused
                self._free[slot].record(self._comm_stream)
            finally:
                self._handle = None
                torch.cuda.set_stream(render_stream)
            return
        self._complete(slot ^ 1)
        self._pending[slot] = self._start(slot, async_op=True)

    def finish(self) -> torch.Tensor | None:
        for slot in (0, 1):
            self._complete(slot)
        if self._reorder_stream is not None:
            torch.cuda.current_stream().wait_stream(self._reorder_stream)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        return self.image if self.rank == self.dst else None

    # ------------------------------------------------------------------ internals

    def _complete(self, slot: int) -> None:
        if self._pending[slot] is not None:
            if self._reorder_stream is not None:
                # wait for the gather and reorder on the side stream; frames reach the image in
                # order because that stream runs them in order
                with torch.cuda.stream(self._reorder_stream):
                    self._pending[slot].wait()
                    self._assemble(slot)
                    self._free[slot] = self._events[slot]
                    self._free[slot].record()
                self._pending[slot] = None
                return
            self._pending[slot].wait()
            self._pending[slot] = None
            self._assemble(slot)
            if self.on_gpu:
                # the stream this ran on may not be the one that renders into the buffer next
                self._free[slot] = self._events[slot]
                self._free[slot].record()

    def _start(self, slot: int, async_op: bool):
        if not self.collective:
            return None                 # rendered straight into the image
        parts = None
        if self.rank == self.dst:
            parts = self._receive_parts(slot)
            if self._row_source is not None and len(self._recvs) > 1 and self._free[slot] is not None:
                # this slot's receive buffer; its previous frame's rows were reordered before the
                # event that also frees the slot's band buffer, which begin() made this stream wait for
                torch.cuda.current_stream().wait_event(self._free[slot])
        return dist.gather(self._pack(slot), parts, dst=self.dst, async_op=async_op)

    def _assemble(self, slot: int = 0) -> torch.Tensor | None:
        if self.rank != self.dst:
            return None
        if self._narrow is not None and self._use_lib:
            c0, c1 = self._columns
            rc = self._lib.aehip_band_expand_columns(self._nrecvs[slot if len(self._nrecvs) > 1 else 0].data_ptr(), self.world * self.max_rows,
                                                     self._row_source.data_ptr(), self.height, self.width, c0, c1, self.image.data_ptr(),
                                                     self._stream_handle())
            if rc != 0:
                raise RuntimeError(f"aehip_band_expand_columns failed ({rc}): {self._lib.aehip_last_error().decode()}")
        elif self._narrow is not None:
            torch.index_select(self._nrecv_flat[slot if len(self._nrecvs) > 1 else 0], 0, self._elem_source, out=self._image_flat)
        elif self._row_source is not None and self.collective:
            torch.index_select(self._recv_rows[slot if len(self._recvs) > 1 else 0], 0, self._row_source, out=self.image)
        return self.image
