"""Row-band partition of a frame over ranks and the gather of the framebuffer to rank 0.

One process per GPU.  The scene is replicated, the frame is cut into bands of `band_rows` rows
dealt round-robin to the ranks (band b -> rank b % world), every rank renders its bands into a
compact buffer, and the only exchange is one gather of that buffer to rank 0 (RCCL over xGMI
with the "nccl" backend; "gloo" on CPU tensors works the same way and is what the CPU tests use).
The reference has no counterpart: its two threads share one address space and interleave single
rows (Driver.cpp:377, 741).
"""
import numpy as np
import torch

from . import part_row_indices


class FrameGather:
    def __init__(self, resx, resy, band_rows, world, rank, device, dist=None, channels=3, dtype=torch.uint8):
        self.resx, self.resy, self.band_rows = resx, resy, band_rows
        self.world, self.rank, self.dist = world, rank, dist
        self.rows = [part_row_indices(resy, band_rows, world, r) for r in range(world)]
        self.max_rows = max(len(r) for r in self.rows)
        # equal-sized pieces, as the gather collective requires; short parts leave their tail unused
        self.local = torch.zeros((self.max_rows, resx, channels), dtype=dtype, device=device)
        self.frame = None
        self._pieces = None
        self._index = None
        if rank == 0:
            self.frame = torch.zeros((resy, resx, channels), dtype=dtype, device=device)
            if world > 1:
                self._pieces = [torch.zeros_like(self.local) for _ in range(world)]
                self._index = [torch.from_numpy(np.ascontiguousarray(r)).to(device) for r in self.rows]

    @property
    def local_rows(self):
        return len(self.rows[self.rank])

    def gather(self):
        """Collects every rank's rows on rank 0 and puts them in frame order.  Returns the frame on rank 0."""
        if self.world == 1:
            self.frame[:] = self.local[: self.resy]
            return self.frame
        self.dist.gather(self.local, self._pieces if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world):
            n = self._index[r].numel()
            if n:
                self.frame.index_copy_(0, self._index[r], self._pieces[r][:n])
        return self.frame
