"""Multi-GPU framebuffer assembly: one process per GPU, each owning interleaved scanline bands
(rtx_band_rows); the 8-bit bands are gathered to rank 0 with ONE collective (RCCL over xGMI when
the process group's backend is nccl; gloo in the CPU tests) and scattered into the frame there.
There is no other cross-device step on this path."""
import numpy as np
import torch
import torch.distributed as dist

from . import host


class BandGather:
    """Pre-allocated gather of every rank's packed rgb8 band into rank 0's frame."""

    def __init__(self, width, height, band_rows, world, rank, device):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.rows = [host.band_rows(height, world, r, band_rows) for r in range(world)]
        self.slot = max(len(r) for r in self.rows) * width * 3   # equal-sized slots: pad to the largest band
        self.send = torch.zeros(self.slot, dtype=torch.uint8, device=device)
        self.recv = [torch.empty(self.slot, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        self.row_index = [torch.as_tensor(np.asarray(r, dtype=np.int64), device=device) for r in self.rows] if rank == 0 else None

    @property
    def my_rows(self):
        return self.rows[self.rank]

    def gather(self):
        """Collective: every rank contributes self.send (its band, packed rows in increasing order)."""
        if self.world > 1:
            dist.gather(self.send, self.recv, dst=0)
        elif self.rank == 0:
            self.recv[0].copy_(self.send)

    def assemble(self):
        """Rank 0: scatter the gathered bands into the (height, width, 3) frame."""
        assert self.rank == 0
        for r in range(self.world):
            n = len(self.rows[r])
            if n:
                self.frame[self.row_index[r]] = self.recv[r][: n * self.width * 3].view(n, self.width, 3)
        return self.frame
