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
        self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        if rank == 0:
            # one contiguous receive buffer (the gather's per-rank tensors are views of it), and the two index
            # vectors that move every received row to its place in ONE gather + scatter instead of one per rank
            maxrows = self.slot // (width * 3)
            self.recv_all = torch.empty((world * maxrows, width, 3), dtype=torch.uint8, device=device)
            self.recv = [self.recv_all[r * maxrows:(r + 1) * maxrows].view(-1) for r in range(world)]
            src = np.concatenate([r * maxrows + np.arange(len(self.rows[r]), dtype=np.int64) for r in range(world)])
            dst = np.concatenate([np.asarray(self.rows[r], dtype=np.int64) for r in range(world)])
            self.src_index = torch.as_tensor(src, device=device)
            self.dst_index = torch.as_tensor(dst, device=device)
        else:
            self.recv_all, self.recv, self.src_index, self.dst_index = None, None, None, None

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
        if self.dst_index.numel():
            self.frame.index_copy_(0, self.dst_index, self.recv_all.index_select(0, self.src_index))
        return self.frame
