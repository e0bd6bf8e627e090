"""Multi-GPU framebuffer assembly: one process per GPU, each owning interleaved scanline bands
(rtx_band_rows); the 8-bit bands are gathered to rank 0 with ONE collective (RCCL over xGMI when
the process group's backend is nccl; gloo in the CPU tests) and scattered into the frame there.
There is no other cross-device step on this path.

Two ways to drive it: gather() + assemble() (one frame, synchronous on the caller's stream), or
next_send() / submit() / flush() for a stream of frames: the gather of frame i runs on the
collective's own stream while frame i+1 is traced into the other send buffer, and rank 0 assembles
frame i when it next calls submit() (or flush()).  Every frame is gathered and assembled; only the
wait moves."""
import numpy as np
import torch
import torch.distributed as dist

from . import host


class BandGather:
    """Pre-allocated gather of every rank's packed rgb8 band into rank 0's frame."""

    DEPTH = 2  # send / receive buffers: frame i+1 is traced while frame i is on the wire

    def __init__(self, width, height, band_rows, world, rank, device, via_host=False):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.rows = [host.band_rows(height, world, r, band_rows) for r in range(world)]
        self.slot = max(len(r) for r in self.rows) * width * 3   # equal-sized slots: pad to the largest band
        self.sends = [torch.zeros(self.slot, dtype=torch.uint8, device=device) for _ in range(self.DEPTH)]
        self.send = self.sends[0]
        self.via_host = via_host   # gloo rehearsal on a GPU box: the collective moves host copies
        self._slot, self._pending = 0, None
        self.frame = torch.empty((height, width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        if rank == 0:
            # one contiguous receive buffer (the gather's per-rank tensors are views of it), and the two index
            # vectors that move every received row to its place in ONE gather + scatter instead of one per rank
            maxrows = self.slot // (width * 3)
            self.recv_alls = [torch.empty((world * maxrows, width, 3), dtype=torch.uint8, device=device) for _ in range(self.DEPTH)]
            self.recvs = [[ra[r * maxrows:(r + 1) * maxrows].view(-1) for r in range(world)] for ra in self.recv_alls]
            self.recv_all, self.recv = self.recv_alls[0], self.recvs[0]
            src = np.concatenate([r * maxrows + np.arange(len(self.rows[r]), dtype=np.int64) for r in range(world)])
            dst = np.concatenate([np.asarray(self.rows[r], dtype=np.int64) for r in range(world)])
            self.src_index = torch.as_tensor(src, device=device)
            self.dst_index = torch.as_tensor(dst, device=device)
        else:
            self.recv_all, self.recv, self.src_index, self.dst_index = None, None, None, None
            self.recv_alls, self.recvs = [None] * self.DEPTH, [None] * self.DEPTH

    @property
    def my_rows(self):
        return self.rows[self.rank]

    def gather(self):
        """Collective: every rank contributes self.send (its band, packed rows in increasing order)."""
        if self.world > 1:
            dist.gather(self.send, self.recv, dst=0)
        elif self.rank == 0:
            self.recv[0].copy_(self.send)

    def assemble(self, slot=0):
        """Rank 0: scatter the gathered bands into the (height, width, 3) frame."""
        assert self.rank == 0
        if self.dst_index.numel():
            self.frame.index_copy_(0, self.dst_index, self.recv_alls[slot].index_select(0, self.src_index))
        return self.frame

    # ---- a stream of frames ----
    def next_send(self):
        """The buffer the next frame's band goes into (packed rows, as for gather())."""
        return self.sends[self._slot]

    def submit(self):
        """Collective: start the gather of the band in next_send(), then finish the frame submitted before
        it (wait for its gather; rank 0 assembles it).  Returns rank 0's frame once one is complete, else None."""
        s = self._slot
        if self.world == 1:
            work = None
            if self.rank == 0:
                self.recvs[s][0].copy_(self.sends[s])
        elif self.via_host:
            send = self.sends[s].cpu()
            recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
            work = dist.gather(send, recv, dst=0, async_op=True)
            work = (work, recv)
        else:
            work = dist.gather(self.sends[s], self.recvs[s], dst=0, async_op=True)
        prev, self._pending = self._pending, (work, s)
        self._slot = (s + 1) % self.DEPTH
        return self._finish(prev) if prev is not None else None

    def flush(self):
        """Finish the last submitted frame.  Returns rank 0's frame (None on other ranks or if nothing is pending)."""
        prev, self._pending = self._pending, None
        return self._finish(prev) if prev is not None else None

    def _finish(self, pending):
        work, s = pending
        if isinstance(work, tuple):
            work, host_recv = work
            work.wait()
            if self.rank == 0:
                for r in range(self.world):
                    self.recvs[s][r].copy_(host_recv[r])
        elif work is not None:
            work.wait()   # the caller's stream waits for the collective's; the host does not block on RCCL
        return self.assemble(s) if self.rank == 0 else None
