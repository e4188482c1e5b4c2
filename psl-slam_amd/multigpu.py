"""Batched many-frames mode across the GPUs of one node (BASELINE config 4).

Frames and whole streams are independent, so they are sharded with no data-path collective:
stream s runs on rank s mod world.  The one exchange step is the RESULT GATHER: every rank ends a
step with fixed-size padded per-frame records (counts, keypoints [cap,7] f32, descriptors
[cap,32] u8, matches [cap] i32) and they are all-gathered with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Record sizes are tiny
next to a 153 GB/s xGMI link (<= ~64 KB per frame), so the gather is issued once per batch.
"""
import torch
import torch.distributed as dist


def streams_for_rank(n_streams, rank, world):
    """Stream s -> rank s mod world."""
    return [s for s in range(n_streams) if s % world == rank]


class ResultGather:
    """Double-buffered all-gather of one step's result tensors; overlap with the next step's compute."""

    def __init__(self, templates, world, device, group=None):
        self.world, self.group = world, group
        self.stage = [[torch.empty_like(t, device=device) for t in templates] for _ in range(2)]
        self.out = [[torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=device) for t in templates]
                    for _ in range(2)]
        self.works = [[], []]
        self.k = 0

    def submit(self, tensors):
        """Copy this step's results to a staging slot (so the producers can be overwritten) and start
        the all-gather.  Returns the slot index."""
        k = self.k
        self.wait(k)
        for s, t in zip(self.stage[k], tensors):
            s.copy_(t, non_blocking=True)
        self.works[k] = [dist.all_gather_into_tensor(o, s, group=self.group, async_op=True)
                         for o, s in zip(self.out[k], self.stage[k])]
        self.k ^= 1
        return k

    def wait(self, k=None):
        for kk in ([0, 1] if k is None else [k]):
            for w in self.works[kk]:
                w.wait()
            self.works[kk] = []

    def result(self, k):
        self.wait(k)
        return self.out[k]
