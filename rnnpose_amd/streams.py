"""Helper streams of the two engines, bound to DISTINCT hardware queues.

ROCm multiplexes every HIP stream of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a stream gets
its queue when it is first used -- a new queue while fewer than the maximum exist, otherwise the least-referenced existing
one -- and two streams that land on the same queue execute strictly one after the other.  Measured (r02,
tools/loop_overlap.py): the inner loops of the two batch halves took 6.4 ms each and 13.2 ms "concurrently" when their
streams happened to share a queue, i.e. no overlap at all, and whether they did depended on the order in which unrelated
streams (capture streams, warm-up streams) had been touched before.

So the streams that must run concurrently are created once per device and TOUCHED immediately, in a fixed order, right
after the default stream: default = queue 1, chain[0] = queue 2, chain[1] = queue 3, aux = queue 4.  Everything created
later (torch's capture streams, temporary warm-up streams) shares those queues and only ever carries work that does not
need to overlap with ours.
"""
from __future__ import annotations

import torch

_sets = {}


class StreamSet:
    def __init__(self, device):
        self.chain = [torch.cuda.Stream(device=device) for _ in range(2)]   # the two batch halves / image sets
        self.aux = torch.cuda.Stream(device=device)                          # side chain of an unsplit (B = 1) step
        self.extra = [torch.cuda.Stream(device=device) for _ in range(2)]    # (experiments with more than two parts)
        for s in self.chain + [self.aux] + self.extra:
            with torch.cuda.stream(s):
                torch.zeros(1, device=device)        # first use: binds the stream to its hardware queue now
            s.synchronize()


def reserve(device) -> StreamSet:
    """The StreamSet of `device` (created and bound on first call -- call it before any other side stream is used)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _sets.get(dev)
    if st is None:
        st = _sets[dev] = StreamSet(dev)
    return st
