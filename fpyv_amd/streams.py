"""Streams for kernel chains that are meant to overlap.

Two chains of dependent kernels overlap on the GPU only when their streams sit on different hardware queues.  The HIP
runtime owns a handful of hardware queues per device and shares them among streams once all are handed out; chains on a
shared queue do not overlap - they interleave, slower than one chain alone (measured on MI355X, 2^20 drones as two
partitions: 21.6 us per step on separate queues, 26-35 us on a shared one; `profiles/archive/r04_exp_split_streams.log`).  Which
stream lands on which queue is the runtime's business and differs from stream to stream, so this module does not guess:
it MEASURES - a chain of time-bounded one-wave kernels (`fpv_diag_busy`) on each of two streams takes as long as one
chain alone when the queues differ and twice as long when they are shared - and keeps drawing streams until it holds a
set whose members all overlap with each other (and with the streams the caller wants left alone).
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Sequence, Tuple

import torch

from . import _lib

_BUSY_US, _CHAIN = 40.0, 6          # one probe: two chains of 6 x 40 us = 0.24 ms when they overlap, 0.48 ms when they do not


def chain_time_ratio(a: torch.cuda.Stream, b: torch.cuda.Stream) -> float:
    """Wall time of two chains of busy kernels, one on `a` and one on `b`, divided by the time of one chain:
    ~1.0 = the streams run side by side (different hardware queues), ~2.0 = they take turns (a shared queue, or a == b)."""
    L = _lib.lib()
    dev = a.device
    with torch.cuda.device(dev):
        def run(streams):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(_CHAIN):
                for s in streams:
                    _lib.check(L.fpv_diag_busy(_BUSY_US, s.cuda_stream))
            for s in streams:
                s.synchronize()
            return time.perf_counter() - t0
        run((a,))                                   # first use of a stream sets its queue up: not part of the measurement
        run((b,))
        one = min(run((a,)), run((b,)))
        both = min(run((a, b)), run((a, b)))
    return both / one


def overlapping_streams(device: Any, count: int, avoid: Sequence[torch.cuda.Stream] = (), max_draws: int = 16,
                        threshold: float = 1.2) -> Tuple[List[torch.cuda.Stream], Dict[str, Any]]:
    """`count` streams on `device` whose kernel chains overlap pairwise, and with every stream in `avoid` (typically the
    caller's current stream, where a policy runs).  Draws fresh `torch.cuda.Stream`s and keeps one when its measured
    chain-time ratio against every stream already held is below `threshold`; after `max_draws` draws the best remaining
    candidates are taken as they are and the report says so (`verified: False`) - correctness never depends on the
    choice, only the overlap does.  (Streams that fully share a queue measure 2.0, streams on queues of different
    priority classes 1.35 - and step 50 % slower as partitions -, independent ones 1.04-1.06.)"""
    device = torch.device(device)
    held: List[torch.cuda.Stream] = []
    report: Dict[str, Any] = {"draws": 0, "ratios": [], "verified": True}
    rejected: List[Tuple[float, torch.cuda.Stream]] = []
    while len(held) < count and report["draws"] < max_draws:
        s = torch.cuda.Stream(device=device)
        report["draws"] += 1
        worst = max([chain_time_ratio(s, o) for o in list(avoid) + held] or [1.0])
        report["ratios"].append(round(worst, 2))
        if worst < threshold:
            held.append(s)
        else:
            rejected.append((worst, s))
    if len(held) < count:
        report["verified"] = False
        rejected.sort(key=lambda x: x[0])
        held += [s for _, s in rejected[:count - len(held)]]
        while len(held) < count:
            held.append(torch.cuda.Stream(device=device))
    return held, report
