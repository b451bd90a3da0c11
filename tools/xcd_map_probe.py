#!/usr/bin/env python3
"""The two dispatcher behaviours the rotation's L2 tier rests on (neither is promised by HIP), watched directly through the XCC_ID
hardware register (fpv_diag_xcd_map):

  1. within ONE launch workgroups are dealt round-robin over the eight XCDs: xcd(b) = (b + s) mod 8 for every b;
  2. across the launches of a CHAIN the shift s stays the same - block b meets the same XCD (and its L2) again - as long as every
     grid is a whole number of rounds of eight (step_grid rounds up for exactly that reason); a ragged grid in between moves s.

    python tools/xcd_map_probe.py [--out gpurun_out/r06/xcd_map.json]

Prints one JSON object: per scenario the shifts of consecutive launches, whether every launch was exactly round-robin, and how many
launches kept the shift of their predecessor."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--launches", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.lib()
stream = torch.cuda.Stream(device=dev)


def probe(blocks, k, between=None):
    """k probe launches of `blocks` workgroups on one stream, `between()` enqueued after each; [k, blocks] XCD ids"""
    out = torch.full((k, blocks), 99, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        for t in range(k):
            _lib.check(L.fpv_diag_xcd_map(out[t].data_ptr(), blocks, stream.cuda_stream))
            if between is not None:
                between()
    stream.synchronize()
    return out.cpu()


def summarise(m):
    blocks = m.shape[1]
    b = torch.arange(blocks, dtype=torch.int32)
    shifts, exact = [], []
    for row in m:
        s = int((row[0] - 0) % 8)
        shifts.append(s)
        exact.append(bool(torch.equal(row, (b + s) % 8)))
    kept = sum(1 for i in range(1, len(shifts)) if shifts[i] == shifts[i - 1])
    return {"blocks": blocks, "launches": len(shifts), "every_launch_exactly_round_robin": all(exact), "launches_exactly_round_robin": sum(exact),
            "shifts": shifts, "launches_that_kept_their_predecessors_shift": kept, "xcds_seen": sorted(set(int(x) for x in m.flatten().tolist()))}


res = {}
res["chain_of_8192_block_launches (2^20 drones)"] = summarise(probe(8192, a.launches))
res["chain_of_65536_block_launches (2^23 drones)"] = summarise(probe(65536, 16))
res["chain_of_7816_block_launches (10^6 drones, rounded up to rounds of eight)"] = summarise(probe(7816, a.launches))
res["chain_of_7813_block_launches (10^6 drones, NOT rounded: what step_grid avoids)"] = summarise(probe(7813, a.launches))
# a real step launch between two probes (its grid is a multiple of eight): does the step kernel itself disturb the shift?
env = DroneBatch(load_params(fps=1000, ceiling=100.0), 1 << 20, device=dev, auto_reset=True, with_accel=False)
env.reset()
sticks = torch.zeros((1 << 20, 4), device=dev)
res["8192-block probes with a 2^20-drone fpv_step between them"] = summarise(probe(8192, a.launches, lambda: env.step(sticks, return_imu=False)))
# another kernel between two launches (a policy): torch ops with grids of their own
W = torch.randn(4, 13, device=dev)
obs = env.state[:13, :env.n]
res["8192-block probes with tanh(W @ obs) between them (a policy kernel in the chain)"] = summarise(probe(8192, a.launches, lambda: torch.tanh(W @ obs)))
torch.cuda.synchronize()
print(json.dumps(res, indent=1))
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
