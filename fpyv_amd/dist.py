"""Multi-GPU sharding: one process per GPU, drones split into contiguous shards.

Every drone is independent (nothing in Drone.step couples two drones,
/root/reference/src/utils/components.py:220-248), so the physics needs no collective.  The only
exchange is the per-step all-gather of the done mask (and, on request, episode returns) for a
learner that wants the global view; it runs through torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) on the collective's own stream, double-buffered so the
next step's kernel never waits for it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n_total % world_size ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_total: int, world_size: int) -> List[int]:
    return [shard_range(n_total, world_size, r)[1] - shard_range(n_total, world_size, r)[0]
            for r in range(world_size)]


def pack_done_bits(done: torch.Tensor) -> torch.Tensor:
    """uint8 [n] -> int64 [ceil(n/64)], bit (i % 64) of word i // 64 = done[i]; the layout the
    kernel's wave ballot writes (fpv_buffers_t.done_bits).  Host/torch helper for tests and for
    backends without the kernel."""
    n = done.numel()
    pad = (-n) % 64
    d = torch.cat([done.to(torch.int64), done.new_zeros(pad, dtype=torch.int64)]) if pad else done.to(torch.int64)
    w = d.view(-1, 64)
    lo = (w[:, :63] << torch.arange(63, dtype=torch.int64, device=done.device)).sum(dim=1)
    hi = w[:, 63] * torch.iinfo(torch.int64).min     # bit 63 is the sign bit
    return lo + hi


def unpack_done_bits(bits: torch.Tensor, n: int) -> torch.Tensor:
    sh = torch.arange(64, dtype=torch.int64, device=bits.device)
    return ((bits.view(-1, 1) >> sh) & 1).reshape(-1)[:n].to(torch.uint8)


class DoneGather:
    """Double-buffered, bucketed, asynchronous all-gather of a per-shard tensor written every step.

    The kernel writes step t's mask into row t % block of the current bucket; every `block` steps the
    whole bucket goes out as ONE collective (async), so the host pays one collective launch per
    `block` steps instead of one per step (measured: a per-step `all_gather_into_tensor` costs ~25 us
    of host time, more than the 23 us step kernel; bucketed it disappears) and xGMI sees few, large
    messages.  Two buckets alternate: a bucket is only re-used after its previous collective has
    been waited for on the stream (not on the host).

        ptr = g.row(t)        # tensor the kernel writes at step t
        ... launch the step ...
        g.step_done(t)        # enqueues the all-gather when step t closed a bucket
        g.flush(t_last)       # gather a partially filled last bucket: only its filled rows travel
        g.result(b)           # [world, rows, *shard] masks of bucket b (rows = block, or the filled rows
                              # of a flushed bucket); waits for it

    Shards must be the same size on every rank (pad the last one)."""

    def __init__(self, shard_shape, dtype, device, block: int = 1, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.block = int(block)
        self.world = dist.get_world_size(group)
        shape = (self.block,) + tuple(shard_shape)
        self.local = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(2)]
        # flat concatenation: the one output shape both RCCL and gloo accept for all_gather_into_tensor
        self._flat = [torch.zeros(self.world * self.local[0].numel(), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [f.view((self.world,) + shape) for f in self._flat]
        self._view = list(self.gathered)           # what result() returns per bucket (a flushed bucket is shorter)
        self._row_numel = self.local[0][0].numel()
        self.pending: List[Optional[object]] = [None, None]
        self.launched = 0                          # collectives enqueued so far (warm-up included)
        self._row_ptrs = [[self.local[k][r].data_ptr() for r in range(self.block)] for k in range(2)]

    def row_ptr(self, t: int) -> int:
        """Device address of row(t) without building a tensor view (per-step hot path)."""
        b, r = divmod(t, self.block)
        if r == 0:
            self._wait(b & 1)
        return self._row_ptrs[b & 1][r]

    def _wait(self, k: int) -> None:
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None

    def row(self, t: int) -> torch.Tensor:
        b, r = divmod(t, self.block)
        if r == 0:
            self._wait(b & 1)             # the bucket is about to be overwritten
        return self.local[b & 1][r]

    def _launch(self, k: int, rows: Optional[int] = None) -> None:
        rows = self.block if rows is None else rows
        m = rows * self._row_numel
        out = self._flat[k][:self.world * m]
        self._view[k] = out.view((self.world, rows) + tuple(self.local[k].shape[1:]))
        self.pending[k] = self._dist.all_gather_into_tensor(out, self.local[k][:rows].reshape(-1), group=self.group,
                                                            async_op=True)
        self.launched += 1

    def step_done(self, t: int) -> None:
        if (t + 1) % self.block == 0:
            self._launch((t // self.block) & 1)

    def flush(self, t_last: int) -> None:
        """Gather the bucket holding step t_last if it was not closed by step_done: only the rows filled so
        far are sent (a short run, or the tail of a long one, does not pay for a whole bucket)."""
        rows = (t_last + 1) % self.block
        if rows != 0:
            self._launch((t_last // self.block) & 1, rows)

    def result(self, bucket: int) -> torch.Tensor:
        self._wait(bucket & 1)
        return self._view[bucket & 1]

    def warm_up(self) -> None:
        """Run both message shapes once (a whole bucket, a flushed tail) so that the backend's first-use costs -
        communicator channels, buffer registration - are paid before a timed region; leaves the buckets zeroed."""
        for k in range(2):
            self._launch(k)
            self._wait(k)
            self._launch(k, 1)
            self._wait(k)
            self.local[k].zero_()

    def drain(self) -> None:
        self._wait(0)
        self._wait(1)


# ---------------------------------------------------------------------------------------------------------------------
# Bringing RCCL up without betting the job on it.
#
# A process that has initialised the GPU cannot be replaced (no exec), and a rank stuck inside a collective library's
# bootstrap cannot be rescued from within.  So whoever wants an N-rank job that always comes back with a result keeps
# one GPU-FREE supervisor per rank (the process the launcher started), lets it talk to its peers over gloo on CPU
# tensors, and runs everything that touches the GPU in fresh child processes it can stop by their exact PID:
#   1. a preflight child per rank: RCCL init + one all-gather of the rank ids, under a wall limit;
#   2. the worker child per rank, with the backend the preflight earned ("rccl"), else the host-staged fallback ("gloo").
# The multi-process IPC mode of this pool's driver (HSA_ENABLE_IPC_MODE_LEGACY) is the caller's to set; it is honoured,
# recorded, and - only when the preflight fails with it - the other value is tried once before RCCL is given up.
# ---------------------------------------------------------------------------------------------------------------------
IPC_ENV = "HSA_ENABLE_IPC_MODE_LEGACY"
IPC_DEFAULT = "0"          # this pool's host driver only supports dmabuf IPC (environment note); a caller's value wins


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def gloo_env_fixups(env=None) -> None:
    """gloo picks its interface from the host name; a container whose host name does not resolve needs the loopback
    interface named explicitly (single-node jobs only ever talk over 127.0.0.1)."""
    import os
    import socket
    env = os.environ if env is None else env
    if "GLOO_SOCKET_IFNAME" in env:
        return
    try:
        socket.gethostbyname(socket.gethostname())
    except OSError:
        env["GLOO_SOCKET_IFNAME"] = "lo"


def other_ipc_mode(mode: Optional[str]) -> str:
    return "1" if (mode is None or mode == "0") else "0"


def preflight_child_main(stub: Optional[str] = None, limit_s: float = 60.0) -> int:
    """Entry of the preflight CHILD (`python -m fpyv_amd.dist --preflight`): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    from the environment.  Brings the RCCL process group up on this rank's GPU, all-gathers the rank ids, checks them,
    prints one JSON line and leaves; any failure is a traceback on stderr and exit code 1.  `stub` (tests, no GPU):
    "ok" does the same over gloo on CPU tensors, "fail" exits 1 at once, "hang" never comes back."""
    import datetime
    import json
    import os
    import sys
    import time
    t0 = time.monotonic()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if stub == "fail":
        print(f"preflight stub: rank {rank} told to fail", file=sys.stderr)
        return 1
    if stub == "hang":
        time.sleep(3600)
        return 1
    import torch.distributed as dist
    info = {"ok": False, "rank": rank, "world": world, "ipc_mode": os.environ.get(IPC_ENV)}
    if stub:
        gloo_env_fixups()
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=limit_s))
        dev, info["library_version"] = "cpu", f"torch {torch.__version__} gloo (stub)"
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            print(f"preflight: rank {rank} needs GPU {local_rank}, this node shows {torch.cuda.device_count()}", file=sys.stderr)
            return 1
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=limit_s))
        try:
            info["library_version"] = "RCCL/NCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            info["library_version"] = f"unavailable ({type(e).__name__})"
    ids = torch.full((1,), rank, dtype=torch.int64, device=dev)
    got = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got, ids)
    if not stub:
        torch.cuda.synchronize()
    seen = [int(x) for x in got.cpu()]
    if seen != list(range(world)):
        print(f"preflight: all-gather of the rank ids returned {seen}", file=sys.stderr)
        return 1
    info.update(ok=True, rank_ids_gathered=seen, seconds=time.monotonic() - t0)
    print(json.dumps(info), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def stop_child(proc, grace_s: float = 5.0) -> None:
    """Terminate, then kill, exactly the process we started."""
    import subprocess
    if proc.poll() is None:
        proc.terminate()
        try:
            proc.wait(timeout=grace_s)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()


def run_child(cmd, env, limit_s: float, capture_stdout: bool, on_start=None) -> dict:
    """Run one child under a wall limit; stderr goes to a temporary file whose tail is the failure reason.  Returns
    {"rc", "seconds", "timed_out", "stdout", "stderr_tail"}; rc 124 = stopped at the limit (like timeout(1))."""
    import subprocess
    import sys
    import tempfile
    import time
    t0 = time.monotonic()
    with tempfile.TemporaryFile(mode="w+b") as errf:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if capture_stdout else subprocess.DEVNULL, stderr=errf)
        if on_start is not None:
            on_start(proc)
        timed_out, out = False, b""
        try:
            out, _ = proc.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            stop_child(proc)
            try:
                out, _ = proc.communicate(timeout=5)
            except Exception:       # noqa: BLE001
                out = b""
        finally:
            if on_start is not None:
                on_start(None)
        errf.seek(0)
        err = errf.read().decode("utf-8", "replace")
    if err:
        sys.stderr.write(err if len(err) < 20000 else err[-20000:])
        sys.stderr.flush()
    rc = 124 if timed_out else proc.returncode
    return {"rc": rc, "seconds": time.monotonic() - t0, "timed_out": timed_out,
            "stdout": (out or b"").decode("utf-8", "replace"), "stderr_tail": " | ".join(err.strip().splitlines()[-3:])[-600:]}


class RankSupervisor:
    """The GPU-free side of one rank (see the block comment above).  Peers agree over a gloo group on CPU tensors that
    uses the launcher's own MASTER_ADDR / MASTER_PORT; children get fresh ports that rank 0 picks and broadcasts."""

    def __init__(self, timeout_s: float = 1800.0):
        import datetime
        import os
        import signal
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        self._child = None
        gloo_env_fixups()
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))

        def on_term(signum, frame):                  # the launcher stops us: take the child we started along
            if self._child is not None:
                stop_child(self._child, grace_s=3.0)
            os._exit(128 + signum)
        signal.signal(signal.SIGTERM, on_term)
        signal.signal(signal.SIGINT, on_term)

    def track(self, proc) -> None:
        self._child = proc

    def pick_port(self) -> int:
        box = [free_port() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return int(box[0])

    def agree_min(self, value: float) -> float:
        """The smallest of the ranks' values (a wall limit every rank derived from its own clock)."""
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def all_ok(self, ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def gather_reasons(self, reason: Optional[str]) -> List[str]:
        box: List[Optional[str]] = [None] * self.world
        self.dist.all_gather_object(box, reason)
        return [f"rank {r}: {x}" for r, x in enumerate(box) if x]

    def child_env(self, port: int, ipc_mode: Optional[str]) -> dict:
        import os
        env = dict(os.environ, RANK=str(self.rank), LOCAL_RANK=str(self.local_rank), WORLD_SIZE=str(self.world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # a child is NOT a worker of the launcher's elastic agent: with TORCHELASTIC_USE_AGENT_STORE it would look for the
        # agent's store on the fresh port (nobody listens there) instead of hosting its own rendezvous
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            del env[k]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # `python -m fpyv_amd.dist` from any cwd
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        if ipc_mode is None:
            env.pop(IPC_ENV, None)
        else:
            env[IPC_ENV] = ipc_mode
        return env

    def preflight(self, ipc_mode: Optional[str], limit_s: float, stub: Optional[str] = None, python: Optional[str] = None,
                  wall_s: Optional[float] = None) -> dict:
        """One RCCL preflight across all ranks in fresh children; returns {"ok", "ipc_mode", "seconds", "reasons", "info"}.
        `limit_s` is the bring-up limit the child gives its process group; `wall_s` the wall limit of the child itself
        (default: limit_s plus a start-up allowance of up to 15 s) - a caller under a deadline passes what it has left."""
        import json
        import sys
        port = self.pick_port()
        cmd = [python or sys.executable, "-m", "fpyv_amd.dist", "--preflight", "--limit-s", str(limit_s)] + (["--stub", stub] if stub else [])
        wall = float(wall_s) if wall_s is not None else limit_s + min(15.0, max(2.0, limit_s))      # + start-up of the child
        r = run_child(cmd, self.child_env(port, ipc_mode), wall, capture_stdout=True, on_start=self.track)
        ok = r["rc"] == 0
        info = None
        if ok:
            try:
                info = json.loads([ln for ln in r["stdout"].splitlines() if ln.startswith("{")][-1])
            except Exception:       # noqa: BLE001
                ok = False
        why = None if ok else (f"no answer within {limit_s:.0f} s" if r["timed_out"] else f"exit code {r['rc']}: {r['stderr_tail']}")
        all_ok = self.all_ok(ok)
        reasons = self.gather_reasons(why)
        return {"ok": all_ok, "ipc_mode": ipc_mode, "seconds": r["seconds"], "reasons": reasons,
                "library_version": info.get("library_version") if info else None}

    def close(self) -> None:
        try:
            self.dist.barrier()
            self.dist.destroy_process_group()
        except Exception:           # noqa: BLE001
            pass


def choose_backend(limit_s: float = 60.0, stub: Optional[str] = None, log=None) -> dict:
    """For a program that is ITS OWN rank process (examples/sharded_vec_env.py): decide, before this process touches the
    GPU, whether RCCL works across the job - preflight in fresh children with the caller's IPC mode, then once with the
    other one - and return {"backend": "nccl" | "gloo", "ipc_mode", "fallback_reason", "preflight": [...]}.  The default
    process group is left initialised with gloo (CPU control plane); with "nccl" the caller sets os.environ[IPC_ENV] to
    the returned mode BEFORE its first HIP call and creates its data group with `dist.new_group(backend="nccl")`."""
    import os
    sup = RankSupervisor()
    tried, mode = [], os.environ.get(IPC_ENV, IPC_DEFAULT)
    res = sup.preflight(mode, limit_s, stub)
    tried.append(res)
    if not res["ok"]:
        alt = other_ipc_mode(mode)
        res2 = sup.preflight(alt, limit_s, stub)
        tried.append(res2)
        if res2["ok"]:
            mode, res = alt, res2
    out = {"backend": "nccl" if res["ok"] else "gloo", "ipc_mode": mode if res["ok"] else os.environ.get(IPC_ENV),
           "fallback_reason": None if res["ok"] else "RCCL preflight failed: " + "; ".join(tried[0]["reasons"])[:500],
           "preflight": tried}
    if log is not None and sup.rank == 0:
        log(f"collective backend: {out['backend']} (ipc_mode {out['ipc_mode']!r}" + (f"; {out['fallback_reason']}" if out["fallback_reason"] else "") + ")")
    return out


if __name__ == "__main__":
    import argparse
    import sys
    _ap = argparse.ArgumentParser()
    _ap.add_argument("--preflight", action="store_true")
    _ap.add_argument("--stub", default=None, choices=[None, "ok", "fail", "hang"])
    _ap.add_argument("--limit-s", type=float, default=60.0)
    _a = _ap.parse_args()
    if not _a.preflight:
        _ap.error("the only entry point of this module is --preflight")
    sys.exit(preflight_child_main(_a.stub, _a.limit_s))
