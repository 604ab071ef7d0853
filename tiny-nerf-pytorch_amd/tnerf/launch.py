"""One process per GPU without an external launcher.

`python bench.py --gpus N` / `python src/train.py --gpus N` from a plain shell: the parent process — BEFORE it makes
any GPU call (a process that has initialised HIP must not fork/exec workers) — starts N fresh children of the same
command line with the torch.distributed environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free
MASTER_PORT), relays rank 0's stdout and waits for all of them.  Under `python -m torch.distributed.run` the
environment is already there and nothing is spawned.  No torch import in this module.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence

ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def under_launcher(env: Optional[Dict[str, str]] = None) -> bool:
    """True when a launcher (torchrun, or spawn_ranks below) has already set this process's rank."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of rank `rank` of `world` single-node ranks."""
    if not (0 <= rank < world) or not (0 < port < 65536):
        raise ValueError(f"rank_env: rank={rank} world={world} port={port}")
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNERF_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    return env


def read_env(expect_world: Optional[int] = None):
    """(rank, local_rank, world) of this process; world must equal expect_world when given."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if expect_world is not None and world != expect_world:
        raise SystemExit(f"--gpus {expect_world} but WORLD_SIZE={world}: start it as `python <script> --gpus {expect_world}` "
                         f"(self-spawning) or under torch.distributed.run --nproc-per-node {expect_world}")
    return rank, local, world


def spawn_ranks(world: int, argv: Sequence[str], port: Optional[int] = None, timeout: Optional[float] = None,
                python: Optional[str] = None, extra_env: Optional[Dict[str, str]] = None) -> int:
    """Run `python argv...` as `world` rank processes; rank 0 inherits stdout, every rank inherits stderr.
    Returns the first non-zero exit code (the other ranks are terminated), else 0."""
    if world < 1:
        raise ValueError("spawn_ranks: world must be >= 1")
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if "rocprof" in preload:
        # a profiler's preloaded tool library has initialised the GPU in THIS process before main() ran: starting ranks from here
        # is the fork/exec-after-HIP-init this module exists to avoid (on this pool it takes the node down)
        raise RuntimeError("spawn_ranks: this process runs under rocprofv3 (preloaded tool library); profile one rank directly "
                           "(`rocprofv3 ... -- python bench.py --gpus 1`) or start the ranks with torch.distributed.run outside the profiler")
    port = free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    try:
        for r in range(world):
            env = rank_env(r, world, port)
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([python or sys.executable, *argv], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        deadline = None if timeout is None else time.time() + timeout
        rc = 0
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.time() > deadline):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
