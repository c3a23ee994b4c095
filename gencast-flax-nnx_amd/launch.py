"""One process per GPU, without torch: worker spawning and the single-node rendezvous.

The reference runs ensemble members under `jax.pmap` over the local devices
(common/rollout.py:78-176).  Here every member-shard is an ordinary OS process that owns one
`gc_handle` on one GPU; the only thing the processes have to agree on before RCCL can take over is
the 128-byte `ncclUniqueId` (`_lib.comm_unique_id()` on rank 0 -> `NativeDenoiser.comm_init` on every
rank).  That blob travels through a file: all ranks are on one node (xGMI does not leave it).

Two ways to get N ranks:
  * `python bench.py --gpus N` -> `spawn_workers` starts N fresh children (RANK / LOCAL_RANK /
    WORLD_SIZE / GC_RDV_DIR in their environment) from a parent that never touches the GPU;
  * any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (e.g. `python -m torch.distributed.run`):
    the ranks then find each other through a directory named after MASTER_PORT and the launcher's pid.
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
import time
from typing import Callable, List, Optional, Sequence, Tuple


def world_from_env(env=None) -> Tuple[int, int, int]:
  """(rank, local_rank, world_size) as set by `spawn_workers` or torchrun; (0, 0, 1) when unset."""
  env = os.environ if env is None else env
  world = int(env.get("WORLD_SIZE", "1"))
  rank = int(env.get("RANK", "0"))
  local = int(env.get("LOCAL_RANK", str(rank)))
  if not 0 <= rank < world:
    raise ValueError(f"RANK {rank} outside WORLD_SIZE {world}")
  return rank, local, world


def default_rendezvous_dir(env=None) -> str:
  """A directory every rank of ONE launch derives identically: GC_RDV_DIR when the parent set it, else
  <tmp>/gc_rdv_<uid>_<MASTER_PORT>_<run id>_<restart count>_<parent pid> (under torchrun all workers share the
  agent as parent; an elastic restart keeps the agent's pid but bumps TORCHELASTIC_RESTART_COUNT, so the keys of
  the failed attempt -- a stale ncclUniqueId, exit markers -- are never seen by the new one)."""
  env = os.environ if env is None else env
  if env.get("GC_RDV_DIR"):
    return env["GC_RDV_DIR"]
  token = (f"{os.getuid()}_{env.get('MASTER_PORT', '0')}_{env.get('TORCHELASTIC_RUN_ID', 'none')}_"
           f"{env.get('TORCHELASTIC_RESTART_COUNT', '0')}_{os.getppid()}")
  return os.path.join(tempfile.gettempdir(), f"gc_rdv_{token}")


class FileRendezvous:
  """Single-node exchange of small blobs between the ranks of one launch (atomic rename + polling)."""

  def __init__(self, directory: str, rank: int, world_size: int, timeout: float = 300.0):
    self.dir, self.rank, self.world, self.timeout = directory, rank, world_size, timeout
    os.makedirs(self.dir, mode=0o700, exist_ok=True)
    st = os.stat(self.dir)
    if st.st_uid != os.getuid():                           # somebody else made it first: do not trust its contents
      raise PermissionError(f"rendezvous directory {self.dir} belongs to uid {st.st_uid}, not to this user")

  def _path(self, key: str) -> str:
    return os.path.join(self.dir, key)

  def put(self, key: str, blob: bytes) -> None:
    tmp = self._path(f".{key}.{self.rank}.tmp")
    with open(tmp, "wb") as f:
      f.write(blob)
    os.replace(tmp, self._path(key))                       # readers never see a partial file

  def get(self, key: str) -> bytes:
    deadline = time.monotonic() + self.timeout
    p = self._path(key)
    while True:
      try:
        with open(p, "rb") as f:
          return f.read()
      except FileNotFoundError:
        if time.monotonic() > deadline:
          raise TimeoutError(f"rank {self.rank}: nothing at {p} after {self.timeout:.0f} s")
        time.sleep(0.01)

  def broadcast(self, key: str, make: Callable[[], bytes], root: int = 0) -> bytes:
    """`make()` runs on `root` only; every rank returns root's blob."""
    if self.rank == root:
      blob = make()
      self.put(key, blob)
      return blob
    return self.get(key)

  def barrier(self, key: str) -> None:
    self.put(f"{key}.{self.rank}", b"1")
    for r in range(self.world):
      self.get(f"{key}.{r}")

  def finish(self, key: str = "exit", cleanup: bool = True) -> None:
    """Last call of every rank: a barrier, then rank 0 removes the directory -- in TWO phases, because a barrier
    alone is not enough: a rank still polling `<key>.0` would find the directory gone when rank 0 deletes right
    after ITS barrier returns (it then times out and the launch fails).  Every rank therefore acknowledges having
    LEFT the barrier (`<key>_ack.<rank>`), and rank 0 deletes only after all acknowledgements are in; nobody reads
    anything after writing its acknowledgement."""
    self.barrier(key)
    self.put(f"{key}_ack.{self.rank}", b"1")
    if self.rank != 0:
      return
    for r in range(self.world):
      self.get(f"{key}_ack.{r}")
    if cleanup:
      self.cleanup()

  def cleanup(self) -> None:
    """Rank 0 only, and only once no other rank reads the directory any more (`finish`): remove it (best effort)."""
    if self.rank != 0:
      return
    try:
      for name in os.listdir(self.dir):
        os.unlink(self._path(name))
      os.rmdir(self.dir)
    except OSError:
      pass


def init_library_comm(native, rdv: "FileRendezvous", make_unique_id: Callable[[], bytes],
                      timeout: float = 90.0, gpu_tag: Optional[str] = None) -> bool:
  """Collective: sets up the handle's RCCL communicator (`native.comm_init`) on every rank.  Returns True
  only when EVERY rank succeeded; otherwise every rank tears its communicator down again and returns
  False, so the caller can fall back consistently.  `ncclCommInitRank` blocks until all ranks arrive, so
  it runs in a helper thread and a rank that failed early (e.g. librccl missing) cannot hang the others
  for longer than `timeout`.  A rank whose helper thread is still blocked inside RCCL afterwards gets
  `native.comm_stuck = True`: it must not destroy that handle (finish with `os._exit`).

  `gpu_tag`: something that names this rank's GPU (host name + `_lib.device_pci_bus_id`).  When given, the
  ranks compare tags first and nobody enters RCCL if two of them share a GPU (RCCL refuses that, and the
  rank that called `ncclCommInitRank` first would stay blocked inside it, holding runtime locks)."""
  import threading
  native.comm_stuck = False
  native.comm_any_stuck = False
  if gpu_tag is not None:
    rdv.put(f"gpu_tag.{rdv.rank}", gpu_tag.encode())
    tags = [rdv.get(f"gpu_tag.{r}") for r in range(rdv.world)]
    if len(set(tags)) < rdv.world:
      return False
  state = {"ok": False, "err": None}

  def attempt():
    try:
      uid = rdv.broadcast("nccl_unique_id", make_unique_id)
      native.comm_init(uid, rdv.rank, rdv.world)
      state["ok"] = True
    except Exception as e:  # pylint: disable=broad-except
      state["err"] = e
  t = threading.Thread(target=attempt, daemon=True)
  t.start()
  t.join(timeout)
  mine = state["ok"] and not t.is_alive()
  native.comm_stuck = t.is_alive()
  if not mine and rdv.rank == 0 and not os.path.exists(os.path.join(rdv.dir, "nccl_unique_id")):
    rdv.put("nccl_unique_id", b"\0" * 128)              # unblock ranks waiting for an id rank 0 could not make
  # a rank whose helper thread is still inside ncclCommInitRank says so: every rank must then leave at once (a healthy
  # rank that went on would wait for the departed one in its first barrier until the rendezvous timeout)
  rdv.put(f"comm_status.{rdv.rank}", b"ok" if mine else (b"stuck" if native.comm_stuck else repr(state["err"]).encode()))
  statuses = [rdv.get(f"comm_status.{r}") for r in range(rdv.world)]
  everyone = all(s == b"ok" for s in statuses)
  native.comm_any_stuck = any(s == b"stuck" for s in statuses)
  if not everyone and mine:
    try:
      native.comm_destroy()
    except Exception:  # pylint: disable=broad-except
      pass
  return everyone


EXIT_COMM_FALLBACK = 4     # world > 1, but the exchange is not RCCL spanning every rank (and that was not allowed)
EXIT_COMM_STUCK = 5        # this rank's helper thread is still blocked inside ncclCommInitRank


def comm_exit_code(world: int, mode: str, rccl_ranks: int, stuck: bool, allow_host_broadcast: bool) -> int:
  """Exit code a multi-rank benchmark / driver must leave with, given how its exchange came up: 0 only when the
  result it is about to report was produced the way it claims (`mode == "rccl"` over `world` ranks), or when the
  host fallback was asked for explicitly.  A stuck RCCL set-up is never a success, whatever else worked."""
  if stuck:
    return EXIT_COMM_STUCK
  if world > 1 and not allow_host_broadcast and not (mode == "rccl" and rccl_ranks == world):
    return EXIT_COMM_FALLBACK
  return 0


class Exchange:
  """How this launch's one exchange (the conditioning broadcast) runs: `mode` "none" | "rccl" | "host-file",
  the rendezvous, the rank count RCCL itself reports (`gc_comm_info` -> ncclCommCount; 0 without RCCL)."""

  def __init__(self, mode: str, rdv: Optional["FileRendezvous"], rccl_ranks: int, stuck: bool):
    self.mode, self.rdv, self.rccl_ranks, self.stuck = mode, rdv, rccl_ranks, stuck


def open_exchange(native, rank: int, world: int, make_unique_id: Callable[[], bytes], *, gpu_tag: Optional[str] = None,
                  allow_host_broadcast: bool = False, timeout: float = 90.0, force: bool = False,
                  rdv: Optional["FileRendezvous"] = None) -> Exchange:
  """Collective over the ranks of one launch: rendezvous + the handle's RCCL communicator.  Returns how the exchange
  will run; when that is not RCCL over all `world` ranks and `allow_host_broadcast` is off, every rank leaves through
  `close_exchange` with a NON-ZERO code -- a run that silently took the host path must not look like a measured
  multi-GPU result.  `force`: build the (single-rank) communicator even for world == 1 (rehearsal on one GPU)."""
  if world <= 1 and not force:
    return Exchange("none", None, 0, False)
  rdv = rdv or FileRendezvous(default_rendezvous_dir(), rank, world)
  ok = init_library_comm(native, rdv, make_unique_id, timeout=timeout, gpu_tag=gpu_tag)
  stuck = bool(getattr(native, "comm_stuck", False))
  ranks = 0
  if ok:
    try:
      ranks = int(native.comm_info()[0])
    except Exception:  # pylint: disable=broad-except
      ranks = 0
  ex = Exchange("rccl" if ok else "host-file", rdv, ranks, stuck)
  # the code follows the ALL-RANK view: one stuck rank anywhere ends the launch on every rank (EXIT_COMM_STUCK), also
  # with allow_host_broadcast -- the stuck rank cannot take part in the fallback either
  any_stuck = stuck or bool(getattr(native, "comm_any_stuck", False))
  code = comm_exit_code(world, ex.mode, ex.rccl_ranks, any_stuck, allow_host_broadcast)
  if code:
    print(f"[launch rank {rank}] exchange is {ex.mode!r} with {ranks} RCCL rank(s) of {world}"
          f"{', RCCL set-up still blocked on this rank' if stuck else (', RCCL set-up still blocked on another rank' if any_stuck else '')}: refusing to run (exit {code}); "
          "pass allow_host_broadcast / --allow-host-broadcast to time the host fallback", file=sys.stderr)
    close_exchange(ex, code)
  elif ex.mode != "rccl":
    print(f"[launch rank {rank}] RCCL communicator could not be set up on every rank: host broadcast through the "
          "rendezvous directory (NOT the production path; allowed explicitly)", file=sys.stderr)
  return ex


def close_exchange(ex: Exchange, code: int = 0) -> None:
  """Last call of a rank that opened an exchange: two-phase exit barrier + directory removal (`FileRendezvous.finish`),
  then -- if this rank must not return normally (non-zero `code`, or a helper thread still inside RCCL, which would
  block interpreter teardown) -- leave the process with that code."""
  if ex.rdv is not None:
    try:
      ex.rdv.finish()
    except TimeoutError:
      code = code or EXIT_COMM_STUCK
  if ex.stuck:
    code = code or EXIT_COMM_STUCK
  if code:
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)                                         # pylint: disable=protected-access


def spawn_workers(argv: Sequence[str], world_size: int, *, env_extra: Optional[dict] = None,
                  timeout: Optional[float] = None) -> Tuple[int, str]:
  """Starts `world_size` children running `python argv...`, rank r with RANK = LOCAL_RANK = r.

  The caller must not have initialised the GPU (a process that has may neither fork-and-use nor
  exec on this platform; children are fresh interpreters).  Rank 0's stdout is captured and returned;
  the other ranks' stdout goes to this process's stderr.  Returns (worst exit code, rank-0 stdout).
  If one rank fails the others are terminated (by pid)."""
  rdv = tempfile.mkdtemp(prefix="gc_rdv_")
  procs: List[subprocess.Popen] = []
  try:
    for r in range(world_size):
      env = dict(os.environ)
      env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world_size), "GC_RDV_DIR": rdv,
                  "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
      if env_extra:
        env.update({k: str(v) for k, v in env_extra.items()})
      procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                    stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is drained WHILE it runs: read only after exit, a rank 0 that writes more than the pipe buffer
    # (~64 KB: library banners, a verbose worker) would block in write() and never exit
    import threading
    chunks: List[str] = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    codes = [None] * world_size
    while any(c is None for c in codes):
      for r, p in enumerate(procs):
        if codes[r] is None:
          codes[r] = p.poll()
      failed = [c for c in codes if c not in (None, 0)]
      if failed or (deadline is not None and time.monotonic() > deadline):
        for r, p in enumerate(procs):
          if codes[r] is None:
            p.terminate()
        for r, p in enumerate(procs):
          if codes[r] is None:
            try:
              codes[r] = p.wait(timeout=20)
            except subprocess.TimeoutExpired:
              p.kill()
              codes[r] = p.wait()
        if not failed:
          codes = [c if c is not None else 124 for c in codes]
          codes[0] = codes[0] or 124
        break
      time.sleep(0.02)
    reader.join(timeout=20)
    worst = max((abs(c) for c in codes if c is not None), default=0)
    return worst, "".join(chunks)
  finally:
    for p in procs:
      if p.poll() is None:
        p.kill()
    try:
      for name in os.listdir(rdv):
        os.unlink(os.path.join(rdv, name))
      os.rmdir(rdv)
    except OSError:
      pass
