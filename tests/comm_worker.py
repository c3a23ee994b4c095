"""Worker of tests/test_launch.py: one rank going through `launch.open_exchange` / `close_exchange` -- the code path
bench.py uses to bring up (and verify) its one exchange -- on a stand-in for the GPU handle.

    comm_worker.py ok|fail1|stuck1 [allow]

ok: every rank's communicator comes up (the stand-in reports WORLD_SIZE ranks); fail1: rank 1's comm_init raises;
stuck1: rank 1's comm_init never returns (what ncclCommInitRank does when RCCL refuses the set-up on another rank);
allow: allow_host_broadcast=True."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gencast_flax_nnx_amd import launch  # noqa: E402


class Native:
  def __init__(self, rank, world, case):
    self.rank, self.world, self.case, self.up = rank, world, case, False

  def comm_init(self, uid, rank, world):
    if self.rank == 1 and self.case == "fail1":
      raise RuntimeError("no RCCL here")
    if self.rank == 1 and self.case == "stuck1":
      threading.Event().wait()                                 # never returns
    self.up = True

  def comm_info(self):
    return (self.world, self.rank) if self.up else (0, -1)

  def comm_destroy(self):
    self.up = False


def main():
  case, allow = sys.argv[1], "allow" in sys.argv[2:]
  rank, _, world = launch.world_from_env()
  ex = launch.open_exchange(Native(rank, world, case), rank, world, lambda: bytes(128), gpu_tag=f"host/gpu{rank}",
                            allow_host_broadcast=allow, timeout=3.0,
                            rdv=launch.FileRendezvous(launch.default_rendezvous_dir(), rank, world, timeout=60))
  # what a benchmark does between opening and closing: a collective step through the rendezvous.  (without
  # it the worker hid that a healthy rank, told to go on beside a stuck one, waits here for the departed rank.)
  ex.rdv.barrier("work")
  if rank == 0:
    print(json.dumps({"mode": ex.mode, "rccl_ranks": ex.rccl_ranks, "world": world}))
  launch.close_exchange(ex)


if __name__ == "__main__":
  main()
