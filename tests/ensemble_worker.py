"""Worker of tests/test_launch.py: one rank of a torch-free ensemble run on a CPU-only box.

Started by `launch.spawn_workers` (RANK / LOCAL_RANK / WORLD_SIZE / GC_RDV_DIR in the environment).
The GPU handle is replaced by a recording stand-in whose `comm_*` methods move bytes through the
rendezvous directory; everything else -- `world_from_env`, the unique-id hand-off, `EnsembleSampler`
with `library_comm=True`, member sharding and seeding -- is the product code bench.py runs."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gencast_flax_nnx_amd import EnsembleSampler, launch  # noqa: E402
from gencast_flax_nnx_amd.datasets import Dataset, Variable  # noqa: E402

G, B, C_IN, C_OUT = 12, 1, 5, 2


class FileCommNative:
  """FakeNative of test_ensemble_gloo.py + the gc_comm_* surface over FileRendezvous."""

  def __init__(self, rdv):
    self.rdv, self.cond, self.noise, self.sample, self.comm = rdv, None, None, None, None
    self.n_bcast = 0

  def comm_init(self, uid, rank, world):
    assert len(uid) == 128
    self.comm = (bytes(uid), rank, world)

  def comm_broadcast_cond(self, root=0):
    assert self.comm is not None, "comm_init first"
    key = f"cond{self.n_bcast}"
    blob = self.rdv.broadcast(key, lambda: self.cond.tobytes(), root)
    self.cond = np.frombuffer(blob, np.float32).reshape(G, B, C_IN).copy()
    self.n_bcast += 1

  def set_noisy_slots(self, s):
    self.slots = np.asarray(s)

  def upload_cond(self, c):
    self.cond = np.array(c, copy=True)

  def upload_noise(self, z):
    self.noise = np.array(z, copy=True)

  def sample_resident(self, sigmas, skip_dead_call=True, want_stats=True):
    self.sample = self.noise * float(sigmas[0]) + self.cond[..., :C_OUT]

  def download_sample(self):
    return self.sample


def main():
  members = int(sys.argv[1])
  rank, local_rank, world = launch.world_from_env()
  rdv = launch.FileRendezvous(launch.default_rendezvous_dir(), rank, world, timeout=60)
  uid = rdv.broadcast("uid", lambda: bytes(range(128)))
  native = FileCommNative(rdv)
  native.comm_init(uid, rank, world)

  class Den:
    dims = types.SimpleNamespace(c_out=C_OUT)

    def init_for(self, inputs, template, forcings):
      cond = (np.arange(G * B * C_IN, dtype=np.float32).reshape(G, B, C_IN) if rank == 0
              else np.full((G, B, C_IN), -7.0, np.float32))       # only rank 0 holds the real conditioning
      return cond, (3, 4), np.arange(C_IN - C_OUT, C_IN, dtype=np.int32)
  den = Den()
  den.native = native
  sampler = types.SimpleNamespace(_denoiser=den, noise_levels=np.array([80.0, 1.0, 0.0]))
  tmpl = Dataset({"a": Variable(("batch", "time", "lat", "lon"), np.zeros((B, 1, 3, 4), np.float32)),
                  "b": Variable(("batch", "time", "lat", "lon"), np.zeros((B, 1, 3, 4), np.float32))},
                 coords=dict(lat=np.arange(3), lon=np.arange(4)))
  ens = EnsembleSampler(sampler, rank=rank, world_size=world, base_seed=11, library_comm=True)
  out = ens(None, tmpl, None, members)
  np.savez(os.path.join(rdv.dir, f"result{rank}.npz"), cond=native.cond,
           **{f"m{m}_{k}": v.data for m, ds in out for k, v in ds.items()})
  rdv.barrier("done")
  if rank == 0:
    print(json.dumps({"world": world, "local_rank": local_rank, "members": [m for m, _ in out],
                      "rdv": rdv.dir}))


if __name__ == "__main__":
  main()
