"""N > 1 host path on CPU: world_size-2 gloo processes drive EnsembleSampler with a
recording stand-in for the GPU handle.  Checks the one broadcast (rank 0's
conditioning reaches every rank), the member sharding, and that a member's noise
stream does not depend on how members are sharded (reference: per-member RNG +
replicated inputs, common/rollout.py:123-139)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gencast_flax_nnx_amd import EnsembleSampler, member_seed, member_shard
from gencast_flax_nnx_amd.datasets import Dataset, Variable

G, B, C_IN, C_OUT = 12, 1, 5, 2


class FakeNative:
  def __init__(self):
    self.cond = None
    self.noise = None
    self.sample = None
    self.slots = None

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

  # --- what `concurrent_members` needs of a handle ---
  def sync(self):
    pass

  def cond_device_ptr(self):
    return self, self.cond.nbytes                          # the "device pointer" of the fake is the object

  def upload_cond_dev(self, ptr):
    self.cond = np.array(ptr.cond, copy=True)


class FakeDenoiser:
  def __init__(self, rank):
    self.native = FakeNative()
    self.dims = types.SimpleNamespace(c_out=C_OUT)
    self.rank = rank
    self.lanes = []

  def member_lanes(self, count):
    while len(self.lanes) < count:
      self.lanes.append(FakeNative())
    return self.lanes[:count]

  def init_for(self, inputs, template, forcings):
    # only rank 0 holds the real conditioning; other ranks start from garbage
    cond = (np.arange(G * B * C_IN, dtype=np.float32).reshape(G, B, C_IN) if self.rank == 0
            else np.full((G, B, C_IN), -7.0, np.float32))
    return cond, (3, 4), np.arange(C_IN - C_OUT, C_IN, dtype=np.int32)


def _template():
  return Dataset({"a": Variable(("batch", "time", "lat", "lon"), np.zeros((B, 1, 3, 4), np.float32)),
                  "b": Variable(("batch", "time", "lat", "lon"), np.zeros((B, 1, 3, 4), np.float32))},
                 coords=dict(lat=np.arange(3), lon=np.arange(4)))


def _run(rank, world, bcast, members):
  den = FakeDenoiser(rank)
  sampler = types.SimpleNamespace(_denoiser=den, noise_levels=np.array([80.0, 1.0, 0.0]))
  ens = EnsembleSampler(sampler, rank=rank, world_size=world, broadcast_host=bcast, base_seed=11)
  out = ens(None, _template(), None, members)
  return den, out


def _worker(rank, world, port, members, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)

  def bcast(arr, src):
    t = torch.from_numpy(arr)
    dist.broadcast(t, src=src)
    return t.numpy()
  den, out = _run(rank, world, bcast, members)
  q.put((rank, den.native.cond, [(m, {k: v.data for k, v in ds.items()}) for m, ds in out]))
  dist.barrier()
  dist.destroy_process_group()


def test_member_shard_and_seed():
  assert member_shard(8, 0, 8) == [0] and member_shard(8, 7, 8) == [7]
  assert member_shard(5, 1, 2) == [1, 3] and member_shard(5, 0, 2) == [0, 2, 4]
  allm = sorted(sum((member_shard(11, r, 4) for r in range(4)), []))
  assert allm == list(range(11))
  assert member_seed(3, 5) == member_seed(3, 5) and member_seed(3, 5) != member_seed(3, 6)
  with pytest.raises(ValueError):
    member_shard(4, 2, 2)


def test_two_rank_gloo_ensemble_matches_single_process():
  members = 5
  _, ref = _run(0, 1, None, members)
  ref = {m: {k: v.data for k, v in ds.items()} for m, ds in ref}
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, members, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  want_cond = np.arange(G * B * C_IN, dtype=np.float32).reshape(G, B, C_IN)
  seen = {}
  for rank, cond, outs in got:
    np.testing.assert_array_equal(cond, want_cond)        # the broadcast reached this rank
    assert [m for m, _ in outs] == member_shard(members, rank, 2)
    for m, d in outs:
      seen[m] = d
  assert sorted(seen) == list(range(members))
  for m in range(members):
    for k in ref[m]:
      np.testing.assert_array_equal(seen[m][k], ref[m][k])  # sharding does not change a member


def test_concurrent_members_equal_sequential_members():
  """K members in flight on K handles: same members, same results, noise uploaded to the handle that runs it."""
  den1, seq = _run(0, 1, None, 5)
  den = FakeDenoiser(0)
  sampler = types.SimpleNamespace(_denoiser=den, noise_levels=np.array([80.0, 1.0, 0.0]))
  ens = EnsembleSampler(sampler, rank=0, world_size=1, base_seed=11, concurrent_members=3)
  out = ens(None, _template(), None, 5)
  assert [m for m, _ in out] == [m for m, _ in seq] == [0, 1, 2, 3, 4]
  for (_, a), (_, b) in zip(out, seq):
    for k in a.keys():
      np.testing.assert_array_equal(a[k].data, b[k].data)
  assert len(den.lanes) == 2
  for lane in den.lanes:
    np.testing.assert_array_equal(lane.cond, den.native.cond)
    np.testing.assert_array_equal(lane.slots, den.native.slots)
  # fewer members than lanes: no lane is created for nothing
  den2 = FakeDenoiser(0)
  ens2 = EnsembleSampler(types.SimpleNamespace(_denoiser=den2, noise_levels=np.array([80.0, 1.0, 0.0])),
                         rank=0, world_size=1, base_seed=11, concurrent_members=4)
  assert len(ens2(None, _template(), None, 2)) == 2 and len(den2.lanes) == 1
  with pytest.raises(ValueError, match="concurrent_members"):
    EnsembleSampler(sampler, concurrent_members=0)
