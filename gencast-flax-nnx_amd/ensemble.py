"""Ensemble-member sampling sharded one process per GPU.

Members are independent samples of the same (inputs, forcings) with different
noise (reference: per-member RNG + replicated inputs, common/rollout.py:123-139,
312-322; results pulled per device, :357-360).  Sharding: member m runs on rank
m % world_size; the only exchange is ONE broadcast of the packed conditioning
[G,B,C_in] from rank 0 -- `gc_comm_broadcast_cond`: RCCL over xGMI, issued by the
library itself on the handle's stream (no torch) once `NativeDenoiser.comm_init`
has run; there is no collective inside the denoiser or the sampler.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from . import datasets
from .denoiser import Denoiser
from .sampler import Sampler


def member_shard(num_members: int, rank: int, world_size: int) -> List[int]:
  """Members handled by `rank` (round-robin so every rank gets ceil or floor)."""
  if not 0 <= rank < world_size:
    raise ValueError("rank out of range")
  return list(range(rank, num_members, world_size))


def member_seed(base_seed: int, member: int) -> int:
  """Noise stream of a member: independent of how members are sharded."""
  return int(np.random.SeedSequence([int(base_seed), int(member)]).generate_state(1)[0])


class EnsembleSampler:
  """Runs this rank's share of an ensemble.

  library_comm=True : the handle's own RCCL communicator (`native.comm_init` done by the caller)
      broadcasts the resident conditioning in place -- the production path.
  broadcast_host(array, src) -> array : broadcasts a host float32 array in place
      (e.g. gloo on CPU-only test boxes); used when no device broadcast is given.
  concurrent_members=K : this rank keeps K of its members in flight at once, each on its own library handle
      (own HIP stream, `Denoiser.member_lanes`); the conditioning reaches the extra handles by a
      device-to-device copy after the exchange.  Every member's result is bit-identical to K = 1.
  """

  def __init__(self, sampler: Sampler, rank: int = 0, world_size: int = 1,
               broadcast_host: Optional[Callable] = None, base_seed: int = 0,
               library_comm: bool = False, concurrent_members: int = 1):
    self._sampler = sampler
    self._denoiser: Denoiser = sampler._denoiser  # pylint: disable=protected-access
    self.rank, self.world_size = rank, world_size
    self._bh = broadcast_host
    self._library_comm = library_comm
    self.base_seed = base_seed
    if concurrent_members < 1:
      raise ValueError("concurrent_members must be >= 1")
    self.concurrent_members = int(concurrent_members)

  def member_noise(self, member: int, shape, template) -> np.ndarray:
    """Initial noise of one member: the SAME generator as `Sampler.__call__` (isotropic spherical
    white noise on an equiangular grid, dpm_solver_plus_plus_2s.py:71-78; `noise_kind` honoured),
    seeded by (base_seed, member) only."""
    gen = np.random.default_rng(member_seed(self.base_seed, member))
    draw = getattr(self._sampler, "draw_noise", None)
    if draw is None:                                       # bare stand-ins in tests
      return gen.standard_normal(shape, dtype=np.float32)
    return np.asarray(draw(gen, shape, template), np.float32)

  def __call__(self, inputs, targets_template, forcings, num_members: int
               ) -> List[Tuple[int, datasets.Dataset]]:
    template = datasets.as_dataset(targets_template)
    # every rank packs its (possibly stale) local copy to size buffers; rank 0's data wins
    cond, grid_shape, slots = self._denoiser.init_for(inputs, template, forcings)
    native = self._denoiser.native
    native.set_noisy_slots(slots)
    if self.world_size > 1 and self._library_comm:
      if self.rank == 0:
        native.upload_cond(cond)
      native.comm_broadcast_cond(0)
    else:
      if self.world_size > 1 and self._bh is not None:
        cond = self._bh(np.ascontiguousarray(cond, dtype=np.float32), 0)
      native.upload_cond(cond)
    sigmas = np.asarray(self._sampler.noise_levels, dtype=np.float32)
    shape = (cond.shape[0], cond.shape[1], self._denoiser.dims.c_out)
    mine = member_shard(num_members, self.rank, self.world_size)
    lanes = [native]
    k = min(self.concurrent_members, len(mine))
    if k > 1:
      lanes += list(self._denoiser.member_lanes(k - 1))
      native.sync()                                        # the conditioning is complete on lane 0's stream
      ptr, _ = native.cond_device_ptr()
      for lane in lanes[1:]:
        lane.set_noisy_slots(slots)
        lane.upload_cond_dev(ptr)                          # device-to-device, on the lane's own stream
    out = []
    for g0 in range(0, len(mine), len(lanes)):
      group = mine[g0:g0 + len(lanes)]
      for lane, m in zip(lanes, group):                    # enqueue only: no host synchronisation in here
        lane.upload_noise(self.member_noise(m, shape, template))
        lane.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
      for lane, m in zip(lanes, group):
        out.append((m, datasets.like_inputs(Denoiser.unpack_outputs(lane.download_sample(), grid_shape, template),
                                            targets_template, inputs, forcings)))
    return out
