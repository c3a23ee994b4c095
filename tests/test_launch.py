"""Torch-free N > 1 host path (CPU): `launch.spawn_workers` + `FileRendezvous` + `EnsembleSampler` with the
library-side broadcast (`library_comm=True`), on a stand-in for the GPU handle.  The RCCL call itself
(`gc_comm_broadcast_cond`) needs GPUs and is exercised by `bench.py --gpus N` on the GPU node."""
import json
import os
import sys
import threading
import types

import numpy as np
import pytest

from gencast_flax_nnx_amd import EnsembleSampler, Sampler, launch, member_seed, member_shard
from gencast_flax_nnx_amd.datasets import Dataset, Variable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_from_env():
  assert launch.world_from_env({}) == (0, 0, 1)
  assert launch.world_from_env({"RANK": "3", "WORLD_SIZE": "8"}) == (3, 3, 8)
  assert launch.world_from_env({"RANK": "3", "LOCAL_RANK": "1", "WORLD_SIZE": "8"}) == (3, 1, 8)
  with pytest.raises(ValueError):
    launch.world_from_env({"RANK": "2", "WORLD_SIZE": "2"})
  assert launch.default_rendezvous_dir({"GC_RDV_DIR": "/x/y"}) == "/x/y"
  a = launch.default_rendezvous_dir({"MASTER_PORT": "29500"})
  assert a == launch.default_rendezvous_dir({"MASTER_PORT": "29500"}) and "29500" in a
  assert a != launch.default_rendezvous_dir({"MASTER_PORT": "29501"})


def test_file_rendezvous_between_threads(tmp_path):
  got = {}

  def run(rank):
    rdv = launch.FileRendezvous(str(tmp_path), rank, 3, timeout=20)
    got[rank] = rdv.broadcast("uid", lambda: b"\x01" * 128)
    rdv.barrier("b")
  ts = [threading.Thread(target=run, args=(r,)) for r in (2, 1, 0)]
  for t in ts:
    t.start()
  for t in ts:
    t.join(30)
  assert got == {0: b"\x01" * 128, 1: b"\x01" * 128, 2: b"\x01" * 128}
  with pytest.raises(TimeoutError):
    launch.FileRendezvous(str(tmp_path), 1, 2, timeout=0.05).get("never")


def test_spawned_two_rank_ensemble_matches_single_process(tmp_path):
  members = 5
  worker = os.path.join(ROOT, "tests", "ensemble_worker.py")
  keep = str(tmp_path / "rdv")
  code, out = launch.spawn_workers([worker, str(members)], 2, env_extra={"GC_RDV_DIR": keep}, timeout=120)
  assert code == 0, out
  line = json.loads(out.strip().splitlines()[-1])
  assert line["world"] == 2 and line["members"] == member_shard(members, 0, 2)
  res = [np.load(os.path.join(keep, f"result{r}.npz")) for r in range(2)]
  want_cond = np.arange(12 * 1 * 5, dtype=np.float32).reshape(12, 1, 5)
  for r in res:
    np.testing.assert_array_equal(r["cond"], want_cond)              # the broadcast reached every rank
  code1, out1 = launch.spawn_workers([worker, str(members)], 1, env_extra={"GC_RDV_DIR": str(tmp_path / "one")},
                                     timeout=120)
  assert code1 == 0
  ref = np.load(os.path.join(str(tmp_path / "one"), "result0.npz"))
  seen = {}
  for r in res:
    for k in r.files:
      if k != "cond":
        seen[k] = r[k]
  assert sorted(seen) == sorted(k for k in ref.files if k != "cond")
  for k, v in seen.items():
    np.testing.assert_array_equal(v, ref[k])                         # sharding does not change a member


def test_finish_lets_rank0_delete_only_after_every_rank_has_left(tmp_path):
  """rank 0 used to delete the rendezvous right after ITS barrier returned; a rank still polling
  `exit.0` then found nothing and ran into the timeout.  With rank 0 arriving LAST (the usual case: it builds the
  JSON line first) every other rank is asleep in its poll when the barrier completes."""
  d = str(tmp_path / "rdv")
  errors = []

  def run(rank):
    try:
      rdv = launch.FileRendezvous(d, rank, 3, timeout=5)
      if rank == 0:
        import time
        time.sleep(0.3)
      rdv.finish()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((rank, repr(e)))
  for _ in range(5):
    ts = [threading.Thread(target=run, args=(r,)) for r in (1, 2, 0)]
    for t in ts:
      t.start()
    for t in ts:
      t.join(30)
    assert not errors, errors
    assert not os.path.exists(d)                                       # and the directory is gone afterwards


def test_comm_exit_code_table():
  f = launch.comm_exit_code
  assert f(1, "none", 0, False, False) == 0
  assert f(8, "rccl", 8, False, False) == 0
  assert f(8, "rccl", 4, False, False) == launch.EXIT_COMM_FALLBACK    # RCCL sees fewer ranks than the launch has
  assert f(2, "host-file", 0, False, False) == launch.EXIT_COMM_FALLBACK
  assert f(2, "host-file", 0, False, True) == 0                        # the fallback was asked for
  assert f(2, "rccl", 2, True, False) == launch.EXIT_COMM_STUCK
  assert f(2, "host-file", 0, True, True) == launch.EXIT_COMM_STUCK    # a stuck set-up is never a success


@pytest.mark.parametrize("case,allow,ok", [("ok", False, True), ("fail1", False, False), ("fail1", True, True),
                                           ("stuck1", False, False), ("stuck1", True, False)])
def test_a_launch_whose_exchange_is_not_rccl_over_all_ranks_exits_nonzero(tmp_path, case, allow, ok):
  """a stuck rank used to leave with os._exit(0) and a host-file run printed a normal line.  Now the
  launch fails unless RCCL spans every rank -- or the host fallback was requested AND nothing is stuck."""
  worker = os.path.join(ROOT, "tests", "comm_worker.py")
  code, out = launch.spawn_workers([worker, case] + (["allow"] if allow else []), 2,
                                   env_extra={"GC_RDV_DIR": str(tmp_path / "rdv")}, timeout=120)
  assert (code == 0) == ok, (code, out)
  if ok:
    line = json.loads(out.strip().splitlines()[-1])
    assert line["mode"] == ("rccl" if case == "ok" else "host-file")
    assert line["rccl_ranks"] == (2 if case == "ok" else 0)


def test_a_stuck_rank_ends_the_launch_on_every_rank_at_once(tmp_path):
  """`stuck` is part of the all-rank consensus -- with allow_host_broadcast the healthy rank used to get
  code 0, walk into its first barrier and wait for the departed rank until the rendezvous timeout (60 s here)."""
  import time
  worker = os.path.join(ROOT, "tests", "comm_worker.py")
  t0 = time.monotonic()
  code, out = launch.spawn_workers([worker, "stuck1", "allow"], 2, env_extra={"GC_RDV_DIR": str(tmp_path / "rdv")}, timeout=120)
  assert code != 0 and out.strip() == "" and time.monotonic() - t0 < 40, (code, out, time.monotonic() - t0)


def test_eight_ranks_open_and_close_the_exchange(tmp_path):
  """the world-8 path of BASELINE configs[2] / [4] on the stand-in handle -- rendezvous of the
  128-byte id, GPU-tag comparison, comm_status consensus, one collective step, two-phase exit -- with 8 processes."""
  worker = os.path.join(ROOT, "tests", "comm_worker.py")
  code, out = launch.spawn_workers([worker, "ok"], 8, env_extra={"GC_RDV_DIR": str(tmp_path / "rdv8")}, timeout=180)
  assert code == 0, (code, out)
  line = json.loads(out.strip().splitlines()[-1])
  assert line == {"mode": "rccl", "rccl_ranks": 8, "world": 8}
  assert not os.path.exists(str(tmp_path / "rdv8"))          # rank 0 removed the rendezvous after every acknowledgement


def test_rank0_stdout_larger_than_the_pipe_buffer_does_not_block():
  code, out = launch.spawn_workers(["-c", "import os,sys; sys.stdout.write('x' * 300000) if os.environ['RANK']=='0' else None"],
                                   2, timeout=60)
  assert code == 0 and len(out) == 300000


def test_spawn_reports_a_failing_rank():
  code, _ = launch.spawn_workers(["-c", "import os,sys,time; sys.exit(3) if os.environ['RANK']=='1' else time.sleep(30)"],
                                 2, timeout=60)
  assert code != 0


def test_ensemble_member_noise_is_the_samplers_spherical_noise():
  """members start from the same distribution as Sampler / full_sampling (isotropic spherical
  white noise on an equiangular grid), seeded by (base_seed, member) only."""
  lat, lon = np.linspace(-90, 90, 9), np.arange(16) * 22.5
  tmpl = Dataset({"a": Variable(("batch", "time", "lat", "lon"), np.zeros((1, 1, 9, 16), np.float32))},
                 coords=dict(lat=lat, lon=lon))
  den = types.SimpleNamespace(native=None, dims=types.SimpleNamespace(c_out=1))
  sampler = Sampler(den, 80.0, 0.03, 4, 7.0, 0.0, 0.05, 50.0, 1.0)
  ens = EnsembleSampler(sampler, rank=0, world_size=1, base_seed=5)
  shape = (9 * 16, 1, 1)
  z = ens.member_noise(3, shape, tmpl)
  want = sampler.draw_noise(np.random.default_rng(member_seed(5, 3)), shape, tmpl)
  np.testing.assert_array_equal(z, want)
  white = np.random.default_rng(member_seed(5, 3)).standard_normal(shape, dtype=np.float32)
  assert not np.allclose(z, white)                                   # spherical, not per-node white
  pole = z.reshape(9, 16)[0]
  assert np.allclose(pole, pole[0], atol=1e-5)                       # one value at the pole: a field on the sphere
  sampler.noise_kind = "white"
  np.testing.assert_array_equal(ens.member_noise(3, shape, tmpl), white)


def test_init_library_comm_reaches_consensus(tmp_path):
  """Every rank returns the same verdict: all ok -> True; one rank failing -> everybody False and the ones
  that had succeeded tear their communicator down."""
  for bad_rank in (None, 1, 0):
    d = str(tmp_path / f"case_{bad_rank}")
    out, destroyed = {}, []

    class Native:
      def __init__(self, rank):
        self.rank = rank

      def comm_init(self, uid, rank, world):
        assert len(uid) == 128
        if self.rank == bad_rank and bad_rank != 0:
          raise RuntimeError("no RCCL here")

      def comm_destroy(self):
        destroyed.append(self.rank)

    def make_id():
      if bad_rank == 0:
        raise RuntimeError("ncclGetUniqueId failed")
      return bytes(128)

    def run(rank):
      rdv = launch.FileRendezvous(d, rank, 3, timeout=20)
      out[rank] = launch.init_library_comm(Native(rank), rdv, make_id, timeout=10)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in ts:
      t.start()
    for t in ts:
      t.join(60)
    want = bad_rank is None
    assert out == {0: want, 1: want, 2: want}, (bad_rank, out)
    if bad_rank == 1:
      assert sorted(destroyed) == [0, 2]


def test_ranks_sharing_a_gpu_skip_the_collective_setup(tmp_path):
  """Two ranks that name the same GPU never enter comm_init (RCCL would refuse, and block the first caller)."""
  import threading

  class Native:
    def __init__(self):
      self.calls = 0

    def comm_init(self, uid, rank, world):
      self.calls += 1

    def comm_destroy(self):
      pass

  results, natives = {}, [Native(), Native()]

  def run(rank, tag):
    rdv = launch.FileRendezvous(str(tmp_path / tag[0]), rank, 2, timeout=20)
    results[(tag[0], rank)] = launch.init_library_comm(natives[rank], rdv, lambda: b"\1" * 128, timeout=10, gpu_tag=tag[1][rank])

  for case in (("same", ["host/0000:c1:00.0", "host/0000:c1:00.0"]), ("distinct", ["host/0000:c1:00.0", "host/0000:c5:00.0"])):
    ts = [threading.Thread(target=run, args=(r, case)) for r in range(2)]
    for t in ts:
      t.start()
    for t in ts:
      t.join()
  assert results[("same", 0)] is False and results[("same", 1)] is False
  assert results[("distinct", 0)] is True and results[("distinct", 1)] is True
  assert natives[0].calls == 1 and natives[1].calls == 1          # only the "distinct" case reached comm_init
