"""In-kernel timeline of the attention kernel at the nano size (diagnostic library only).

    bash gencast-flax-nnx_amd/csrc/build.sh stamps && python tools/stamp_attention.py [1deg]

Loads libgencast_hip_stamps.so (-DGC_STAMPS), runs one denoiser call and prints, per phase, the median
over waves of the s_memtime deltas (shader cycles) of the LAST attention launch."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gencast_flax_nnx_amd import _lib  # noqa: E402
from tests import helpers  # noqa: E402

lib = _lib.load_library(os.path.join(ROOT, "gencast-flax-nnx_amd", "csrc", "libgencast_hip_stamps.so"))
_lib._lib = lib
lib.gc_debug_attention_stamps.restype = ctypes.c_int
lib.gc_debug_attention_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64]

from gencast_flax_nnx_amd.sampler import noise_schedule  # noqa: E402

gr, dims, params, x, sigma = helpers.one_degree_setup() if "1deg" in sys.argv else helpers.nano_setup()
nd = helpers.make_native(gr, dims, params, 1)
sigmas = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
nd.set_noisy_slots(np.arange(dims.c_in - dims.c_out, dims.c_in, dtype=np.int32))
for feat in ("f32", "f16"):
  nd.set_option("features", feat)
  nd.denoise(x, sigma)
  nd.upload_cond(x)                                        # (gc_denoise overwrote the resident conditioning)
  nd.upload_noise(np.random.default_rng(0).standard_normal((x.shape[0], 1, dims.c_out), dtype=np.float32))
  words = 8192 * 12
  assert lib.gc_debug_attention_stamps(nd._h, None, -words) == 0
  for _ in range(3):                                       # 117 calls back to back: the clock the sampler runs at
    nd.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
  buf = np.zeros(words, np.uint64)
  rc = lib.gc_debug_attention_stamps(nd._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), words)
  assert rc == 0, rc
  st = buf.reshape(-1, 12).astype(np.int64)
  st = st[st[:, 0] > 0]
  ticks = st[:, 11] >> 16                                   # s_memrealtime (100 MHz) over the wave's life
  st[:, 11] &= 0xFFFF
  names = ["q load+split+park, idx issue", "idx barrier", "first K/V issue + V staged", "(loop) QK^T incl. K wait", "(loop) softmax",
           "(loop) P split + tr reads + PV issue", "(loop) next V staged", "epilogue: partial stores issue", "store drain"]
  d = [st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4], st[:, 5], st[:, 6], st[:, 7],
       st[:, 9] - st[:, 8], st[:, 10] - st[:, 9]]
  life = st[:, 10] - st[:, 0]
  print(f"   shader clock seen by the waves: {float(np.median(life / np.maximum(ticks, 1))) * 100:.0f} MHz")
  print(f"== features {feat}: waves {len(st)}, chunks per wave median {int(np.median(st[:, 11]))}; wave lifetime median "
        f"{int(np.median(life))} p90 {int(np.percentile(life, 90))} cycles")
  for n, v in zip(names, d):
    print(f"  {n:40s} median {int(np.median(v)):7d}  p90 {int(np.percentile(v, 90)):7d}")
  per = (st[:, 4] + st[:, 5] + st[:, 6] + st[:, 7]) / np.maximum(st[:, 11], 1)
  print(f"  per chunk: {int(np.median(per))} cycles")
  # work-item lists (1 degree): whole tiles and key-range pieces are different populations
  big = 0.5 * np.percentile(st[:, 11], 95)                # whole tiles: 12-15 chunks at k_hop 8, pieces 3-4
  for label, sel in (("whole tiles", st[:, 11] > big), ("pieces", st[:, 11] <= big)):
    if sel.sum() and (~sel).sum():
      fixed = (life - st[:, 4] - st[:, 5] - st[:, 6] - st[:, 7])[sel]
      print(f"  {label}: waves {int(sel.sum())}, chunks median {int(np.median(st[sel, 11]))}, lifetime median {int(np.median(life[sel]))} cycles, "
            f"outside the chunk loop median {int(np.median(fixed))}, per chunk {int(np.median(per[sel]))}")
nd.close()
