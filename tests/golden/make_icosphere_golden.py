"""Generates tests/golden/icosphere.npz from the REFERENCE's own mesh module.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_icosphere_golden.py

`common/icosahedral_mesh.py` is one of the three reference modules that import
with numpy/scipy alone.  The fixture holds, for splits 0..5, the
float32 vertices and int32 faces of `get_last_triangular_mesh_for_sphere`, plus
`faces_to_edges` of every level.  These are data (inputs/outputs), not source.
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from common import icosahedral_mesh as ref  # noqa: E402

out = {}
for s in range(6):
  m = ref.get_last_triangular_mesh_for_sphere(splits=s)
  out[f"vertices_{s}"] = m.vertices
  out[f"faces_{s}"] = m.faces
  snd, rcv = ref.faces_to_edges(m.faces)
  out[f"senders_{s}"] = snd.astype(np.int32)
  out[f"receivers_{s}"] = rcv.astype(np.int32)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icosphere.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
