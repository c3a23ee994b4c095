#!/usr/bin/env python3
"""Reference orbax checkpoint -> the flat .npz weight file of this build.

    python tools/orbax_to_flat.py <checkpoint_step_N dir> out.npz [--latent 256 --layers 16 --heads 4
                                   --ffw 2048 --c-in 262 --c-out 82]

Run it where `orbax-checkpoint` is installed (the reference's environment; this build's image has
no orbax and no network).  It restores the raw PyTree the reference saved
(training/train_helpers.py:338-358: `PyTreeCheckpointer().save(path, other_state)`), applies the
reference's own clean-up (`clean_state`, training/evaluation.py:137-176, restated in
gencast-flax-nnx_amd/weights.py) and writes {NNX path: float32 array} -- exactly what
`weights.load_params` / `Denoiser(params=...)` / `gc_load_weight` take.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def to_plain(tree):
  """orbax / flax containers -> dicts, lists and numpy arrays."""
  if hasattr(tree, "items"):
    return {k: to_plain(v) for k, v in tree.items()}
  if isinstance(tree, (list, tuple)):
    return [to_plain(v) for v in tree]
  return np.asarray(tree)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("checkpoint")
  ap.add_argument("out")
  ap.add_argument("--latent", type=int, default=256)
  ap.add_argument("--layers", type=int, default=16)
  ap.add_argument("--heads", type=int, default=4)
  ap.add_argument("--ffw", type=int, default=2048)
  ap.add_argument("--c-in", type=int, default=262)
  ap.add_argument("--c-out", type=int, default=82)
  ap.add_argument("--lenient", action="store_true", help="skip missing / mis-shaped parameters instead of failing")
  args = ap.parse_args()
  import orbax.checkpoint as ocp   # only here: absent from the build image
  from gencast_flax_nnx_amd import weights
  raw = ocp.PyTreeCheckpointer().restore(os.path.abspath(os.path.expanduser(args.checkpoint)))
  dims = weights.ModelDims(c_in=args.c_in, c_out=args.c_out, latent=args.latent, d_model=args.latent,
                           num_heads=args.heads, ffw_hidden=args.ffw, num_layers=args.layers)
  flat = weights.import_reference_state(to_plain(raw), dims, strict=not args.lenient)
  weights.save_params(args.out, flat)
  n = sum(int(v.size) for v in flat.values())
  print(f"wrote {args.out}: {len(flat)} arrays, {n / 1e6:.2f} M parameters")


if __name__ == "__main__":
  main()
