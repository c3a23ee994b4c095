"""The C-ABI library loads on a CPU-only box and exports every symbol the headers
declare; argument validation that needs no GPU; loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from gencast_flax_nnx_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
  text = open(os.path.join(ROOT, "include", header)).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(gc_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
  lib = _lib.load_library()
  names = sorted(set(_declared("gencast_hip.h") + _declared("gencast_hip_debug.h")))
  assert len(names) >= 30
  for n in names:
    assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
  assert sorted(_lib.SIGNATURES) == names
  assert lib.gc_abi_version() == 1
  assert b"gfx950" in lib.gc_build_info()


def test_kernel_class_table():
  lib = _lib.load_library()
  n = lib.gc_num_kernel_classes()
  names = [lib.gc_kernel_class_name(i).decode() for i in range(n)]
  assert n == 13 and len(set(names)) == n and all(x.startswith("gc_") for x in names)


def test_config_struct_matches_header_layout():
  assert ctypes.sizeof(_lib.GcConfig) == 11 * 4
  assert ctypes.sizeof(_lib.GcSampleStats) == 8


def _cfg(**kw):
  base = dict(latent_size=128, d_model=128, num_heads=2, ffw_hidden=256, num_layers=1, c_in=20, c_out=6)
  base.update(kw)
  return base


@pytest.mark.parametrize("kw,exc", [
    (dict(latent_size=0), ValueError), (dict(c_out=30), ValueError), (dict(num_heads=3), ValueError),
    (dict(latent_size=256, d_model=128), ValueError), (dict(latent_size=96, d_model=96), ValueError),
    (dict(ffw_hidden=100), ValueError), (dict(num_heads=16), ValueError)])
def test_create_rejects_bad_configs_without_touching_a_gpu(kw, exc):
  with pytest.raises(exc):
    _lib.NativeDenoiser(**_cfg(**kw))


def test_no_cpu_fallback():
  """Without a HIP device the product path must fail loudly, not compute on the CPU."""
  if _lib.device_count() > 0:
    pytest.skip("a GPU is visible")
  with pytest.raises(_lib.GencastHipError, match="no HIP device"):
    _lib.NativeDenoiser(**_cfg())


def test_missing_library_is_an_error(tmp_path):
  with pytest.raises(_lib.GencastHipError, match="no CPU fallback"):
    _lib.load_library(str(tmp_path / "nope.so"))


def test_product_does_not_import_the_oracle():
  pkg = os.path.join(ROOT, "gencast-flax-nnx_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".cpp", ".h")):
        src = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in src and "from oracle" not in src and "gencast_oracle" not in src, f


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
  """include/gencast_hip.h compiles as C99 (no C++ / torch types in the signatures) and a C program linked
  against the shared library resolves every declared symbol and can call the GPU-free entry points."""
  import shutil
  import subprocess
  if not shutil.which("gcc"):
    pytest.skip("gcc not available")
  names = _declared("gencast_hip.h")
  src = tmp_path / "abi.c"
  refs = "\n".join(f"  p[{i}] = (fn){n};" for i, n in enumerate(names))
  src.write_text(f"""
#include <stdio.h>
#include "gencast_hip.h"
typedef void (*fn)(void);
int main(void) {{
  fn p[{len(names)}];
{refs}
  gc_config cfg = {{128, 128, 2, 256, 1, 20, 6, 1, 32, 32, 16.0f}};
  gc_handle* h = 0;
  int bad = gc_create(0, 0, &h);                 /* null config: invalid argument, no GPU touched */
  cfg.latent_size = 96; cfg.d_model = 96;
  int unsup = gc_create(&cfg, 0, &h);            /* unsupported width: rejected before any HIP call */
  printf("%d %s %d %d %d %d\\n", gc_abi_version(), gc_build_info(), gc_num_kernel_classes(), bad, unsup, p[0] != 0);
  return 0;
}}
""")
  exe = tmp_path / "abi"
  libdir = os.path.dirname(_lib.LIB_PATH)
  subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                  "-L", libdir, "-lgencast_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
  out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
  assert out[0] == "1" and out[-3:] == ["1", "5", "1"] and "gfx950" in " ".join(out)


def test_device_queries_without_a_gpu():
  """gc_device_pci_bus_id refuses an index beyond gc_device_count (0 here) instead of touching the runtime."""
  from gencast_flax_nnx_amd import _lib
  if _lib.device_count() == 0:
    with pytest.raises(_lib.GencastHipError):
      _lib.device_pci_bus_id(0)


def test_library_was_built_from_the_sources_it_sits_beside():
  """csrc/build.sh compiles the hash of csrc/*.{hip,cpp,h,inc} into gc_build_info(); an edit or a checkout without a
  rebuild would otherwise be tested, profiled and benchmarked under the wrong tree's name."""
  import os
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  assert bench.library_source_hash() == bench.source_hash(), (
      "libgencast_hip.so is stale: run gencast-flax-nnx_amd/csrc/build.sh (or __graft_entry__.build())")


def test_bench_quotes_profile_figures_only_for_the_tree_they_were_measured_on():
  """`roofline.traffic` / `mfma_busy` / `inter_kernel_gaps` come from committed rocprofv3 passes; they may
  only be quoted while profiles/profile_meta.json carries the hash of THIS tree's kernel and host sources."""
  import json
  import os
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  import bench
  meta = json.load(open(os.path.join(root, "profiles", "profile_meta.json")))
  fig = bench.profile_figures("gc_gemm_ffw1")
  if meta["source_hash"] == bench.source_hash() and bench.library_source_hash() == bench.source_hash():
    assert fig["from_profile"]["used"] is True and fig["from_profile"]["tag"] == meta["tag"]
    assert fig["traffic"] > 1e6 and 0.0 < fig["mfma_busy"] < 1.0 and fig["inter_kernel_gaps"]["gap_avg_us"] > 0
  else:
    assert fig["from_profile"]["used"] is False
    assert fig["traffic"] is None and fig["mfma_busy"] is None and fig["inter_kernel_gaps"] is None
