"""Synthetic ERA5-shaped inputs for the nano GenCast task (no dataset access needed).

Shapes follow the reference pipeline (training/era5_dataset.py:531-561,
585-789): inputs carry 2 time steps of 4 surface + 6x13 atmospheric + 4 progress +
2 (time-broadcast) static variables = 176 channels; forcings carry the 4 progress
variables of the target step; targets are 4 surface + 6x13 atmospheric = 82
channels.  C_in = 176 + 4 + 82 = 262.
"""
from __future__ import annotations

import numpy as np

from . import config as cfg
from .datasets import Dataset, Variable


def grid_2p5deg():
  """lat -90..90 step 2.5 (73), lon 0..357.5 (144)."""
  return np.arange(-90.0, 90.0 + 1e-9, 2.5), np.arange(0.0, 360.0, 2.5)


def grid_1deg():
  """lat -90..90 step 1 (181), lon 0..359 (360)."""
  return np.arange(-90.0, 90.0 + 1e-9, 1.0), np.arange(0.0, 360.0, 1.0)


def make_example(lat=None, lon=None, batch: int = 1, seed: int = 0,
                 task: cfg.TaskConfig = cfg.TASK):
  """Returns (inputs, targets_template, forcings) Datasets with N(0,1) data."""
  if lat is None or lon is None:
    lat, lon = grid_2p5deg()
  rng = np.random.default_rng(seed)
  nlat, nlon, nlev = len(lat), len(lon), len(task.pressure_levels)
  coords = dict(lat=np.asarray(lat, np.float32), lon=np.asarray(lon, np.float32),
                level=np.asarray(task.pressure_levels))
  f32 = lambda *s: rng.standard_normal(s).astype(np.float32)

  def field(name, nt):
    if name in cfg.ALL_ATMOSPHERIC_VARS:
      return Variable(("batch", "time", "level", "lat", "lon"), f32(batch, nt, nlev, nlat, nlon))
    if name.startswith("year_progress"):
      return Variable(("batch", "time"), f32(batch, nt))
    if name.startswith("day_progress"):
      return Variable(("batch", "time", "lon"), f32(batch, nt, nlon))
    return Variable(("batch", "time", "lat", "lon"), f32(batch, nt, nlat, nlon))

  inputs = Dataset({n: field(n, 2) for n in task.input_variables}, coords)
  forcings = Dataset({n: field(n, 1) for n in task.forcing_variables}, coords)
  targets = Dataset({n: field(n, 1) for n in task.target_variables}, coords)
  return inputs, targets, forcings
