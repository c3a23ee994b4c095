"""`NaNCleaner`: predictor wrapper that fills NaNs of one variable before the wrapped predictor sees
them and (optionally) puts them back into the predictions.

Mirrors gencast/nan_cleaning.py:27-156 (`__call__`, `full_sampling`; `loss` is training-only).  With
this repo's TASK the cleaned variable (`sea_surface_temperature`, training/train_helpers.py:170-178)
is absent, so the wrapper is a pass-through at run time -- it exists so that the reference's model
stack `NaNCleaner(InputsAndResiduals(GenCast))` can be assembled unchanged.  It also keeps NaNs
(land points of SST) away from the f16x3 domain guard, which would otherwise re-run every call in f32.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import datasets
from .datasets import Dataset, Variable


class NaNCleaner:

  def __init__(self, predictor, var_to_clean: str, fill_value, reintroduce_nans: bool = False):
    self.predictor = predictor
    fv = datasets.as_dataset(fill_value)[var_to_clean]
    self._fill_value = fv
    self._var_to_clean = var_to_clean
    self._reintroduce_nans = reintroduce_nans

  def _clean(self, dataset: Dataset) -> Dataset:
    """nan_cleaning.py:52-58: `data_array.fillna(fill_value)` with xarray broadcasting by dim name."""
    var = dataset[self._var_to_clean]
    fill = np.asarray(self._fill_value.data, dtype=var.data.dtype)
    shape = [1] * len(var.dims)
    for d, n in zip(self._fill_value.dims, fill.shape):
      if d not in var.dims:
        raise ValueError(f"fill value dimension {d!r} is not a dimension of {self._var_to_clean!r}")
      shape[var.dims.index(d)] = n
    order = sorted(range(len(self._fill_value.dims)), key=lambda i: var.dims.index(self._fill_value.dims[i]))
    fill = np.transpose(fill, order).reshape(shape) if fill.ndim else fill
    data = np.where(np.isnan(var.data), fill, var.data)
    return dataset.assign(Dataset({self._var_to_clean: Variable(var.dims, data)}))

  def _maybe_reintroduce_nans(self, stale_inputs: Dataset, predictions: Dataset) -> Dataset:
    """nan_cleaning.py:60-69: NaN where ANY input frame was NaN."""
    if self._var_to_clean in predictions.keys() and self._var_to_clean in stale_inputs.keys():
      src = stale_inputs[self._var_to_clean]
      mask = np.isnan(src.data)
      dims = list(src.dims)
      if "time" in dims:
        mask = mask.any(axis=dims.index("time"))
        dims.remove("time")
      pred = predictions[self._var_to_clean]
      shape = [1] * len(pred.dims)
      for d, n in zip(dims, mask.shape):
        shape[pred.dims.index(d)] = n
      order = sorted(range(len(dims)), key=lambda i: pred.dims.index(dims[i]))
      m = np.transpose(mask, order).reshape(shape)
      predictions = predictions.assign(Dataset(
          {self._var_to_clean: Variable(pred.dims, np.where(m, np.nan, pred.data).astype(pred.data.dtype))}))
    return predictions

  def _wrap(self, fn, inputs, targets_template, forcings, **kwargs):
    given = (targets_template, inputs, forcings)            # xarray in -> xarray out (datasets.like_inputs)
    inputs = datasets.as_dataset(inputs)
    forcings = None if forcings is None else datasets.as_dataset(forcings)
    original = inputs if self._reintroduce_nans else None
    if self._var_to_clean in inputs.keys():
      inputs = self._clean(inputs)
    if forcings is not None and self._var_to_clean in forcings.keys():
      forcings = self._clean(forcings)
    preds = datasets.as_dataset(fn(inputs, targets_template, forcings, **kwargs))
    if self._reintroduce_nans:
      preds = self._maybe_reintroduce_nans(original, preds)
    return datasets.like_inputs(preds, *given)

  def __call__(self, inputs, targets_template, forcings=None, **kwargs):
    return self._wrap(self.predictor, inputs, targets_template, forcings, **kwargs)

  def full_sampling(self, inputs, targets_template, forcings: Optional[Dataset] = None, **kwargs):
    return self._wrap(self.predictor.full_sampling, inputs, targets_template, forcings, **kwargs)

  def loss(self, *args, **kwargs):
    raise NotImplementedError("training (loss) is outside the sampling hot path")

  loss_and_predictions = loss
