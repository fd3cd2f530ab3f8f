#!/usr/bin/env python3
"""Writes tests/golden/sklearn022_like/training_*.pkl: KernelDensity pickles laid out the way scikit-learn 0.22.1 + joblib wrote the
pre-trained NanoSim models (README.md:41; requirements.txt:13) — WITHOUT scikit-learn 0.22, which this image cannot install:

  * class paths of 0.22: sklearn.neighbors._kde.KernelDensity, sklearn.neighbors._kd_tree.KDTree / newObj,
    sklearn.neighbors._dist_metrics.EuclideanDistance / newObj;
  * the estimator's state is its attribute dict with `bandwidth` (no `bandwidth_`: that name appeared in 1.2) and `_sklearn_version`;
  * the tree pickles through __reduce__ -> (newObj, (KDTree,), state tuple) with the 0.22 tuple layout (training matrix first);
  * pickle protocol 2, numpy arrays through joblib's NumpyArrayWrapper.

The classes below are stand-ins that exist only while this script runs (registered under the sklearn module names so that the pickle
carries those names).  nanosim_amd.model._kde_pickle_tolerant must read the files without scikit-learn (tests/test_host.py)."""
import os
import sys
import types

import joblib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "sklearn022_like")


def _module(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


for pkg in ("sklearn", "sklearn.neighbors"):
    if pkg not in sys.modules or not isinstance(sys.modules[pkg], types.ModuleType) or getattr(sys.modules[pkg], "__file__", None):
        _module(pkg)
kd, dm, kde_mod = _module("sklearn.neighbors._kd_tree"), _module("sklearn.neighbors._dist_metrics"), _module("sklearn.neighbors._kde")


def _new_obj(cls):
    return cls.__new__(cls)


class EuclideanDistance:
    def __reduce__(self):
        return (dm.newObj, (EuclideanDistance,), (2.0, np.zeros(1), np.zeros((1, 1))))      # 0.22: (p, vec, mat)

    def __setstate__(self, st):
        self.state = st


class KDTree:
    def __init__(self, data, leaf_size=40):
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.leaf_size = leaf_size

    def __reduce__(self):
        n = len(self.data)
        idx = np.arange(n, dtype=np.intp)
        node_data = np.zeros(1, dtype=[("idx_start", np.intp), ("idx_end", np.intp), ("is_leaf", np.intp), ("radius", np.float64)])
        node_data["idx_end"] = n
        node_data["is_leaf"] = 1
        bounds = np.stack([self.data.min(0, keepdims=True), self.data.max(0, keepdims=True)])
        # 0.22's BinaryTree.__getstate__: (data_arr, idx_array_arr, node_data_arr, node_bounds_arr, leaf_size, n_levels, n_nodes, n_trims,
        #                                  n_leaves, n_splits, n_calls, dist_metric, sample_weight)
        return (kd.newObj, (KDTree,), (self.data, idx, node_data, bounds, int(self.leaf_size), 1, 1, 0, 1, 0, 0, EuclideanDistance(), None))

    def __setstate__(self, st):
        self.state = st


class KernelDensity:
    def __init__(self, data, bandwidth):
        self.algorithm, self.atol, self.bandwidth, self.breadth_first = "auto", 0, float(bandwidth), True
        self.kernel, self.leaf_size, self.metric, self.metric_params, self.rtol = "gaussian", 40, "euclidean", None, 0
        self.tree_ = KDTree(data, 40)
        self._sklearn_version = "0.22.1"


for cls, mod in ((EuclideanDistance, dm), (KDTree, kd), (KernelDensity, kde_mod)):
    cls.__module__ = mod.__name__
    setattr(mod, cls.__name__, cls)
for mod in (kd, dm):
    _new_obj.__module__ = mod.__name__
    f = types.FunctionType(_new_obj.__code__, _new_obj.__globals__, "newObj")
    f.__module__, f.__qualname__ = mod.__name__, "newObj"
    mod.newObj = f

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(22)
    tables = {
        "aligned_region": (rng.lognormal(np.log(7000), 0.6, 300)[:, None], 10.0),
        "aligned_reads": (rng.lognormal(np.log(7600), 0.6, 300)[:, None], 10.0),
        "unaligned_length": (rng.lognormal(np.log(1500), 0.9, 200)[:, None], 10.0),
        "ht_length": (np.log10(rng.lognormal(np.log(40), 0.8, 250) + 1)[:, None], 0.01),
        "ht_ratio": (rng.beta(2, 2, 250)[:, None], 0.01),
        "gap_length": (np.log10(rng.lognormal(np.log(20), 1.0, 150) + 1)[:, None], 0.01),
        "aligned_region_2d": (np.stack([rng.lognormal(np.log(1400), 0.6, 300), rng.lognormal(np.log(900), 0.5, 300)], axis=1), 10.0),
    }
    expect = {}
    for name, (data, bw) in tables.items():
        joblib.dump(KernelDensity(data, bw), os.path.join(OUT, "training_%s.pkl" % name), protocol=2)
        expect[name + "_data"], expect[name + "_bw"] = data, bw
    np.savez(os.path.join(OUT, "expected.npz"), **expect)
    print("written:", sorted(os.listdir(OUT)))
