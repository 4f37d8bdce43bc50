"""Batching and data loading with the reference's ``dataio`` call shapes.

Host code (NumPy), behaviour-identical to /root/reference/dataio.py for the parts on the hot
path - verified against id/column streams produced by importing the real module
(tests/golden/iter_streams.npz):

* ``ShuffleIterator(inputs, batch_size)`` (dataio.py:94-117): columns are stacked into one
  ``[N, ncols]`` array whose dtype is NumPy's promotion of the column dtypes (int32 ids +
  float32 outcomes -> float64); every ``next()`` draws ``np.random.randint(0, N, (B,))`` from
  the *global legacy* RNG (seeded by the driver, svd_train_val.py:15) - sampling WITH
  replacement - and returns the fancy-indexed columns.
* ``OneEpochIterator`` (dataio.py:120-138): ``np.array_split`` chunks of ``arange(N)`` (or the
  whole set when ``batch_size <= 0``); raises ``StopIteration`` and rewinds.

Extension for the HBM-resident feed: ``next_ids()`` draws the same ``randint`` stream but
returns only the row numbers, so the device gathers the triples itself
(``Session.run(train_op, feed_dict={resident_ids: ...})``).
"""
from __future__ import annotations

import math
import os

import numpy as np


class _ColumnTable(object):
    """Equal-length 1-D columns held as one row-major table (what dataio.py:103 builds)."""

    def __init__(self, inputs):
        cols = [np.asarray(c) for c in inputs]
        if not cols:
            raise ValueError("need at least one column")
        n = len(cols[0])
        if any(c.ndim != 1 or len(c) != n for c in cols):
            raise ValueError("columns must be 1-D and of equal length")
        dtype = np.result_type(*cols)                  # vstack's promotion rule
        table = np.empty((n, len(cols)), dtype=dtype)
        for c, col in enumerate(cols):
            table[:, c] = col
        self.num_cols = len(cols)
        self.len = n
        self.inputs = table

    def _rows(self, idx):
        picked = self.inputs[idx, :]
        return [picked[:, c] for c in range(self.num_cols)]


class ShuffleIterator(_ColumnTable):
    """Randomly generate batches (uniform, with replacement) - dataio.py:94-117."""

    def __init__(self, inputs, batch_size=10):
        super(ShuffleIterator, self).__init__(inputs)
        self.batch_size = batch_size

    def __len__(self):
        return self.len

    def __iter__(self):
        return self

    def __next__(self):
        return self.next()

    def next_ids(self):
        """The row numbers of the next batch: the identical RNG draw ``next()`` makes."""
        return np.random.randint(0, self.len, (self.batch_size,))

    def next(self):
        return self._rows(self.next_ids())


class OneEpochIterator(ShuffleIterator):
    """Sequentially generate one-epoch batches, typically for test data - dataio.py:120-138."""

    def __init__(self, inputs, batch_size=10):
        super(OneEpochIterator, self).__init__(inputs, batch_size=batch_size)
        if batch_size > 0:
            nchunks = math.ceil(self.len / batch_size)
            self.idx_group = np.array_split(np.arange(self.len), nchunks)
        else:
            self.idx_group = [np.arange(self.len)]
        self.group_id = 0

    def next_ids(self):
        if self.group_id >= len(self.idx_group):
            self.group_id = 0                          # rewinds, so it can be iterated every epoch
            raise StopIteration
        idx = self.idx_group[self.group_id]
        self.group_id += 1
        return idx


# ---------------------------------------------------------------------------- files
COL_NAMES = ["user", "item", "outcome", "wins", "fails"]      # dataio.py:40


def build_paths(dataset_name, data_folder="data"):
    """data/<name>/{train,test,val}.csv, config.yml, qmatrix.npz - dataio.py:8-16."""
    folder = os.path.join(data_folder, dataset_name)
    return (folder, os.path.join(folder, "train.csv"), os.path.join(folder, "test.csv"),
            os.path.join(folder, "val.csv"), os.path.join(folder, "config.yml"),
            os.path.join(folder, "qmatrix.npz"))


def get_config(config_file):
    """Per-dataset YAML (keys USER_NUM, ITEM_NUM, NB_CLASSES, BATCH_SIZE) - dataio.py:31-35.
    safe_load: the reference's bare ``yaml.load(f)`` no longer runs on PyYAML >= 6."""
    import yaml
    with open(config_file) as f:
        return yaml.safe_load(f)


def read_process(filename, sep="\t"):
    """Header-less CSV ``user,item,outcome,wins,fails`` -> DataFrame with int32 ids and a float32
    outcome - dataio.py:38-46.  Files with only three columns load too (wins/fails = NaN)."""
    import pandas as pd
    df = pd.read_csv(filename, sep=sep, header=None, names=COL_NAMES, engine="python")
    for col in ("user", "item"):
        df[col] = df[col].astype(np.int32)
    df["outcome"] = df["outcome"].astype(np.float32)
    return df


def get_data(dataset_name, data_folder="data"):
    """(train, val, test) DataFrames of a prepared dataset - dataio.py:49-54."""
    _, csv_train, csv_test, csv_val, _, _ = build_paths(dataset_name, data_folder)
    return read_process(csv_train, sep=","), read_process(csv_val, sep=","), read_process(csv_test, sep=",")


def build_new_paths(dataset_name, data_folder="data"):
    """data/<name>/{all.csv, config.yml, qmatrix.npz, skill_wins.npz, skill_fails.npz} - the single-file layout of the FM
    experiments (dataio.py:19-28, read by fm.py:34 and sharedtask.py:24)."""
    folder = os.path.join(data_folder, dataset_name)
    return (folder,) + tuple(os.path.join(folder, f) for f in ("all.csv", "config.yml", "qmatrix.npz", "skill_wins.npz", "skill_fails.npz"))


def get_new_data(dataset_name, data_folder="data"):
    """the whole dataset as one frame - dataio.py:57-60 (fm.py:41)."""
    return read_process(build_new_paths(dataset_name, data_folder)[1], sep=",")


_AGENTS = ("users", "items", "skills", "attempts", "wins", "fails", "item_wins", "item_fails", "extra")   # dataio.py:67
_MODEL_NAMES = (({"users", "items"}, False, "IRT: "), ({"users", "items"}, True, "MIRTb: "),
                ({"skills", "attempts"}, False, "AFM: "), ({"skills", "wins", "fails"}, False, "PFA: "))


def get_legend(experiment_args):
    """Labels of an FM experiment from its switches (dataio.py:63-87): (short code, full legend, LaTeX legend, active blocks).
    The short code takes one letter per active block - W / F for the item_wins / item_fails blocks - followed by the
    dimension; the legends name the classical model the blocks amount to (IRT, MIRTb, AFM, PFA) where there is one."""
    dim = experiment_args["d"]
    active = [a for a in _AGENTS if experiment_args.get(a)]
    letters = [("W" if "_w" in a else "F") if "_" in a else a[0] for a in active]
    prefix = ""
    for blocks, with_dim, name in _MODEL_NAMES:
        if set(active) == blocks and (dim > 0) == with_dim and (with_dim or dim == 0):
            prefix = name
    latex = prefix + ", ".join(active)
    return "".join(letters) + str(dim), latex + " d = {:d}".format(dim), latex, active


def read_movielens(filename):
    """MovieLens ``user::item::rating::timestamp`` with 1-based ids (README.md:18-25) -> the same
    frame layout with 0-based ids."""
    import pandas as pd
    df = pd.read_csv(filename, sep="::", header=None, names=["user", "item", "outcome", "st"], engine="python")
    df["user"] = (df["user"] - 1).astype(np.int32)
    df["item"] = (df["item"] - 1).astype(np.int32)
    df["outcome"] = df["outcome"].astype(np.float32)
    df["wins"] = 0.0
    df["fails"] = 0.0
    return df[COL_NAMES]


def prepare_folder(path):
    if not os.path.isdir(path):
        os.makedirs(path)
