"""The sliver of the TensorFlow-1.x session API that svd_train_val.py drives the hot path
through (svd_train_val.py:20-21,40-57,70-72,120-122,189-198), backed by the HIP library, under the
names the reference spells (``from tfrecomm_amd import graph as tf``):

    tf.int32 / tf.float32, tf.placeholder, tf.train.get_or_create_global_step / get_global_step,
    tf.global_variables_initializer, tf.local_variables_initializer, tf.local_variables, tf.group,
    tf.Session().run(fetches, feed_dict) (+ .graph), tf.train.Saver().save/restore,
    tf.summary.FileWriter(logdir, graph).add_summary(summary, step) with
    summary_pb2.Summary(value=[summary_pb2.Summary.Value(tag=..., simple_value=...)])
    (``from tfrecomm_amd.graph import summary_pb2`` for ``tensorflow.core.framework.summary_pb2``),
    tf.reset_default_graph

There is no graph compiler here: ``ops.inference_svd`` / ``ops.optimization`` record one model
spec in the default graph, and ``Session.run`` maps the fetch list onto one C-ABI call:
``train_op`` in the fetches -> ``tfr_train_step`` (forward + backward + apply, returning the
pre-update logits like TF does for ``sess.run([train_op, logits, infer])``), otherwise
``tfr_forward``.  Handles are opaque; values only exist on the device.
"""
from __future__ import annotations

import json
import os
import time
import types

import numpy as np

from . import _lib as L
from .engine import SvdModel

# dtypes as the reference names them (svd_train_val.py:40-44)
int32, int64, float32, float64 = np.dtype(np.int32), np.dtype(np.int64), np.dtype(np.float32), np.dtype(np.float64)


class Handle(object):
    """An opaque graph node: placeholder, variable, tensor or op."""

    def __init__(self, kind, name=None, dtype=None, graph=None, table=None):
        self.kind, self.name, self.dtype, self.graph, self.table = kind, name, dtype, graph, table

    def __repr__(self):
        return "<tfrecomm %s %r>" % (self.kind, self.name)

    __hash__ = object.__hash__


class Graph(object):
    def __init__(self):
        self.spec = None              # dict: user_num, item_num, dim, loss, item_abs, reg_bias
        self.train = None             # dict: optimizer, adam_mode, lr, reg, frozen mask
        self.placeholders = {}        # role -> Handle
        self.global_step = None
        self.model = None             # SvdModel, created by the first Session that needs it
        self.nodes = {}

    def node(self, kind, name, table=None):
        key = (kind, name)
        if key not in self.nodes:
            self.nodes[key] = Handle(kind, name, graph=self, table=table)
        return self.nodes[key]


_default = Graph()


def get_default_graph():
    return _default


def reset_default_graph():
    global _default
    if _default.model is not None:
        _default.model.close()
    _default = Graph()


def placeholder(dtype, shape=None, name=None):
    """tf.placeholder (svd_train_val.py:40-44).  dtype may be a string or a NumPy dtype."""
    return Handle("placeholder", name, dtype=np.dtype(dtype) if not isinstance(dtype, str) else np.dtype(dtype),
                  graph=_default)


def get_global_step():
    return _default.global_step


def get_or_create_global_step():
    """tf.train.get_or_create_global_step (svd_train_val.py:48); ops.optimization asserts it
    exists (ops.py:119-120)."""
    if _default.global_step is None:
        _default.global_step = Handle("global_step", "global_step", graph=_default)
    return _default.global_step


def local_variables():
    """tf.local_variables() (svd_train_val.py:17-18): this graph has none."""
    return []


def global_variables_initializer():
    return _default.node("init", "init_global")


def local_variables_initializer():
    return _default.node("noop", "init_local")


def group(*ops, **kw):
    g = Handle("group", kw.get("name"), graph=_default)
    g.members = [o for o in ops]
    return g


class Session(object):
    """tf.Session for this one graph.  ``seed`` drives the device-side initialisers."""

    def __init__(self, graph=None, seed=13575, device=0):
        self.graph = graph or _default
        self.seed, self.device = seed, device

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        pass

    # -- model lifetime -------------------------------------------------------------------
    def _model(self):
        g = self.graph
        if g.spec is None:
            raise RuntimeError("no model in the graph: call ops.inference_svd first")
        if g.model is None:
            t = g.train or dict(optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.0, frozen=0)
            s = g.spec
            g.model = SvdModel(s["user_num"], s["item_num"], s["dim"], loss=s["loss"], item_abs=s["item_abs"],
                               reg_bias=s["reg_bias"], optimizer=t["optimizer"], adam_mode=t["adam_mode"],
                               lr=t["lr"], reg=t["reg"], device=self.device)
            if t["frozen"]:
                g.model.set_frozen(t["frozen"])
            g.initialized = False
        return g.model

    @property
    def model(self):
        return self._model()

    # -- run ------------------------------------------------------------------------------
    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        out = self._run(flist, feed_dict or {})
        return out[0] if single else out

    def _feeds(self, feed_dict):
        g = self.graph
        vals = {}
        for role, h in g.placeholders.items():
            for k, v in feed_dict.items():
                if k is h:
                    vals[role] = v
        for k, v in feed_dict.items():                 # the fork re-feeds logits/infer to fetch the cost
            if isinstance(k, Handle) and k.kind in ("logits", "infer"):
                vals["fed_" + k.kind] = v
        return vals

    def _run(self, flist, feed_dict):
        g = self.graph
        flat = []
        for f in flist:
            flat.extend(getattr(f, "members", [f]) if isinstance(f, Handle) and f.kind == "group" else [f])
        kinds = [f.kind if isinstance(f, Handle) else None for f in flist]
        if any(isinstance(f, Handle) and f.kind == "init" for f in flat):
            m = self._model()
            m.init_tables(seed=self.seed)              # ops.py:9-12,29-32 initialisers, on the device
            g.initialized = True
            return [None] * len(flist)
        if all(isinstance(f, Handle) and f.kind in ("noop",) for f in flat):
            return [None] * len(flist)
        m = self._model()
        if not getattr(g, "initialized", False):
            raise RuntimeError("variables are not initialised: run global_variables_initializer() "
                               "(or Saver.restore) first")
        feeds = self._feeds(feed_dict)
        res = {}
        want_train = "train_op" in kinds
        need_fwd = any(k in ("logits", "infer", "cost", "regularizer") for k in kinds)
        if want_train:
            if g.train is None:
                raise RuntimeError("train_op fetched but ops.optimization was never called")
            u, i, r = feeds.get("user"), feeds.get("item"), feeds.get("rate")
            if u is None or i is None or r is None:
                raise ValueError("train_op needs user_batch, item_batch and rate_batch in feed_dict")
            logits, loss, reg = m.train_step(u, i, r, want_logits=True)
            res.update(logits=logits, cost=np.float32(loss), regularizer=np.float32(reg), train_op=None)
        elif need_fwd:
            if "fed_logits" in feeds or "fed_infer" in feeds:
                # svd_train_val.py:94,100: cost re-evaluated from fed logits/infer and rates (host)
                x = np.asarray(feeds.get("fed_logits", feeds.get("fed_infer")), np.float32)
                r = np.asarray(feeds["rate"], np.float32)
                if g.spec["loss"] == "mse":
                    res["cost"] = np.float32(0.5) * np.sum((x - r) ** 2, dtype=np.float32)
                else:
                    res["cost"] = np.sum(np.maximum(x, 0) - x * r + np.log1p(np.exp(-np.abs(x))), dtype=np.float32)
                res["logits"] = x
            else:
                u, i = feeds.get("user"), feeds.get("item")
                if u is None or i is None:
                    raise ValueError("forward fetch needs user_batch and item_batch in feed_dict")
                res["logits"] = m.forward(u, i)
                if "cost" in kinds or "regularizer" in kinds:
                    raise ValueError("cost/regularizer are only available together with train_op "
                                     "(or from fed logits); fetch them in the training run")
        if "logits" in res:
            x = res["logits"]
            if g.spec["loss"] == "mse":
                res["infer"] = x                                  # canonical: infer = logits (README.md:33)
            else:                                                 # ops.py:77-78
                res["infer"] = np.round(1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
        out = []
        for f, k in zip(flist, kinds):
            if k == "variable":
                # the handles ops.inference_svd returns for the embeddings are the GATHERED tensors (ops.py:13-14,37-38:
                # embedding_lookup of the fed ids) - adaptive_test.py:42-44 fetches them with user_batch: range(USER_NUM);
                # without the id feed (or for bias_global) the whole variable comes back.  With train_op in the same run
                # the rows are read after the step.
                tab = m.get_table(f.table)
                ids = feeds.get("user") if f.table in (L.BU, L.P) else feeds.get("item") if f.table in (L.BI, L.Q) else None
                out.append(tab if ids is None else tab[L.as_i32(ids, "ids")])
            elif k == "global_step":
                out.append(m.step)
            elif k in ("train_op", "noop", "group", "init"):
                out.append(None)
            elif k in res:
                out.append(res[k])
            else:
                raise ValueError("cannot fetch %r" % (f,))
        return out


class Saver(object):
    """tf.train.Saver (svd_train_val.py:54,197-198; restored by adaptive_test.py:37-40): the five
    tables, their Adam slots, global_step and the beta powers in one .npz."""

    NAMES = {L.MU: "bias_global", L.BU: "user_bias", L.BI: "item_bias", L.P: "user_features",
             L.Q: "item_features"}

    def save(self, sess, path):
        m = sess.model
        data = {}
        for tid, name in self.NAMES.items():
            data[name] = m.get_table(tid)
            if m.optimizer == "adam":
                data[name + "/Adam"] = m.get_table(tid | L.SLOT_M)
                data[name + "/Adam_1"] = m.get_table(tid | L.SLOT_V)
        step, b1p, b2p = m.get_step()
        data["global_step"] = np.int64(step)
        data["beta1_power"], data["beta2_power"] = np.float32(b1p), np.float32(b2p)
        if not path.endswith(".npz"):
            path = path + ".npz"
        np.savez(path, **data)
        return path

    def restore(self, sess, path):
        if not path.endswith(".npz"):
            path = path + ".npz"
        data = np.load(path, allow_pickle=False)
        m = sess.model
        for tid, name in self.NAMES.items():
            m.set_table(tid, data[name])
            if m.optimizer == "adam" and name + "/Adam" in data.files:
                m.set_table(tid | L.SLOT_M, data[name + "/Adam"])
                m.set_table(tid | L.SLOT_V, data[name + "/Adam_1"])
        m.set_step(int(data["global_step"]), float(data["beta1_power"]), float(data["beta2_power"]))
        sess.graph.initialized = True


# ---- TensorBoard-free stand-ins for the summary calls of svd_train_val.py:20-21,57,189-192 -------------
class _SummaryValue(object):
    def __init__(self, tag=None, simple_value=None):
        self.tag, self.simple_value = tag, simple_value


class _Summary(object):
    """summary_pb2.Summary(value=[summary_pb2.Summary.Value(tag=name, simple_value=val)])"""
    Value = _SummaryValue

    def __init__(self, value=()):
        self.value = list(value)


summary_pb2 = types.SimpleNamespace(Summary=_Summary)


class FileWriter(object):
    """tf.summary.FileWriter(logdir=..., graph=...): one JSON line per scalar in <logdir>/events.jsonl
    ({"wall_time", "step", "tag", "simple_value"}) instead of a TensorBoard event file."""

    def __init__(self, logdir, graph=None, **_):
        self.logdir = logdir
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.jsonl")
        self._fh = open(self.path, "a")
        if graph is not None and getattr(graph, "spec", None):
            self._fh.write(json.dumps(dict(wall_time=time.time(), graph=dict(graph.spec))) + "\n")

    def add_summary(self, summary, global_step=None):
        for v in getattr(summary, "value", []):
            self._fh.write(json.dumps(dict(wall_time=time.time(), step=None if global_step is None else int(global_step),
                                           tag=v.tag, simple_value=None if v.simple_value is None else float(v.simple_value))) + "\n")
        self._fh.flush()

    def flush(self):
        self._fh.flush()

    def close(self):
        self._fh.close()


# the namespaces the reference reaches these through: tf.train.*, tf.summary.*
train = types.SimpleNamespace(get_or_create_global_step=get_or_create_global_step, get_global_step=get_global_step, Saver=Saver)
summary = types.SimpleNamespace(FileWriter=FileWriter, Summary=_Summary)
