"""``ops.inference_svd`` / ``ops.optimization`` with the reference's call shapes, recording the
model in the default graph instead of building TensorFlow ops (reference: ops.py:6-91,118-153).

Two call styles bind:

* north-star / canonical (README.md:31-39)::

      infer, regularizer = inference_svd(user_batch, item_batch, user_num=U, item_num=I, dim=D)
      cost, train_op = optimization(infer, regularizer, rate_batch, learning_rate=lr, reg=lam)

* the fork's live signature (ops.py:6,118; svd_train_val.py:47,50)::

      infer, logits, regularizer, user_bias, user_features, item_bias, item_features = \\
          inference_svd(user_batch, item_batch, wins_batch, fails_batch, user_num=U, item_num=I, dim=D)
      cost_nll, train_op = optimization(infer, logits, regularizer, rate_batch, learning_rate=lr, reg=lam)

  ``wins_batch`` / ``fails_batch`` are accepted and ignored, as the live graph ignores them.

The arithmetic variant is chosen by keywords (defaults = the canonical model BASELINE.json
names): ``loss="mse"|"nll"`` (ops.py:124 | 125-126), ``item_abs`` (ops.py:44),
``reg_bias`` (ops.py:85-89), and in ``optimization``: ``optimizer="adam"|"sgd"``
(ops.py:144 | 145), ``adam_mode="tf1"|"lazy"``.  ``fork_semantics=True`` selects the fork's
live combination (nll, |item|, bias l2, SGD) in one switch.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import graph as G


def inference_svd(user_batch, item_batch, *args, **kw):
    """Variables + forward of ops.py:6-91.  Returns handles: ``(infer, regularizer)`` or, when
    called with the fork's ``wins_batch, fails_batch`` positionals, the fork's 7-tuple."""
    names = ("user_num", "item_num", "dim", "device")
    fork_style = False
    rest = list(args)
    if len(rest) >= 2 and all(isinstance(a, G.Handle) or a is None for a in rest[:2]):
        fork_style = True                              # wins_batch, fails_batch: unused by the live graph
        rest = rest[2:]
    for n, v in zip(names, rest):
        if n in kw:
            raise TypeError("inference_svd() got multiple values for %r" % n)
        kw[n] = v
    kw.pop("wins_batch", None)
    kw.pop("fails_batch", None)
    user_num, item_num = int(kw.pop("user_num")), int(kw.pop("item_num"))
    dim = int(kw.pop("dim", 5))
    kw.pop("device", None)                             # tf.device strings (ops.py:7,43): one GPU here
    fork_sem = bool(kw.pop("fork_semantics", False))
    loss = kw.pop("loss", "nll" if fork_sem else "mse")
    item_abs = bool(kw.pop("item_abs", fork_sem))
    reg_bias = bool(kw.pop("reg_bias", fork_sem))
    if kw:
        raise TypeError("inference_svd() got unexpected arguments %s" % sorted(kw))
    if loss not in L.LOSS:
        raise ValueError("loss must be 'mse' or 'nll'")
    g = G.get_default_graph()
    if g.spec is not None:
        raise RuntimeError("the default graph already holds a model; call graph.reset_default_graph()")
    g.spec = dict(user_num=user_num, item_num=item_num, dim=dim, loss=loss, item_abs=item_abs,
                  reg_bias=reg_bias, fork_default=fork_sem)
    g.placeholders["user"] = user_batch
    g.placeholders["item"] = item_batch
    infer, logits, regularizer = g.node("infer", "infer"), g.node("logits", "logits"), g.node("regularizer", "regularizer")
    if not fork_style:
        return infer, regularizer
    return (infer, logits, regularizer,
            g.node("variable", "user_bias", L.BU), g.node("variable", "user_features", L.P),
            g.node("variable", "item_bias", L.BI), g.node("variable", "item_features", L.Q))


def variables():
    """Handles of the five trainables (ops.py:8-12,29-32) for ``var_list`` / fetching."""
    g = G.get_default_graph()
    return dict(bias_global=g.node("variable", "bias_global", L.MU), user_bias=g.node("variable", "user_bias", L.BU),
                item_bias=g.node("variable", "item_bias", L.BI), user_features=g.node("variable", "user_features", L.P),
                item_features=g.node("variable", "item_features", L.Q))


def optimization(infer, *args, **kw):
    """Loss + ``Optimizer.minimize`` of ops.py:118-153.  Returns ``(cost, train_op)`` handles;
    ``cost`` is the data term only (ops.py:152-153)."""
    g = G.get_default_graph()
    assert G.get_global_step() is not None             # ops.py:119-120
    rest = list(args)
    if rest and isinstance(rest[0], G.Handle) and rest[0].kind == "logits":
        rest = rest[1:]                                # fork style passes logits second (ops.py:118)
    names = ("regularizer", "rate_batch", "learning_rate", "reg", "device", "var_list")
    for n, v in zip(names, rest):
        if n in kw:
            raise TypeError("optimization() got multiple values for %r" % n)
        kw[n] = v
    if g.spec is None:
        raise RuntimeError("call inference_svd before optimization")
    rate_batch = kw.pop("rate_batch")
    lr = float(kw.pop("learning_rate"))
    lam = float(kw.pop("reg"))
    kw.pop("regularizer", None)
    kw.pop("device", None)
    var_list = kw.pop("var_list", None)
    optimizer = kw.pop("optimizer", "sgd" if g.spec.get("fork_default") else "adam")
    adam_mode = kw.pop("adam_mode", "tf1")
    if kw:
        raise TypeError("optimization() got unexpected arguments %s" % sorted(kw))
    if optimizer not in L.OPTIMIZER or adam_mode not in L.ADAM_MODE:
        raise ValueError("optimizer must be 'adam'|'sgd', adam_mode 'tf1'|'lazy'")
    frozen = 0
    if var_list is not None:                           # ops.py:146-149; adaptive_test.py:28
        keep = set()
        for v in var_list:
            if not (isinstance(v, G.Handle) and v.kind == "variable"):
                raise TypeError("var_list entries must be variable handles")
            keep.add(v.table)
        frozen = sum(1 << t for t in (L.MU, L.BU, L.BI, L.P, L.Q) if t not in keep)
    g.train = dict(optimizer=optimizer, adam_mode=adam_mode, lr=lr, reg=lam, frozen=frozen)
    g.placeholders["rate"] = rate_batch
    return g.node("cost", "cost"), g.node("train_op", "train_op")


def sigmoid(x):
    """ops.py:94-95 - NumPy helper the reference's callers use (svd_train_val.py:95)."""
    return 1 / (1 + np.exp(-x))
