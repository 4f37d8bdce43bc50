"""The per-user fine-tuning drivers on the HIP path (reference: adaptive_test.py, non_adaptive_test.py).

Both reference scripts restore a trained model (`saver.restore`, adaptive_test.py:37-40), freeze everything but the user
tables (`var_list=[user_bias, user_features]`, :28) and then, for one user at a time, alternate

    predict one (user, item)        sess.run(logits, ...)                         adaptive_test.py:100-102
    EPOCH_MAX x sess.run(train_op)  on the items this user has been asked so far  adaptive_test.py:104-116

adaptive_test.py asks `BUDGET` items per user chosen by a selector from cats.py (:76,87); non_adaptive_test.py walks the
test frame in order, predicts each row before training on it and scores accuracy / AUC of those predictions at the end
(:56-121).  Here the EPOCH_MAX inner loop is ONE library call (`tfr_train_steps_repeat`: the tiny batch is uploaded once,
the steps run back to back on the device) instead of EPOCH_MAX host round trips.

The scripts are written against the ordinal variant of the graph (ten return values, `logits_cdf`, `thresholds`), which
ops.py at HEAD does not build; what is kept is what they do with the live binary / regression model.
`model` is an `engine.SvdModel` holding the restored tables (`graph.Saver.restore` or `set_tables`); `test` is a frame
with columns user, item, outcome (dataio.read_process) or a (users, items, outcomes) tuple.
"""
from collections import defaultdict

import numpy as np

from . import _lib as L
from . import cats
from .ops import sigmoid

# var_list=[user_bias, user_features] (adaptive_test.py:28): every other table is frozen
FROZEN_BUT_USER = (1 << L.MU) | (1 << L.BI) | (1 << L.Q)


def _columns(test):
    if hasattr(test, "columns"):
        return (np.asarray(test["user"], np.int64), np.asarray(test["item"], np.int64), np.asarray(test["outcome"], np.float32))
    u, i, r = test
    return np.asarray(u, np.int64), np.asarray(i, np.int64), np.asarray(r, np.float32)


def _head(model, logits):
    """`infer` of the live graph: round(sigmoid(logit)) for the fork's binary head (ops.py:76-78), the logit itself for
    the canonical regression head (README.md:33)."""
    return np.round(sigmoid(logits)) if model.loss == "nll" else logits


def roc_auc(truth, score):
    """roc_auc_score by rank sums with mid-ranks for ties (what sklearn computes); nan when one class is missing."""
    truth = np.asarray(truth, np.float64) > 0.5
    score = np.asarray(score, np.float64)
    npos, nneg = int(truth.sum()), int((~truth).sum())
    if npos == 0 or nneg == 0:
        return float("nan")
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    ranks = np.empty(s.size, np.float64)
    lo = 0
    while lo < s.size:                          # mid-rank of each block of equal scores
        hi = lo
        while hi + 1 < s.size and s[hi + 1] == s[lo]:
            hi += 1
        ranks[order[lo:hi + 1]] = 0.5 * (lo + hi) + 1.0
        lo = hi + 1
    return float((ranks[truth].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def non_adaptive_test(model, test, epoch_max=100, max_user=None, freeze=True, log=None):
    """non_adaptive_test.py:56-121.  Rows of `test` in order; each is predicted with the parameters of the moment, added
    to its user's history, and the user tables are then trained `epoch_max` steps on that history.  Stops at the first row
    whose user id exceeds `max_user` (:58-59).  Returns accuracy / AUC of the predictions made before training (:120-121)
    and the lists behind them."""
    users, items, outcomes = _columns(test)
    if freeze:
        model.set_frozen(FROZEN_BUT_USER)
    hist = defaultdict(lambda: ([], [], []))
    truth, pred = [], []
    for u, i, r in zip(users.tolist(), items.tolist(), outcomes.tolist()):
        if max_user is not None and u > max_user:
            break
        hu, hi, hr = hist[u]
        hu.append(u); hi.append(i); hr.append(r)
        proba = float(sigmoid(model.forward([u], [i]))[0])                   # :70-72
        truth.append(r)
        pred.append(proba)
        _, loss = model.train_steps_repeat(hu, hi, hr, epoch_max, want_logits=False, want_loss=log is not None)   # :77-82
        if log is not None:
            log(dict(user=u, item=i, outcome=r, predicted=proba, history=len(hu), last_cost=float(loss[-1])))
    truth_a, pred_a = np.asarray(truth, np.float32), np.asarray(pred, np.float64)
    return dict(accuracy=float(np.mean(np.round(pred_a) == truth_a)) if truth else float("nan"),
                auc=roc_auc(truth_a, pred_a), truth=truth, pred=pred)


def adaptive_test(model, test, budget=10, epoch_max=300, selector=cats.Next, max_users=3, ask_everything=False,
                  popularity=None, freeze=True, log=None):
    """adaptive_test.py:54-127.  For each of the first `max_users` distinct users of `test` (:56, "FIXME" slice): a
    selector over that user's test items hands out `budget` items; each is predicted, its outcome (first matching row,
    :94-96) joins the training set, and the user tables are trained `epoch_max` steps on everything asked so far.
    `ask_everything` starts from the user's whole test set instead (:70-75).  Returns one record per user with the asked
    items, the predictions made before each was trained on, and the training metrics of the last step (:121-131)."""
    users, items, outcomes = _columns(test)
    if freeze:
        model.set_frozen(FROZEN_BUT_USER)
    out = []
    seen = []
    for u in users.tolist():                                                # df_test['user'].unique(): first-appearance order
        if u not in seen:
            seen.append(u)
    for this_user in seen[:max_users]:
        rows = np.nonzero(users == this_user)[0]
        t_items, t_rates = items[rows], outcomes[rows]
        if len(t_items) < budget:
            raise ValueError("user %d has %d test items, fewer than the budget of %d" % (this_user, len(t_items), budget))
        tr_i, tr_r = ([], []) if not ask_everything else (t_items.tolist(), t_rates.tolist())
        cat = selector(t_items, popularity) if selector is cats.Popular else selector(t_items)
        rec = dict(user=int(this_user), asked=[], predicted=[], outcome=[])
        logits = loss = None
        for b in range(budget):
            item = cat.next_item()
            rate = float(t_rates[np.nonzero(t_items == item)[0][0]])
            if not ask_everything:
                tr_i.append(item); tr_r.append(rate)
            proba = float(sigmoid(model.forward([this_user], [item]))[0])    # :100-102
            rec["asked"].append(int(item)); rec["predicted"].append(proba); rec["outcome"].append(rate)
            logits, loss = model.train_steps_repeat([this_user] * len(tr_i), tr_i, tr_r, epoch_max)      # :104-116
            if log is not None:
                log(dict(user=int(this_user), budget=b, item=int(item), predicted=proba, outcome=rate))
        infer = _head(model, logits)
        tr = np.asarray(tr_r, np.float32)
        rec.update(size=len(tr_i), macc=float(np.mean(infer == tr)), mobo=float(np.mean(np.abs(infer - tr) <= 1)),
                   rmse=float(np.sqrt(np.mean((infer - tr) ** 2))), mcost=float(loss[-1]))               # :121-131
        out.append(rec)
    return out
