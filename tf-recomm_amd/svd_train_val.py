"""Training / validation driver with the reference's structure (svd_train_val.py:23-198):
ShuffleIterator minibatches -> sess.run([train_op, logits, infer]) -> per-epoch validation
forward over the whole validation set -> the README's ``epoch train_error val_error
elapsed_time`` rows -> Saver.save.

    python -m tfrecomm_amd.svd_train_val [--data NAME | --movielens ratings.dat | --synthetic]
"""
from __future__ import annotations

import argparse
import os
import time
from collections import deque

import numpy as np

from . import config as C
from . import dataio, graph as tf, ops


def svd(train, test, *, user_num=None, item_num=None, dim=None, batch_size=None, epoch_max=None,
        learning_rate=None, reg=None, discrete=None, device=None, optimizer=None, adam_mode="tf1",
        save_path=None, log=print):
    """One training run; returns the list of (epoch, train_error, val_error, seconds) rows."""
    user_num = user_num or C.USER_NUM
    item_num = item_num or C.ITEM_NUM
    dim = dim or C.DIM
    batch_size = batch_size or C.BATCH_SIZE
    epoch_max = C.EPOCH_MAX if epoch_max is None else epoch_max
    learning_rate = C.LEARNING_RATE if learning_rate is None else learning_rate
    reg = C.LAMBDA_REG if reg is None else reg
    discrete = C.DISCRETE if discrete is None else discrete
    device = C.DEVICE if device is None else device

    nb_batches = len(train["user"]) // batch_size                       # svd_train_val.py:24
    iter_train = dataio.ShuffleIterator([train["user"], train["item"], train["outcome"]],
                                        batch_size=batch_size)          # :26-31
    iter_test = dataio.OneEpochIterator([test["user"], test["item"], test["outcome"]],
                                        batch_size=-1)                  # :33-38 whole set, one batch

    tf.reset_default_graph()
    user_batch = tf.placeholder("int32", shape=[None], name="id_user")  # :40-42
    item_batch = tf.placeholder("int32", shape=[None], name="id_item")
    rate_batch = tf.placeholder("float32", shape=[None])

    infer, regularizer = ops.inference_svd(user_batch, item_batch, user_num=user_num, item_num=item_num,
                                           dim=dim, device=device, fork_semantics=discrete)
    logits = tf.get_default_graph().node("logits", "logits")
    tf.get_or_create_global_step()                                      # :48
    kw = {} if optimizer is None else {"optimizer": optimizer}
    cost, train_op = ops.optimization(infer, regularizer, rate_batch, learning_rate=learning_rate, reg=reg,
                                      device=device, adam_mode=adam_mode, **kw)

    init_op = tf.group(tf.global_variables_initializer(), tf.local_variables_initializer())
    saver = tf.Saver()
    rows = []
    with tf.Session(seed=C.SEED, device=device) as sess:
        sess.run(init_op)
        log("{} {} {} {}".format("epoch", "train_error", "val_error", "elapsed_time"))
        train_se = deque(maxlen=nb_batches)                             # :59
        train_acc = deque(maxlen=nb_batches)
        train_auc = deque(maxlen=nb_batches)                            # :61-64 (fork): per-batch AUC and NLL
        train_nll = deque(maxlen=nb_batches)
        start = time.time()
        for i in range(epoch_max * nb_batches):                         # :66
            train_users, train_items, train_rates = next(iter_train)
            _, train_logits, train_infer, nll_batch = sess.run(
                [train_op, logits, infer, cost],
                feed_dict={user_batch: train_users, item_batch: train_items, rate_batch: train_rates})
            if discrete:
                train_acc.append(np.round(ops.sigmoid(train_logits)) == train_rates)      # :96
                train_auc.append(sess.model.last_batch_auc())           # :97 roc_auc_score, on the device
                train_nll.append(nll_batch)                             # :94,98 (the cost of this very run)
            else:
                train_se.append(np.power(train_rates - train_infer, 2))                   # :104
            if i % nb_batches == 0:                                     # :106 (also at i=0, after ONE step)
                train_err = np.mean(train_acc) if discrete else np.sqrt(np.mean(train_se))
                test_se, test_acc = [], []
                for test_users, test_items, test_rates in iter_test:    # :120-122
                    test_logits, test_infer = sess.run([logits, infer],
                                                       feed_dict={user_batch: test_users, item_batch: test_items})
                    if discrete:
                        test_acc.append(np.round(ops.sigmoid(test_logits)) == test_rates)
                    else:
                        test_se.append(np.power(test_rates - test_infer, 2))
                end = time.time()
                test_err = np.mean(test_acc) if discrete else np.sqrt(np.mean(test_se))   # :149
                row = (i // nb_batches, float(train_err), float(test_err), end - start)
                rows.append(row)
                if discrete:                                            # :170-178: the fork's epoch line, metrics from the device
                    tm = sess.model.eval_binary(test["user"], test["item"], test["outcome"])
                    log("{:3d} TRAIN(size={:d}/{:d}, macc={:f}, mauc={:f}, mnll={:f}) TEST(size={:d}, macc={:f}, auc={:f}, mnll={:f}) {:f}(s)".format(
                        i // nb_batches, len(train_users), len(train["user"]), float(train_err), float(np.mean(train_auc)),
                        float(np.mean(train_nll)) / batch_size, tm["n"], tm["acc"], tm["auc"], tm["mean_nll"], end - start))
                else:
                    log("{:3d} {:f} {:f} {:f}(s)".format(*row))         # README.md:49-57
                start = end
        if save_path:
            log(saver.save(sess, save_path))                            # :197-198
    return rows


def svd_resident(train, test, *, user_num=None, item_num=None, dim=None, batch_size=None, epoch_max=None,
                 learning_rate=None, reg=None, device=None, adam_mode="tf1", json_log=None, log=print):
    """The same run with every input resident in HBM: the rating store and the
    validation set are uploaded once, the reference's ``randint`` id stream is drawn ON THE DEVICE (NumPy's
    generator state goes there before an epoch and comes back after it, so host code that draws in between
    sees one unbroken stream), one epoch of minibatches is one C-ABI call, and the validation error is
    reduced on the device.  Prints the README rows and (optionally) writes one JSON line per epoch with
    ratings/sec."""
    import json
    from .engine import SvdModel
    user_num, item_num = user_num or C.USER_NUM, item_num or C.ITEM_NUM
    dim, batch_size = dim or C.DIM, batch_size or C.BATCH_SIZE
    epoch_max = C.EPOCH_MAX if epoch_max is None else epoch_max
    learning_rate = C.LEARNING_RATE if learning_rate is None else learning_rate
    reg = C.LAMBDA_REG if reg is None else reg
    device = C.DEVICE if device is None else device
    n_train = len(train["user"])
    nb_batches = n_train // batch_size                                   # svd_train_val.py:24
    rows, fh = [], open(json_log, "w") if json_log else None
    with SvdModel(user_num, item_num, dim, optimizer="adam", adam_mode=adam_mode, lr=learning_rate, reg=reg,
                  device=device) as m:
        m.init_tables(seed=C.SEED)
        m.upload_triples(train["user"], train["item"], train["outcome"])
        m.upload_eval_triples(test["user"], test["item"], test["outcome"])
        log("{} {} {} {}".format("epoch", "train_error", "val_error", "elapsed_time"))
        start = time.time()
        for epoch in range(epoch_max):
            # epoch 0 of the reference is evaluated after ONE step (svd_train_val.py:106); later rows
            # after nb_batches more
            steps = 1 if epoch == 0 else nb_batches
            m.rng_from_numpy()                                            # dataio.py:115: randint(0, n_train, (batch,)) per step,
            loss = m.train_steps_drawn(batch_size, steps, want_loss=True)  # drawn by the device; data term 0.5*sum(err^2) per step
            m.rng_to_numpy()
            train_err = float(np.sqrt(2.0 * loss.astype(np.float64).sum() / (steps * batch_size)))
            sse, _, n = m.eval_resident()
            end = time.time()
            row = (epoch, train_err, float(np.sqrt(sse / n)), end - start)
            rows.append(row)
            log("{:3d} {:f} {:f} {:f}(s)".format(*row))
            if fh:
                fh.write(json.dumps(dict(epoch=epoch, train_error=row[1], val_error=row[2], elapsed_time=row[3],
                                         ratings_per_sec=steps * batch_size / max(row[3], 1e-9))) + "\n")
                fh.flush()
            start = end
    if fh:
        fh.close()
    return rows


def synthetic_frames(user_num, item_num, n, seed=C.SEED):
    """ML-1M-shaped ratings from a low-rank ground truth (no dataset ships; no network)."""
    rs = np.random.RandomState(seed)
    u = rs.randint(0, user_num, n).astype(np.int32)
    i = rs.randint(0, item_num, n).astype(np.int32)
    p, q = rs.normal(0, 0.35, (user_num, 8)), rs.normal(0, 0.35, (item_num, 8))
    r = 3.58 + rs.normal(0, 0.35, user_num)[u] + rs.normal(0, 0.45, item_num)[i] + (p[u] * q[i]).sum(1)
    r = np.clip(np.rint(r + rs.normal(0, 0.85, n)), 1, 5).astype(np.float32)
    cut = int(0.9 * n)
    mk = lambda s: {"user": u[s], "item": i[s], "outcome": r[s]}
    return mk(slice(0, cut)), mk(slice(cut, n))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", help="prepared dataset under data/<name>/ (train.csv, val.csv)")
    ap.add_argument("--movielens", help="MovieLens ratings.dat (user::item::rating::timestamp)")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=C.BATCH_SIZE)
    ap.add_argument("--dim", type=int, default=C.DIM)
    a = ap.parse_args(argv)
    np.random.seed(C.SEED)                                              # svd_train_val.py:15
    if a.data:
        df_train, df_val, _ = dataio.get_data(a.data)
        cfg = dataio.get_config(dataio.build_paths(a.data)[4])
        un, inum = cfg["USER_NUM"], cfg["ITEM_NUM"]
    elif a.movielens:
        df = dataio.read_movielens(a.movielens)
        perm = np.random.RandomState(C.SEED).permutation(len(df))
        cut = int(0.9 * len(df))
        df_train, df_val = df.iloc[perm[:cut]], df.iloc[perm[cut:]]
        un, inum = int(df["user"].max()) + 1, int(df["item"].max()) + 1
    else:
        un, inum = C.USER_NUM, C.ITEM_NUM
        df_train, df_val = synthetic_frames(un, inum, 1000209)
    svd(df_train, df_val, user_num=un, item_num=inum, dim=a.dim, batch_size=a.batch, epoch_max=a.epochs,
        save_path=os.path.join(os.getcwd(), "fm.ckpt"))
    print("Done!")


if __name__ == "__main__":
    main()
