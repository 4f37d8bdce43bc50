"""Graph-recording host code (ops / graph) - call shapes of ops.py:6,118 and
svd_train_val.py:40-57.  CPU only: nothing here creates a device model."""
import numpy as np
import pytest

from tfrecomm_amd import _lib as L
from tfrecomm_amd import graph as tf
from tfrecomm_amd import ops


@pytest.fixture(autouse=True)
def fresh_graph():
    tf.reset_default_graph()
    yield
    tf.reset_default_graph()


def _ph():
    return tf.placeholder("int32", shape=[None], name="id_user"), tf.placeholder("int32", shape=[None], name="id_item"), \
        tf.placeholder("float32", shape=[None])


def test_north_star_signature():
    u, i, r = _ph()
    infer, reg = ops.inference_svd(u, i, 6040, 3952, 15)               # (user_batch,item_batch,user_num,item_num,dim)
    spec = tf.get_default_graph().spec
    assert (spec["user_num"], spec["item_num"], spec["dim"]) == (6040, 3952, 15)
    assert (spec["loss"], spec["item_abs"], spec["reg_bias"]) == ("mse", False, False)
    with pytest.raises(AssertionError):                                # ops.py:119-120: needs a global step
        ops.optimization(infer, reg, r, learning_rate=1e-3, reg=0.05)
    tf.get_or_create_global_step()
    cost, train_op = ops.optimization(infer, reg, r, learning_rate=1e-3, reg=0.05)
    assert tf.get_default_graph().train == dict(optimizer="adam", adam_mode="tf1", lr=1e-3, reg=0.05, frozen=0)
    assert cost.kind == "cost" and train_op.kind == "train_op"


def test_fork_signature_and_var_list():
    u, i, r = _ph()
    w, f = tf.placeholder("float32", name="nb_wins"), tf.placeholder("float32", name="nb_fails")
    out = ops.inference_svd(u, i, w, f, user_num=30, item_num=20, dim=20, device="/cpu:0", fork_semantics=True)
    assert len(out) == 7                                               # ops.py:91
    infer, logits, regularizer, user_bias, user_features, item_bias, item_features = out
    spec = tf.get_default_graph().spec
    assert (spec["loss"], spec["item_abs"], spec["reg_bias"]) == ("nll", True, True)
    tf.get_or_create_global_step()
    ops.optimization(infer, logits, regularizer, r, learning_rate=5e-3, reg=0.01, device="/cpu:0",
                     var_list=[user_bias, user_features])               # adaptive_test.py:28
    t = tf.get_default_graph().train
    assert t["optimizer"] == "sgd"                                     # ops.py:145,149
    assert t["frozen"] == (1 << L.MU) | (1 << L.BI) | (1 << L.Q)


def test_default_is_dim_5_and_bad_arguments():
    u, i, r = _ph()
    ops.inference_svd(u, i, user_num=10, item_num=10)
    assert tf.get_default_graph().spec["dim"] == 5                     # ops.py:6 default
    with pytest.raises(RuntimeError):
        ops.inference_svd(u, i, user_num=10, item_num=10)              # one model per graph
    tf.reset_default_graph()
    with pytest.raises(TypeError):
        ops.inference_svd(u, i, 10, 10, 8, user_num=3)
    with pytest.raises(ValueError):
        ops.inference_svd(u, i, 10, 10, 8, loss="hinge")


def test_sigmoid_helper():
    assert ops.sigmoid(0.0) == 0.5
    assert np.allclose(ops.sigmoid(np.array([-2.0, 2.0])).sum(), 1.0)


def test_run_without_model_or_init_is_an_error():
    with pytest.raises(RuntimeError):
        tf.Session().run(tf.global_variables_initializer())


def test_csv_loaders_match_the_reference_readers_frames(golden):
    """f3 pinned: tests/golden/csv_frames.npz holds the frames the REAL /root/reference/dataio.py
    (build_paths / get_data / read_process) returned for the files under tests/golden/csv_fixture/
    (make_golden.py part 7); the product's loaders must return the same columns, dtypes and values."""
    import os
    from tfrecomm_amd import dataio
    g = golden("csv_frames.npz")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "csv_fixture")
    got_paths = dataio.build_paths("tiny")
    assert [p.replace(os.sep, "/") for p in got_paths] == list(g["paths"])
    tr, va, te = dataio.get_data("tiny", data_folder=os.path.join(root, "data"))
    frames = {"train": tr, "val": va, "test": te,
              "tabbed": dataio.read_process(os.path.join(root, "data", "tiny", "tabbed.tsv")),
              "all": dataio.get_new_data("tiny", data_folder=os.path.join(root, "data"))}      # dataio.py:57-60
    assert [p.replace(os.sep, "/") for p in dataio.build_new_paths("tiny")] == list(g["new_paths"])   # dataio.py:19-28
    import json
    cases, want = json.loads(str(g["legend_cases"])), json.loads(str(g["legends"]))                # dataio.py:63-87
    assert len(cases) == len(want) >= 9
    for args, (short, full, latex, active) in zip(cases, want):
        assert dataio.get_legend(args) == (short, full, latex, active), args
    for name, df in frames.items():
        assert list(df.columns) == list(g[name + "/columns"])
        for c in df.columns:
            assert str(df[c].dtype) == str(g["%s/%s/dtype" % (name, c)]), (name, c)
            assert np.array_equal(df[c].to_numpy(), g["%s/%s" % (name, c)], equal_nan=True), (name, c)
    cfg = dataio.get_config(os.path.join(root, "data", "tiny", "config.yml"))
    assert cfg == {"USER_NUM": 40, "ITEM_NUM": 30, "NB_CLASSES": 2, "BATCH_SIZE": 8}


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver uses at N=1) must start two
    ranks itself, before anything touches a GPU.  In this GPU-less container each child then stops at "no HIP device";
    the parent relays that and returns their code instead of raising SystemExit("launch with torch.distributed.run")."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = env["ROCR_VISIBLE_DEVICES"] = ""         # also on a GPU box: this test is about the launcher
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode()
    assert "started 2 ranks" in err, err
    assert "rank 0/2 needs an MI355X" in err and "rank 1/2 needs an MI355X" in err, err
    assert p.returncode != 0 and p.stdout.decode().strip() == ""


def test_bench_finds_the_committed_counter_summaries():
    """bench.py reads `roofline.traffic` from profiles/<round>_pmc_<workload>.csv: every kernel a BENCH roofline block names must be
    in the committed summaries of the current round (a renamed template instantiation or a missing file would silently give null)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    r = bench.PROFILE_ROUND
    for fname, kern in ((r + "_pmc_c2.csv", "k_tile_step<16, 4, 2>"), (r + "_pmc_c2.csv", "k_dense_tiles<16, 4, false, 10>"),
                        (r + "_pmc_c3.csv", "k_seg_reduce<32, 4, 1, true, true, true>"), (r + "_pmc_c3.csv", "k_seg_reduce<32, 4, 1, false, true, true>"),
                        (r + "_pmc_c4.csv", "k_seg_reduce<32, 4, 1, true, true, true>"), (r + "_pmc_c5.csv", "k_fm_forward<16, 4, false, true>"),
                        (r + "_pmc_forward_uniform.csv", "k_forward<32, 4, 0, 4, true>"), (r + "_pmc_forward_zipf.csv", "k_forward<32, 4, 0, 4, true>"),
                        (r + "_pmc_forward_8x_batch.csv", "k_forward<32, 4, 0, 4, true>")):
        t = bench.profiled_traffic(fname, kern)
        assert t is not None and t > 1e6, (fname, kern, t)
    # the forward's counter traffic: 2.61 M read requests of 128 B per 262144-rating launch (+ ~1 MB written)
    assert abs(bench.profiled_traffic(r + "_pmc_forward_uniform.csv", "k_forward<32, 4, 0, 4, true>") - 335.5e6) < 5e6
