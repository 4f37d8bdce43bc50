"""The C-ABI library loads and exports every symbol include/tfrecomm.h declares; argument
checking and the no-CPU-fallback rule.  No kernels are launched here (CPU suite)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tfrecomm_amd as T
from tfrecomm_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "tfrecomm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfr_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 30 and "tfr_train_step" in names and "tfr_forward" in names
    lib = L.load()
    for n in names:
        assert hasattr(lib, n), "libtfrecomm_hip.so does not export %s" % n
        assert n in L.SIGNATURES, "binding missing for %s" % n
    assert sorted(L.SIGNATURES) == names, "binding lists symbols the header does not declare"


def test_version_and_default_opts():
    lib = L.load()
    assert lib.tfr_version() == L.ABI_VERSION == 3
    o = L.TfrOpts()
    lib.tfr_default_opts(C.byref(o))
    assert (o.loss, o.item_abs, o.reg_bias, o.optimizer, o.adam_mode) == (0, 0, 0, 0, 0)
    assert o.beta1 == pytest.approx(0.9) and o.beta2 == pytest.approx(0.999) and o.eps == pytest.approx(1e-8)
    assert C.sizeof(L.TfrOpts) == 64


def test_bad_arguments_are_rejected_before_any_device_work():
    lib = L.load()
    o = L.TfrOpts()
    lib.tfr_default_opts(C.byref(o))
    h = L._p()
    assert lib.tfr_create(C.byref(h), 0, 10, 8, C.byref(o)) == L.ERR_ARG
    assert lib.tfr_create(C.byref(h), 10, 10, 0, C.byref(o)) == L.ERR_ARG
    assert lib.tfr_create(C.byref(h), 10, 10, 300, C.byref(o)) == L.ERR_ARG      # dim % 4 == 0 but > 256
    assert lib.tfr_create(C.byref(h), 10, 10, 65, C.byref(o)) == L.ERR_ARG       # odd dim > 64
    assert b"dim" in lib.tfr_last_error()
    o.loss = 7
    assert lib.tfr_create(C.byref(h), 10, 10, 8, C.byref(o)) == L.ERR_ARG
    assert lib.tfr_create(None, 10, 10, 8, C.byref(o)) == L.ERR_ARG
    assert lib.tfr_forward(None, None, None, 0, None) == L.ERR_ARG
    assert lib.tfr_destroy(None) == L.OK


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to run (it never routes through the oracle)."""
    if L.load().tfr_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu-marked tests")
    with pytest.raises(T.TfrError) as e:
        T.SvdModel(10, 10, 8)
    assert e.value.code == L.ERR_HIP and "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tf-recomm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "svd_oracle" not in text, f


def test_id_casts_are_exact():
    assert L.as_i32(np.array([1.0, 2.0])).dtype == np.int32
    assert L.as_i32(np.array([3, 4], np.int64)).tolist() == [3, 4]
    with pytest.raises(ValueError):
        L.as_i32(np.array([1.5]))
    with pytest.raises(IndexError):
        L.as_i32(np.array([2 ** 31], np.int64))
    with pytest.raises(TypeError):
        L.as_i32(np.array(["a"]))
