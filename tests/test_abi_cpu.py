"""No-GPU checks of the C ABI: the library builds for gfx950, loads, and exports every symbol include/srhip.h declares;
the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__
    return __graft_entry__.build()


def test_every_declared_symbol_is_exported_and_bound(libpath):
    hdr = open(os.path.join(ROOT, "include", "srhip.h")).read()
    declared = set(re.findall(r"^int (sr_\w+)\(", hdr, flags=re.M))
    assert len(declared) >= 19
    lib = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(lib, name), name
    from situation_recognition_amd import _lib
    assert set(_lib.SIGNATURES) == declared
    assert lib.sr_abi_version() == _lib.ABI_VERSION == 5


def test_argument_validation_needs_no_gpu(libpath):
    from situation_recognition_amd import _lib
    l = _lib.lib()
    assert l.sr_gemm(None, 1, None) == -1                         # SR_ERR_ARG
    a = _lib.GemmArgs()
    a.npairs, a.M, a.N = 1, 4, 4
    assert l.sr_gemm(ctypes.byref(a), 7, None) in (-1, -2)
    assert l.sr_conv2d(None, 1, None) == -1
    # partial-statistics rows: one per (workgroup group, flush of up to 32 tiles, wave group of the tile shape the kernel
    # will pick for (M, N): 4 wave groups for 256x64 tiles, 2 for 256x128 / 256x256); 256 CUs assumed without a device
    assert l.sr_gemm_stats_tiles(1000, 64) == 16
    assert l.sr_gemm_stats_tiles(1000, 128) == 8
    assert l.sr_gemm_stats_tiles(1000, 256) == 16
    assert l.sr_gemm_stats_tiles(1204224, 256) == 256 * 2          # 256 workgroups x 19 tiles each: one flush
    assert l.sr_gemm_stats_tiles(1204224, 64) == 512 * 4
    assert l.sr_gemm_stats_tiles(19267584, 64) == 512 * 5 * 4      # 147 tiles per workgroup: five flushes


def test_gram_plan_needs_no_gpu(libpath):
    """Row slices of the Gram statistics kernel: one per CU (256 assumed without a device), split further so that no
    fp32 accumulator sums more than 8192 pixels; 64/128-wide inputs write 4/2 k-split partials per slice.
    (Fewer, longer slices to save partial traffic were measured slower at every size: the kernel wants all CUs.)"""
    from situation_recognition_amd import _lib
    l = _lib.lib()
    n, f = ctypes.c_int64(), ctypes.c_int64()
    assert l.sr_gram_plan(1204224, 256, ctypes.byref(n), ctypes.byref(f)) == 0 and (n.value, f.value) == (256, 256 * 256 + 256)
    assert l.sr_gram_plan(19267584, 64, ctypes.byref(n), ctypes.byref(f)) == 0 and (n.value, f.value) == (768 * 4, 64 * 64 + 64)
    assert l.sr_gram_plan(37, 512, ctypes.byref(n), ctypes.byref(f)) == 0 and (n.value, f.value) == (2, 512 * 512 + 512)
    assert l.sr_gram_plan(100, 96, ctypes.byref(n), ctypes.byref(f)) == -1
    assert l.sr_gram(None, 100, 64, 64, 1, None, 4, None) == -1


def test_no_cpu_fallback():
    from situation_recognition_amd._lib import SrError
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN, GGSNN
    enc = imsitu_encoder.synthetic(V=7, NR=5, L=11, R=3)
    net = FCGGNN(enc, 64, backbone=18, width=8, blocks=(1, 1, 1, 1), dtype=torch.float32)
    with pytest.raises(SrError):
        net(torch.randn(1, 3, 32, 32), torch.tensor([0]))
    with pytest.raises(SrError):
        GGSNN(64)(torch.randn(3, 64), mask=torch.ones(1, 3, 3))
    src = open(os.path.join(ROOT, "situation_recognition_amd", "model.py")).read() + \
        open(os.path.join(ROOT, "situation_recognition_amd", "ops.py")).read()
    assert "oracle" not in src.replace("the oracle", "")          # the product never imports the checker


def test_encoder_matches_reference_golden_tables():
    import json
    import numpy as np
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    g = np.load(os.path.join(ROOT, "tests", "golden", "g1_encoder.npz"))
    ts = json.load(open(os.path.join(ROOT, "tests", "golden", "overfitting.json")))
    enc = imsitu_encoder(ts, quiet=True)
    assert enc.verb_list == list(g["verb_list"]) and enc.role_list == list(g["role_list"]) and enc.label_list == list(g["label_list"])
    assert np.array_equal(enc.roles_to_verb_tensor_list.numpy(), g["roles_to_verb"])
    V = enc.get_num_verbs()
    assert np.array_equal(enc.get_adj_matrix_noself(torch.arange(V)).numpy(), g["adj_all_verbs"])
    assert np.array_equal(enc.get_role_ids_batch(torch.tensor([4, 0, 2, 2, 1])).numpy(), g["role_ids_batch"])
    assert [enc.get_role_count(v) for v in range(V)] == list(g["role_counts"])
    for i, ann in enumerate(ts.values()):
        v, lab = enc.encode(ann)
        assert v == int(g["encode_verb_%d" % i]) and np.array_equal(lab.numpy(), g["encode_labels_%d" % i])
    e2 = imsitu_encoder.from_state(json.loads(json.dumps(enc.state())))
    assert torch.equal(e2.adj_table, enc.adj_table) and torch.equal(e2.roles_to_verb_tensor_list, enc.roles_to_verb_tensor_list)
