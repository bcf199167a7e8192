"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/adenet.h declares; argument validation works without a GPU (no compute is launched)."""
import ctypes as C
import os
import re

import pytest

from ip_avsr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from ip_avsr_amd.build import build
        build(verbose=False)
    return _lib.load()


def test_header_and_binding_agree(lib):
    header = open(os.path.join(ROOT, "include", "adenet.h")).read()
    declared = set(re.findall(r"\b(adn_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(lib):
    # adn_stream_config: 2 + 8 + 8 + 4 (use_delta, bidirectional, peepholes, dropout_p) + 2 (batchnorm, aux_dim) 4-byte fields
    assert C.sizeof(_lib.StreamConfig) == 4 * (2 + 8 + 8 + 4 + 2)
    assert C.sizeof(_lib.Config) == 4 + 8 * C.sizeof(_lib.StreamConfig) + 4 * 14
    assert C.sizeof(_lib.ParamInfo) == 96 + 4 + 4 + 16 + 8      # name, ndim, pad, dims, numel
    out = (C.c_int32 * 4)()
    lib.adn_abi_sizes(out)                                       # ... and what the compiled library itself says
    assert list(out) == [C.sizeof(_lib.StreamConfig), C.sizeof(_lib.Config), C.sizeof(_lib.ParamInfo), C.sizeof(_lib.ProfileEntry)]


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.adn_version()
    assert isinstance(lib.adn_last_error(), bytes)


def test_invalid_config_is_rejected_before_touching_the_device(lib):
    cfg = _lib.Config()
    cfg.n_streams = 0
    h = C.c_void_p()
    st = lib.adn_create(C.byref(cfg), C.byref(h))
    assert st == 1 and b"n_streams" in lib.adn_last_error()
    cfg.n_streams = 2
    cfg.lstm_size = 8
    cfg.classes = 4
    cfg.fusion = _lib.FUSION["concat"]
    cfg.agg = 0
    for k in range(2):
        cfg.streams[k].input_dim = 4
    st = lib.adn_create(C.byref(cfg), C.byref(h))
    assert st == 1 and b"concat" in lib.adn_last_error()


def test_product_path_fails_loudly_without_a_gpu(lib):
    if lib.adn_device_count() > 0:
        pytest.skip("a gfx950 device is visible")
    cfg = _lib.Config()
    cfg.n_streams = 1
    cfg.lstm_size = 8
    cfg.classes = 4
    cfg.streams[0].input_dim = 4
    h = C.c_void_p()
    st = lib.adn_create(C.byref(cfg), C.byref(h))
    assert st == 3, "expected ADN_ERR_NO_DEVICE"
    assert b"no CPU fallback" in lib.adn_last_error() or b"gfx950" in lib.adn_last_error()
    from ip_avsr_amd.model import AdeNetModel, AdenetError
    spec = dict(streams=[dict(input_dim=4, enc_names=[], enc_shapes=[], enc_acts=[], delta=True,
                              lstm_names=["lstm"], peepholes=False)], fusion="none", fuse_name="",
                agg_names=[], agg_peepholes=False, lstm_size=8, classes=4, softmax_name="softmax")
    with pytest.raises(AdenetError):
        AdeNetModel(spec)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ip_avsr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)


def test_lstm_launch_plan_covers_every_pair_once_and_fits_the_device(lib):
    """csrc/lstm_cluster.hip::plan_pairs (host logic, runs without a GPU): the (LSTM, utterance group) pairs of a call go out in
    the fewest launches whose workgroups are all resident at once, in equal shares"""
    def plan(n, groups, cwg, cus):
        p0, cnt = (C.c_int32 * 64)(), (C.c_int32 * 64)()
        k = lib.adn_debug_plan_lstm_launches(n, groups, cwg, cus, p0, cnt, 64)
        assert 0 < k <= 64
        return [(p0[i], cnt[i]) for i in range(k)]
    for n, groups, cwg, cus in [(3, 17, 4, 256), (2, 17, 4, 256), (4, 17, 8, 256), (2, 17, 8, 256), (4, 11, 16, 256), (2, 11, 16, 256),
                                (6, 1, 16, 256), (5, 7, 4, 24), (1, 64, 4, 256), (8, 3, 16, 64)]:
        launches = plan(n, groups, cwg, cus)
        cap = max(1, cus // cwg)
        assert len(launches) == -(-n * groups // cap)                      # the fewest launches the device allows
        nxt = 0
        for p0, cnt in launches:
            assert p0 == nxt and 1 <= cnt <= cap                           # contiguous, resident
            nxt += cnt
        assert nxt == n * groups                                           # every pair once
        assert max(c for _, c in launches) - min(c for _, c in launches) <= 1          # equal shares
    assert plan(4, 17, 8, 256) == [(0, 23), (23, 23), (46, 22)]            # four 512-unit LSTMs at B = 520: three launches, not four
    assert lib.adn_debug_plan_lstm_launches(0, 1, 4, 256, None, None, 0) < 0
