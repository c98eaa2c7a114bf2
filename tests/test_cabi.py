"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/audiodec_hip.h
declares; argument validation that needs no device is exercised.  No compute is launched."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "audiodec_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from audiodec_amd import native
    return native.lib()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(lib):
    from audiodec_amd import native
    decl = declared_symbols()
    assert len(decl) >= 18
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(adk_[a-z0-9_]+)\b", out))
    missing = [s for s in decl if s not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    unbound = [s for s in decl if s not in native.SYMBOLS]
    assert not unbound, f"declared in the header but not bound in native.SYMBOLS: {unbound}"
    for s in decl:
        assert getattr(lib, s) is not None


def test_abi_version_and_struct_sizes(lib):
    from audiodec_amd import native
    assert lib.adk_abi_version() == native.ABI_VERSION
    # layout the C side compiles to (x86-64): see include/audiodec_hip.h
    assert C.sizeof(native.RingView) == 24
    assert C.sizeof(native.ConvDesc) == 80
    assert C.sizeof(native.RingDesc) == 24
    assert C.sizeof(native.OpDesc) == 184          # fuse_next took the tail padding


def test_argument_validation_without_device(lib):
    from audiodec_amd import native
    assert lib.adk_packed_weight_floats(3, 256, 2816) == 3 * 8 * (2816 // 8) * 256
    assert lib.adk_packed_weight_floats(1, 96, 128) == 3 * 16 * 256            # 96 rows -> 3 m-tiles
    assert lib.adk_packed_weight_floats(1, 32, 352) == 1 * (384 // 8) * 256      # K padded 352 -> 384
    assert lib.adk_packed_weight_floats(1, 32, 7) == -1
    h = C.c_void_p()
    rc = lib.adk_program_create(None, 0, None, 0, 1, 1, None, 0, None, 0, C.byref(h))
    assert rc == -1 and b"null" in lib.adk_last_error()
    assert lib.adk_program_step(None, 1, None, 0, None) == -1
    assert lib.adk_rvq_encode(None, None, None, None, None, 1, 8, 64, 1024, None) == -1
    d = native.ConvDesc()
    v = native.RingView()
    assert lib.adk_causal_conv(C.byref(d), v, v, v, 1, 1, 0, None) == -1
    with pytest.raises(ValueError):
        native.check(-2, "x")
    with pytest.raises(native.NativeError):
        native.check(-3, "x")


def test_product_fails_loudly_without_gpu():
    import torch
    from audiodec_amd import native
    from audiodec_amd.stream_generator import AutoEncoderStreamGenerator
    with pytest.raises(native.NativeError):
        native.require_gpu("cpu")
    if not torch.cuda.is_available():
        m = AutoEncoderStreamGenerator().to("cuda:0")
        with pytest.raises(native.NativeError):
            m.initial_encoder(8192, "cuda:0")


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from audiodec_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeError, match="not built"):
        native.lib()


def test_expm1_neg_formula_error_bound():
    """numpy restatement (f32 arithmetic) of expm1_neg (csrc/adk_common.h), the ELU negative branch of every conv kernel, against
    fp64 expm1: <= 2.5e-7 relative (1.4e-7 with fused multiply-adds and a correctly rounded exp2) over [-20, 0)."""
    import numpy as np
    f = np.float32
    rng = np.random.default_rng(3)
    x = np.concatenate([-np.logspace(-8, np.log10(20.0), 100000), -rng.random(100000) * 0.8]).astype(f)
    q = np.full_like(x, f(1 / 5040))
    for c in (1 / 720, 1 / 120, 1 / 24, 1 / 6, 0.5):
        q = (q * x + f(c)).astype(f)
    p = ((x * x).astype(f) * q + x).astype(f)
    e = (np.exp2((x * f(1.4426950408889634)).astype(f).astype(np.float64)).astype(f) - f(1)).astype(f)
    r = np.where(x > f(-0.4), p, e)
    ref = np.expm1(x.astype(np.float64))
    assert np.max(np.abs(r - ref) / np.abs(ref)) < 2.5e-7
