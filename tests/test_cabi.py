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
    assert C.sizeof(native.RingDesc) == 32 and native.RingDesc.extra_rows.offset == 24      # (ABI 13: extra_rows behind arena_off)
    assert C.sizeof(native.OpDesc) == 208 and native.OpDesc.chain.offset == 184 and native.OpDesc.fuse_next.offset == 180
    assert native.OpDesc.in_shadow.offset == 188 and native.OpDesc.shadow_slope.offset == 200
    # ... and what gcc makes of the header itself (the header is C: a maintainer's cgo / ctypes stub sees these numbers)
    import os, subprocess, tempfile
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        with open(src, "w") as fh:
            fh.write('#include <stdio.h>\n#include <stddef.h>\n#include "audiodec_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %d\\n", sizeof(adk_ring_view), '
                     'sizeof(adk_conv_desc), sizeof(adk_ring_desc), sizeof(adk_op_desc), offsetof(adk_op_desc, chain), ADK_ABI_VERSION);return 0;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), src, "-o", os.path.join(td, "sz")], check=True)
        out = subprocess.run([os.path.join(td, "sz")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in out] == [24, 80, 32, 208, 184, native.ABI_VERSION]


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
    assert lib.adk_program_step_ex(None, 1, None, 0, None, 1) == -1
    t = C.c_int64(0)
    assert lib.adk_program_flags_post(None, None, C.byref(t)) == -1 and lib.adk_program_flags_poll(None, 0, 0, None, None) == -1
    assert lib.adk_rvq_encode(None, None, None, None, None, 1, 8, 64, 1024, None) == -1
    d = native.ConvDesc()
    v = native.RingView()
    assert lib.adk_causal_conv(C.byref(d), v, v, v, 1, 1, 0, None) == -1
    with pytest.raises(ValueError):
        native.check(-2, "x")
    with pytest.raises(native.NativeError):
        native.check(-3, "x")


def test_named_options_validate_their_values(lib):
    """adk_set_option is host state only (no device): known names take their documented values, anything else is ADK_ERR_ARG with a message."""
    for name, good, bad in ((b"rvq_rows", (0, 2, 4, 1), (3, -1, 8)), (b"rvq_v4_min", (1, 192), ()), (b"chain_min_blocks", (160, 0), ()),
                            (b"chain_max_channels", (64, 128), ()), (b"gv16_max_columns", (256, 0, 32), ()),
                            (b"conv_ou16", (0, 1), ()), (b"conv_oc16", (0, 1), ()), (b"conv_cin1w", (0, 1), ())):
        for v in good:
            assert lib.adk_set_option(name, v) == 0, (name, v, lib.adk_last_error())
        for v in bad:
            assert lib.adk_set_option(name, v) == -1 and name in lib.adk_last_error(), (name, v)
    assert lib.adk_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.adk_last_error()
    assert lib.adk_set_option(None, 1) == -1


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
        # the reference's default devices ('cpu'): mapped to the first HIP device when there is one, an error when there is none
        from audiodec_amd.audiodec import AudioDec
        with pytest.raises(native.NativeError, match="no CPU implementation"):
            AudioDec()


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


def test_streamk_plan_covers_every_work_unit_once(lib):
    """adk_streamk_plan / adk_streamk_range_start (the host logic of conv_mfma.hip launch_cfg + sk_u0, shared with the kernels): for
    random (tiles, chunks, cap) the ranges start at 0, end at tiles * chunks, never go backwards, respect the cap; a tile-aligned
    plan never lets a range straddle two tiles (split) or cut a tile (whole tiles), keeps at most 5 ranges on a tile and gives every
    part of a split tile at least 2 chunks; the layers of the benched pipeline get the plans DESIGN.md describes."""
    import ctypes as C
    import random
    rnd = random.Random(7)
    cases = [(240, 44, 0), (240, 44, 256), (200, 6, 0), (400, 4, 256), (300, 22, 0), (80, 28, 0), (32, 40, 0), (4, 24, 0), (1, 3, 0), (938, 1, 0)]
    cases += [(rnd.randint(1, 3000), rnd.randint(1, 60), rnd.choice([0, 0, 8, 64, 256, 384, 512])) for _ in range(400)]
    plan = (C.c_int32 * 4)()
    for tiles, chunks, cap in cases:
        assert lib.adk_streamk_plan(tiles, chunks, cap, plan) == 0
        G, split, tpw, owner = plan[0], plan[1], plan[2], plan[3]
        limit = min(cap, 512) if cap else 512
        assert 1 <= G <= max(limit, 8), (tiles, chunks, cap, list(plan))
        starts = [lib.adk_streamk_range_start(tiles, chunks, plan, r) for r in range(G + 1)]
        assert starts[0] == 0 and starts[-1] == tiles * chunks, (tiles, chunks, cap, list(plan))
        assert all(b >= a for a, b in zip(starts, starts[1:]))
        assert not (split and tpw)
        if split:
            assert G == tiles * split and split <= 5
            for r in range(G):
                a, b = starts[r], starts[r + 1]
                assert a // chunks == r // split and (b - 1) // chunks == r // split, "a range of a split plan stays inside its tile"
                assert b - a >= (2 if split > 1 else 1)
            if split == 2:
                assert starts[1] - starts[0] == owner and 2 <= owner <= chunks - 2
        if tpw:
            assert all(a % chunks == 0 for a in starts) and G == -(-tiles // tpw)
    # the benched layers (256 streams, 64x64 tiles): grouped K11 256-ch -> exact halves with the owner share; 1x1 3C->C -> whole tiles
    assert lib.adk_streamk_plan(240, 44, 0, plan) == 0 and list(plan) == [480, 2, 0, 20]
    assert lib.adk_streamk_plan(200, 6, 0, plan) == 0 and list(plan)[:3] == [200, 1, 0]
    assert lib.adk_streamk_plan(300, 22, 0, plan) == 0 and list(plan)[:3] == [512, 0, 0]
    assert lib.adk_streamk_plan(0, 4, 0, plan) != 0 and lib.adk_streamk_range_start(10, 4, plan, 10 ** 6) == -1
