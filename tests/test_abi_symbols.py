"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/refid_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from refid_amd.build import build
    return build()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "refid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(refid_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("refid_conv2d", "refid_conv2d_wgrad", "refid_pack_conv_weights", "refid_layernorm2d_fwd",
                 "refid_layernorm2d_bwd", "refid_dwconv3x3_gelu_fwd", "refid_se_fwd", "refid_charbonnier",
                 "refid_clip_adamw", "refid_last_error"):
        assert must in syms
    assert len(syms) >= 30


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in refid_hip.h but not exported: {missing}"
    lib.refid_abi_version.restype = ctypes.c_int
    assert lib.refid_abi_version() == 9


def test_python_binding_loads_and_host_queries_work(libpath):
    from refid_amd import _lib, ops
    L = _lib.lib()
    assert L.refid_conv_kc(3, 3, 1, 0) == 8 and L.refid_conv_kc(1, 1, 1, 0) == 32
    assert L.refid_conv_kc(7, 7, 1, 0) == -1
    assert ops.conv_bn(3, 3, 1, 0, 64) == 64 and ops.conv_bn(3, 3, 1, 0, 256) == 128
    assert L.refid_conv_tile_name(3, 3, 1, 0, 64) == b"Cfg<3, 3, 1, 4, 1, 2, 2, 1, 0>"
    # FWD packing of a (64,128,3,3) weight: [16 chunks][9 taps][64 rows][8]
    assert ops.packed_weight_floats(ops.ROLE_FWD, 64, 8, 3, 3, 64, 128) == 16 * 9 * 64 * 8


def test_struct_layouts_match_the_header():
    """ctypes mirrors of refid_conv_desc / refid_wgrad_desc: same field order as the header."""
    from refid_amd._lib import ConvDesc, WgradDesc
    src = open(os.path.join(ROOT, "include", "refid_hip.h")).read()

    def fields(struct_name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct_name, struct_name), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            names = re.sub(r"^(const\s+)?(float|int|size_t|refid_pw_extras)\s*\*?", "", stmt)
            out += [re.sub(r"\[.*\]$", "", n.strip().lstrip("*").strip()) for n in names.split(",")]
        return out

    assert fields("refid_conv_desc") == [f[0] for f in ConvDesc._fields_]
    from refid_amd._lib import PwExtras
    assert fields("refid_pw_extras") == [f[0] for f in PwExtras._fields_]
    assert fields("refid_wgrad_desc") == [f[0] for f in WgradDesc._fields_]
