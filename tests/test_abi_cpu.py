"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/cinema_hip.h declares, the ctypes mirror of `cinema_siren_desc` has the C layout, and the
planner reproduces the reference's parameter count.  No kernel is launched."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cinema_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cinema_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for need in ("cinema_version", "cinema_last_error", "cinema_siren_plan", "cinema_siren_pack", "cinema_siren_forward",
                 "cinema_siren_backward", "cinema_affine_from_embedding", "cinema_latent_to_node_major",
                 "cinema_affine_trilerp_fwd", "cinema_affine_trilerp_bwd", "cinema_sample_loss_fwd_bwd",
                 "cinema_infer_minmax", "cinema_infer_finish"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from cinema_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in cinema_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert lib.cinema_version() >= 100


def test_ctypes_struct_matches_c_layout():
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from cinema_amd._lib import SirenDesc
    fields = ["n_layers", "omega_hidden", "HP", "off_w", "off_wf", "n_params", "pk_w", "pk_total", "sv_y", "gb_t"]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "cinema_hip.h"\nint main(){printf("%zu", sizeof(cinema_siren_desc));' + \
        "".join(f'printf(" %zu", offsetof(cinema_siren_desc, {f}));' for f in fields) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "t.c"), os.path.join(td, "t")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        vals = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == C.sizeof(SirenDesc)
    for f, v in zip(fields, vals[1:]):
        assert getattr(SirenDesc, f).offset == v, f


def test_plan_matches_reference_parameter_count():
    from cinema_amd import _lib, engine as E
    d = E.make_desc(3, 256, 12, 512, 6, [1, 3, 5], 30, 30, _lib.PREC_BF16)
    assert d.n_params == 2114060
    keys = [k for k, _, _ in E.param_layout(d)]
    assert keys[0] == "final_linear.weight" and keys[2] == "net.0.linear.weight"
    assert (d.HP, d.LP) == (512, 256)
    d2 = E.make_desc(3, 257, 11, 512, 6, [1, 3, 5], 30, 30, _lib.PREC_BF16)   # fetal VM config: lat = 257
    assert d2.LP == 320


def test_plan_rejects_bad_configs():
    from cinema_amd import _lib, engine as E
    with pytest.raises(_lib.CinemaHipError):
        E.make_desc(3, 256, 12, 500, 6, [1], 30, 30, _lib.PREC_BF16)     # hidden not a multiple of 16
    with pytest.raises(_lib.CinemaHipError):
        E.make_desc(3, 256, 40, 512, 6, [1], 30, 30, _lib.PREC_BF16)     # too many outputs
    with pytest.raises(_lib.CinemaHipError):
        E.make_desc(3, 256, 12, 512, 40, [1], 30, 30, _lib.PREC_BF16)    # too many layers


def test_engine_refuses_cpu():
    from cinema_amd import _lib, engine as E
    d = E.make_desc(3, 32, 11, 128, 4, [1, 3], 30, 30, _lib.PREC_BF16)
    with pytest.raises(_lib.CinemaHipError):
        E.SirenEngine(d, "cpu")


def test_work_size_queries_validate_their_arguments():
    """The two size queries are host-only arithmetic: good shapes give a positive size, bad ones -1 + a message, and the
    compute entry points refuse the same shapes before they touch the device."""
    import ctypes as C
    from cinema_amd import _lib
    lib = _lib.load()
    assert lib.cinema_conv_mod_work_bytes(64, 256, 3, 3, 3, 3, _lib.PREC_BF16) > 0
    assert lib.cinema_conv_mod_work_bytes(64, 256, 3, 3, 3, 2, _lib.PREC_BF16) > 0          # even kernel sizes are built (round 3)
    for bad in [(0, 256, 3, 3, 3, 3, _lib.PREC_BF16), (64, 250, 3, 3, 3, 3, _lib.PREC_BF16),      # S, C % 8
                (64, 256, 3, 3, 3, 17, _lib.PREC_BF16), (64, 256, 3, 3, 3, 3, 7),                  # kernel > 15, precision
                (64, 256, 17, 17, 17, 3, _lib.PREC_F32), (1 << 22, 256, 3, 3, 3, 3, _lib.PREC_F32)]:   # too many nodes
        assert lib.cinema_conv_mod_work_bytes(*bad) == -1
        assert b"cinema_conv_mod_work_bytes" in lib.cinema_last_error()
    assert lib.cinema_conv_mod_forward(None, None, None, 64, 256, 3, 3, 3, 2, _lib.PREC_BF16, None, None, None) != 0
    assert lib.cinema_conv_mod_forward(None, None, None, 64, 256, 3, 3, 3, 3, 7, None, None, None) != 0
    assert lib.cinema_conv_mod_backward(None, None, None, 64, 256, 3, 3, 3, 3, _lib.PREC_BF16, None, None, None, None, None) != 0
    d = _lib.SineLayerDesc(512, 256, 512, _lib.PREC_BF16, 30.0)
    assert lib.cinema_sine_layer_work_bytes(C.byref(d), 1000) > 0
    for f, v in [("out_feat", 500), ("in_feat", 0), ("lat_feat", -1), ("precision", 9), ("out_feat", 1 << 14)]:
        e = _lib.SineLayerDesc(512, 256, 512, _lib.PREC_BF16, 30.0)
        setattr(e, f, v)
        assert lib.cinema_sine_layer_work_bytes(C.byref(e), 1000) == -1
    assert lib.cinema_sine_layer_work_bytes(C.byref(d), 0) == -1
    assert lib.cinema_sine_layer_work_bytes(C.byref(d), 1 << 28) == -1
    assert lib.cinema_sine_layer_forward(C.byref(d), None, None, None, None, None, None, 1000, None, None, None) != 0
