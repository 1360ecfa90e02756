"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/brl_hip.h declares; the host mirror keeps the reference's names; there is no
fallback path when the extension or the GPU is missing.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from brl_amd import build
    return build.build()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "brl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(brl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = header_symbols()
    for s in ("brl_create", "brl_destroy", "brl_init_random", "brl_init_from_deals", "brl_step", "brl_observe",
              "brl_rollout_random", "brl_policy_step", "brl_gae", "brl_imp_reward", "brl_duplicate_step",
              "brl_get_fields", "brl_set_lut", "brl_set_rng", "brl_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/brl_hip.h but not exported"


def test_binding_covers_every_declared_symbol(built_lib):
    from brl_amd import _capi
    assert sorted(_capi.EXPORTS) == header_symbols()
    L = _capi.lib()
    assert L.brl_version() == 6   # include/brl_hip.h: the round the exported set last changed in
    assert L.brl_last_error() is not None


def test_binding_argument_types_match_header(built_lib):
    """Every argument of every declaration in include/brl_hip.h against the ctypes argtypes: beyond the sixth integer argument
    the x86-64 ABI passes on the stack, where an `int` bound to an `int64_t` parameter reads 32 bits of garbage."""
    from brl_amd import _capi
    text = open(os.path.join(ROOT, "include", "brl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    decls = re.findall(r"\b(?:int|const char \*|void)\s*(brl_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert sorted(n for n, _ in decls) == header_symbols()
    scalar = {"int": 4, "int32_t": 4, "uint32_t": 4, "unsigned": 4, "int64_t": 8, "uint64_t": 8, "float": "f", "double": "d"}

    def c_kind(arg):
        if "*" in arg:
            return "ptr"
        words = arg.split()
        return scalar[" ".join(words[:-1]) if len(words) > 1 else words[0]]

    def py_kind(t):
        if t in (ctypes.c_float,):
            return "f"
        if t in (ctypes.c_double,):
            return "d"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(t, ctypes._Pointer):
            return "ptr"
        return ctypes.sizeof(t)

    L = _capi.lib()
    for name, args in decls:
        want = [] if args.strip() in ("", "void") else [c_kind(a.strip()) for a in args.split(",")]
        got = [py_kind(t) for t in (getattr(L, name).argtypes or [])]
        assert want == got, f"{name}: header {want} != ctypes {got}"


def test_struct_layouts_match_header():
    from brl_amd import _capi
    text = open(os.path.join(ROOT, "include", "brl_hip.h")).read()
    for cls, name in ((_capi.Fields, "brl_fields"), (_capi.TransitionPtrs, "brl_transition"),
                      (_capi.TableInfoPtrs, "brl_table_info")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        members = re.findall(r"\*\s*([a-z_]+)\s*;", body)
        assert members == cls._names, name


def test_macro_ext_layout_matches_header():
    """brl_macro_ext mixes ints, floats and pointers: member order and C types against the ctypes mirror."""
    from brl_amd import _capi
    text = open(os.path.join(ROOT, "include", "brl_hip.h")).read()
    body = re.search(r"typedef struct brl_macro_ext \{(.*?)\} brl_macro_ext;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = re.findall(r"([A-Za-z_0-9 ]+?[ \*]+)([a-z_]+)\s*;", body)
    want = [(n, "ptr" if "*" in t else t.split()[-1]) for t, n in decl]
    kinds = {ctypes.c_int32: "int32_t", ctypes.c_int64: "int64_t", ctypes.c_float: "float", ctypes.c_void_p: "ptr"}
    got = [(n, kinds[t]) for n, t in _capi.MacroExt._fields_]
    assert got == want
    assert ctypes.sizeof(_capi.MacroExt) == 152   # 88 + the in-kernel heads (3 pointers, ldh, hidden, fmt) + the partial-product heads (pointer, stride, ld, nparts)


def test_bad_arguments_are_reported_not_crashed(built_lib):
    from brl_amd import _capi
    L = _capi.lib()
    assert L.brl_set_rng(None, 0, 0) == -1  # BRL_E_ARG
    assert b"handle" in L.brl_last_error()
    assert L.brl_destroy(None) == 0


def test_no_cpu_fallback_without_gpu():
    import torch
    import brl_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(brl_amd._capi.BrlError):
        brl_amd.BridgeBidding(lut=None)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from brl_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.BrlError, match="no CPU fallback"):
        _capi.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "brl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and "bridge_oracle" not in src, f


def test_host_mirror_keeps_reference_names():
    import brl_amd
    from brl_amd import duplicate, gae, models, roll_out, utils
    assert brl_amd.BridgeBidding.observation_shape == (480,)  # ppo.py:241
    for mod, names in ((utils, ["auto_reset", "single_play_step_two_policy_commpetitive",
                                "single_play_step_two_policy_commpetitive_deterministic",
                                "single_play_step_free_run", "normal_step"]),
                       (roll_out, ["Transition", "make_roll_out"]), (gae, ["make_calc_gae"]),
                       (duplicate, ["_imp_reward", "duplicate_init", "duplicate_step", "Table_info"]),
                       (models, ["ActorCritic", "make_forward_pass"])):
        for n in names:
            assert hasattr(mod, n), n
    assert roll_out.Transition._fields == ("done", "action", "value", "reward", "log_prob", "obs", "legal_action_mask")
    assert duplicate.Table_info._fields == ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx")


def test_mlp_shapes_match_reference_model():
    import torch
    from brl_amd.models import make_forward_pass
    for model, nparams in (("DeepMind", 480 * 1024 + 1024 + 3 * (1024 * 1024 + 1024) + 1024 * 38 + 38 + 1024 + 1),
                           ("FAIR", None)):
        fp = make_forward_pass("relu", model)
        net = fp.init(0)
        logits, value = fp.apply(net, torch.zeros(5, 480))
        assert logits.shape == (5, 38) and value.shape == (5,)
        if nparams:
            assert sum(p.numel() for p in net.parameters()) == nparams == 3681319
