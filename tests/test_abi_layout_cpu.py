"""The ctypes mirrors in unimedvl_amd/_lib.py must have exactly the layout a C compiler gives the structs of
include/unimedvl_hip.h (sizes and every field offset): the header is compiled with gcc into a tiny program that prints
them.  Guards the C ABI against drift when a field is added on one side only."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "unimedvl_hip.h")
EXP_HEADER = os.path.join(ROOT, "experimental", "include", "unimedvl_hip_experimental.h")
INCLUDE = os.path.join(ROOT, "include")

PAIRS = {   # C struct -> ctypes class name in unimedvl_amd._lib (the last three: experimental._lib)
    "umv_gemm_args": "GemmArgs",
    "umv_qkv_post_args": "QkvPostArgs",
    "umv_attn_args": "AttnArgs",
    "umv_gemm8_args": "Gemm8Args",
    "umv_attn_decode_args": "AttnDecodeArgs",
    "umv_decode_layout": "DecodeLayout",
    "umv_de_op": "DeOp",
}


def _c_fields(struct):
    """Field names of `typedef struct { ... } <struct>;` in declaration order (comments stripped, `a, b;` lists split)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read() + open(EXP_HEADER).read(), flags=re.S)
    m = re.search(r"typedef struct\s*\{([^{}]*)\}\s*" + struct + r"\s*;", src)
    assert m, struct
    names = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1])
    return names


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_ctypes_structs_match_the_header(tmp_path):
    from experimental import _lib as xlib
    from unimedvl_amd import _lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', f'#include "{EXP_HEADER}"', "int main(void) {"]
    for cs in PAIRS:
        lines.append(f'  printf("{cs} size %zu\\n", sizeof({cs}));')
        for f in _c_fields(cs):
            lines.append(f'  printf("{cs} {f} %zu\\n", offsetof({cs}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", f"-I{INCLUDE}", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True)
    c = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        c.setdefault(s, {})[f] = int(v)
    for cs, pyname in PAIRS.items():
        cls = getattr(_lib, pyname, None) or getattr(xlib, pyname)
        assert ctypes.sizeof(cls) == c[cs]["size"], f"{cs}: sizeof {c[cs]['size']} in C, {ctypes.sizeof(cls)} in ctypes"
        py_fields = [n for n, _ in cls._fields_]
        assert py_fields == _c_fields(cs), f"{cs}: field order differs\n C : {_c_fields(cs)}\n py: {py_fields}"
        for n in py_fields:
            assert getattr(cls, n).offset == c[cs][n], f"{cs}.{n}: offset {c[cs][n]} in C, {getattr(cls, n).offset} in ctypes"


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_integration_md_stubs_match_the_header(tmp_path):
    """INTEGRATION.md section 2 shows a maintainer the ctypes binding to paste into the reference tree.  Every `ctypes.Structure` block
    in that file is EXECUTED here and compared with the layout gcc gives the header's struct it claims to mirror (size, field order,
    every offset) and with unimedvl_amd._lib's own mirror: a stub that lags the header (a copied GemmArgs was 32 bytes
    short, the library would have read garbage for the sampling fields) fails the CPU suite."""
    from unimedvl_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"^class (\w+)\(ctypes\.Structure\):\s*#\s*mirrors (umv_\w+)[^\n]*\n(\s+_fields_ = \[.*?\])\n", md, flags=re.S | re.M)
    assert {b[0] for b in blocks} >= {"GemmArgs", "AttnArgs"}, [b[0] for b in blocks]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for _, cs, _ in blocks:
        lines.append(f'  printf("{cs} size %zu\\n", sizeof({cs}));')
        for f in _c_fields(cs):
            lines.append(f'  printf("{cs} {f} %zu\\n", offsetof({cs}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi_md.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_md"
    subprocess.check_call(["gcc", "-std=c11", f"-I{INCLUDE}", "-o", str(exe), str(src)])
    c = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        st, f, v = ln.split()
        c.setdefault(st, {})[f] = int(v)
    for pyname, cs, fields_src in blocks:
        ns = {"ctypes": ctypes}
        exec(f"class {pyname}(ctypes.Structure):\n{fields_src}\n", ns)
        cls = ns[pyname]
        names = [n for n, _ in cls._fields_]
        assert names == _c_fields(cs), f"INTEGRATION.md {pyname}: fields differ from {cs}\n C : {_c_fields(cs)}\n md: {names}"
        assert ctypes.sizeof(cls) == c[cs]["size"], f"INTEGRATION.md {pyname}: sizeof {ctypes.sizeof(cls)} vs {c[cs]['size']} in C"
        for n in names:
            assert getattr(cls, n).offset == c[cs][n], f"INTEGRATION.md {pyname}.{n}: offset {getattr(cls, n).offset} vs {c[cs][n]} in C"
        mine = getattr(_lib, pyname)
        assert [(n, ctypes.sizeof(t)) for n, t in cls._fields_] == [(n, ctypes.sizeof(t)) for n, t in mine._fields_]


def test_ctypes_signatures_match_the_header():
    """Every function the header declares is bound in _lib._SIGS with the same number of parameters, pointer parameters as
    pointers / void*, 64-bit integers as 64-bit, floats as floats."""
    from experimental import _lib as xlib
    from unimedvl_amd import _lib
    _check_signatures(HEADER, _lib._SIGS, 38)
    _check_signatures(EXP_HEADER, xlib._EXP_SIGS, 8)


def _check_signatures(header, sigs, at_least):
    src = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    src = re.sub(r"typedef struct\s*\{[^{}]*\}\s*\w+\s*;", "", src)
    decls = re.findall(r"\b(?:int|size_t|const char\s*\*)\s+(umv_\w+)\s*\(([^()]*)\)\s*;", src)
    assert len(decls) >= at_least, len(decls)
    for name, params in decls:
        assert name in sigs, f"{name} is declared in the header but not bound"
        plist = [p.strip() for p in params.split(",")] if params.strip() not in ("", "void") else []
        argtypes = sigs[name][1]
        assert len(argtypes) == len(plist), f"{name}: {len(plist)} parameters in the header, {len(argtypes)} in _SIGS"
        for p, t in zip(plist, argtypes):
            if "*" in p or "umv_stream_t" in p:
                assert t is ctypes.c_void_p or t is ctypes.c_char_p or hasattr(t, "_type_") and not isinstance(t._type_, str), f"{name}: {p} vs {t}"
            elif re.search(r"\b(int64_t|size_t|uint64_t)\b", p):
                assert ctypes.sizeof(t) == 8, f"{name}: {p} vs {t}"
            elif re.search(r"\bfloat\b", p):
                assert t is ctypes.c_float, f"{name}: {p} vs {t}"
            else:
                assert t in (ctypes.c_int, ctypes.c_uint, ctypes.c_bool), f"{name}: {p} vs {t}"
    assert set(sigs) == {n for n, _ in decls}, set(sigs) ^ {n for n, _ in decls}
