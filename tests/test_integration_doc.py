"""INTEGRATION.md is the binding a maintainer of the reference copies: its code must not rot.

Every fenced ``python`` block that defines a ctypes Structure is parsed; each class is built and compared, field by field,
with gcc's layout of the struct of include/vcr_hip.h it mirrors (the name is matched: KnnArgs -> vcr_knn_args).  Every
fenced ``c`` block is compiled with ``gcc -Wall -Werror -c`` against the header.  The ABI number the snippets assert must be
the header's.  No GPU, no library call."""
import ast
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vcr_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")


def fenced(lang):
    return re.findall(r"```%s\n(.*?)```" % lang, open(DOC).read(), flags=re.S)


def c_name(cls):
    return "vcr_" + re.sub(r"(?<!^)(?=[A-Z])", "_", cls).lower()       # KnnArgs -> vcr_knn_args


def structures():
    """(class name, ctypes class) of every ctypes.Structure subclass defined in the document's python blocks."""
    out = []
    for block in fenced("python"):
        try:
            tree = ast.parse(block)
        except SyntaxError:
            continue                                       # (a block with elisions is prose, not a binding)
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and any("Structure" in ast.unparse(b) for b in node.bases):
                ns = {"C": ctypes, "ctypes": ctypes}
                exec(compile(ast.Module([node], []), DOC, "exec"), ns)
                out.append((node.name, ns[node.name]))
    return out


def gcc_layout(tmp_path, cname, fields):
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {',
             f'printf("%zu\\n", sizeof({cname}));']
    lines += [f'printf("%zu\\n", offsetof({cname}, {f}));' for f in fields]
    lines.append("return 0; }")
    src = tmp_path / f"{cname}.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / cname
    r = subprocess.run(["gcc", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, f"{cname}: the documented struct names a field the header does not have\n{r.stderr}"
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    return got[0], got[1:]


def header_fields(cname):
    """Field names of a struct of the header, in order (top level only: enough to see a missing TRAILING field)."""
    hdr = open(HEADER).read()
    m = re.search(r"typedef struct[^{]*\{((?:[^{}]|\{[^{}]*\})*)\}\s*%s;" % cname, hdr)
    assert m, cname
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or "{" in decl:
            continue
        for part in decl.split(","):
            nm = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)
            if nm:
                names.append(nm[-1])
    return names


def test_documented_ctypes_structs_match_the_header(tmp_path):
    found = structures()
    assert found, "INTEGRATION.md no longer shows a ctypes binding"
    for cls, ct in found:
        cname = c_name(cls)
        fields = [f for f, _ in ct._fields_]
        size, offs = gcc_layout(tmp_path, cname, fields)
        assert ctypes.sizeof(ct) == size, (
            f"INTEGRATION.md's {cls} is {ctypes.sizeof(ct)} bytes, {cname} in include/vcr_hip.h is {size}: "
            f"the documented binding is stale (header fields: {header_fields(cname)})")
        bad = [(f, getattr(ct, f).offset, o) for f, o in zip(fields, offs) if getattr(ct, f).offset != o]
        assert not bad, (cls, bad)
        assert fields == header_fields(cname), (cls, "field order / names differ from the header")


def test_documented_abi_number_is_the_headers():
    import vcrnet_amd  # noqa: F401
    from vcrnet_amd import native
    nums = re.findall(r"vcr_abi_version\(\)\s*==\s*(\d+)", open(DOC).read())
    assert nums, "the documented binding must check vcr_abi_version()"
    assert {int(n) for n in nums} == {native.ABI_VERSION}
    assert int(re.search(r"#define VCR_ABI_VERSION (\d+)", open(HEADER).read()).group(1)) == native.ABI_VERSION


def test_documented_c_blocks_compile_against_the_header(tmp_path):
    blocks = fenced("c")
    assert blocks
    for i, block in enumerate(blocks):
        src = tmp_path / f"doc_{i}.c"
        src.write_text(block.replace('#include "vcr_hip.h"', f'#include "{HEADER}"'))
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-unused-variable", "-c", str(src), "-o",
                            str(tmp_path / f"doc_{i}.o")], capture_output=True, text=True)
        assert r.returncode == 0, f"C block {i} of INTEGRATION.md does not compile:\n{r.stderr}"


def test_a_stale_binding_is_caught(tmp_path):
    """The check itself: round 5's documented KnnArgs (it ended at `xt`) must fail against today's header."""
    class KnnArgs(ctypes.Structure):
        _fields_ = [("x", ctypes.c_void_p), ("ldx", ctypes.c_int), ("sq", ctypes.c_void_p), ("B", ctypes.c_int),
                    ("N", ctypes.c_int), ("C", ctypes.c_int), ("k", ctypes.c_int), ("idx", ctypes.c_void_p),
                    ("tie_scratch", ctypes.c_void_p), ("tie_cap", ctypes.c_int), ("waves", ctypes.c_int),
                    ("tie_zeroed", ctypes.c_int), ("tie_defer", ctypes.c_int), ("tie_work", ctypes.c_void_p),
                    ("tie_work_bytes", ctypes.c_size_t), ("tie_inline", ctypes.c_int), ("xt", ctypes.c_void_p)]
    size, _ = gcc_layout(tmp_path, "vcr_knn_args", [f for f, _ in KnnArgs._fields_])
    assert ctypes.sizeof(KnnArgs) < size
