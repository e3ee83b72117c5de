"""The Rust facade (rust/petal-decomposition-hip) cannot be compiled in this image (no cargo / rustc), so nothing would notice
its `extern "C"` block drifting away from include/petal_hip.h.  This test parses both and checks, for every function the
facade binds: it exists in the header, the argument count agrees, and every argument / the return value has the same
machine type (pointer vs integer vs double, and the integer width: c_int <-> int / int32_t, i64 <-> int64_t).  The
`#[repr(C)]` struct PetalMatrix is checked field by field against `petal_matrix`, and the status constants by value."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "petal_hip.h")
FFI = os.path.join(ROOT, "rust", "petal-decomposition-hip", "src", "ffi.rs")


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _c_class(t):
    """machine class of a C parameter type"""
    t = t.strip()
    if "*" in t or re.search(r"\bpetal_allreduce_fn\b", t):
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void", "size_t": "u64"}[t]


def _rust_class(t):
    t = t.strip()
    if t.startswith("*"):
        return "ptr"
    return {"c_int": "i32", "i32": "i32", "i64": "i64", "f64": "f64", "c_double": "f64", "usize": "u64"}[t]


def parse_header():
    text = _strip_c_comments(open(HEADER).read())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    fns = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(petal_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "(" in ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a, flags=re.S)   # type, then the parameter name
                params.append(_c_class(mm.group(1)))
        fns[name] = (_c_class(ret), params)
    st = re.search(r"typedef\s+struct\s+petal_matrix\s*\{(.*?)\}\s*petal_matrix\s*;", text, flags=re.S).group(1)
    fields = []
    for decl in st.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(.*?)([\w\s,]+)$", decl, flags=re.S)
        ty = decl.rsplit(None, 1)[0] if "," not in decl else decl.split(None, 1)[0]
        names = decl[len(ty):].replace("*", "").split(",")
        cls = "ptr" if "*" in decl else _c_class(ty)
        fields += [(n.strip(), cls) for n in names]
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(PETAL_[A-Z_0-9]+)\s*=\s*(\d+)", text)}
    return fns, fields, consts


def parse_ffi():
    text = re.sub(r"//.*$", "", open(FFI).read(), flags=re.M)
    block = re.search(r'extern\s+"C"\s*\{(.*)\}', text, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(4)
        params = [_rust_class(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        fns[name] = ("void" if ret is None else _rust_class(ret), params)
    st = re.search(r"pub\s+struct\s+PetalMatrix\s*\{(.*?)\}", text, flags=re.S).group(1)
    fields = [(m.group(1), _rust_class(m.group(2))) for m in re.finditer(r"pub\s+(\w+)\s*:\s*([^,\n]+),", st)]
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub\s+const\s+(PETAL_\w+)\s*:\s*c_int\s*=\s*(\d+)", text)}
    return fns, fields, consts


def test_every_rust_binding_matches_the_header():
    hf, hfields, hconsts = parse_header()
    rf, rfields, rconsts = parse_ffi()
    assert len(hf) >= 20 and "petal_rpca_fit" in hf and "petal_ctx_set_collective" in hf       # the parser sees the whole header
    assert len(rf) >= 8
    for name, (ret, params) in rf.items():
        assert name in hf, f"{name} is bound in ffi.rs but not declared in petal_hip.h"
        hret, hparams = hf[name]
        assert ret == hret, (name, "return", ret, hret)
        assert len(params) == len(hparams), (name, "arity", len(params), len(hparams))
        assert params == hparams, (name, params, hparams)
    assert rfields == hfields, (rfields, hfields)
    for name, value in rconsts.items():
        assert hconsts.get(name) == value, (name, value, hconsts.get(name))


def test_header_parser_reads_known_signatures():
    hf, hfields, _ = parse_header()
    assert hf["petal_ctx_create"] == ("i32", ["i32", "ptr", "ptr"])
    assert hf["petal_rpca_fit"] == ("i32", ["ptr", "ptr", "i64", "i64", "i64", "i32", "ptr", "ptr", "ptr", "ptr", "ptr", "ptr"])
    assert hf["petal_fastica_fit"][1][3] == "f64" and hf["petal_ctx_destroy"][0] == "void"
    assert [f[0] for f in hfields] == ["data", "rows", "cols", "row_stride", "col_stride", "dtype", "space"]
    assert [f[1] for f in hfields] == ["ptr", "i64", "i64", "i64", "i64", "i32", "i32"]
