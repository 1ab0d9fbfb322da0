"""The Rust binding shipped as source (bindings/rust/src/gpu.rs) must mirror include/rtc_hip.h exactly: every
#[repr(C)] struct field for field (name, order, type, size), every extern function (name, arity, argument types),
every constant.  No Rust toolchain exists in the build image, so this parser is the compile check the file gets;
it also checks the ctypes mirror (rtc_amd/abi.py) against the same header, and that INTEGRATION.md quotes no struct
of its own (round 1 shipped a stale RtcStats there)."""
import ctypes as C
import os
import re

from rtc_amd import abi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "rtc_hip.h")
GPU_RS = os.path.join(REPO, "bindings", "rust", "src", "gpu.rs")

C_SCALARS = {"double": ("f64", 8), "float": ("f32", 4), "int32_t": ("i32", 4), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8),
             "uint8_t": ("u8", 1), "int": ("c_int", 4), "void": ("c_void", 0), "char": ("c_char", 1)}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_structs():
    """{name: [(field, rust_type, size, align)]} from the typedef struct blocks of the header."""
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", _strip_c(open(HEADER).read()), flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl, flags=re.S)
            const, base, ptr, rest = m.groups()
            for item in [x.strip() for x in rest.split(",")]:
                star = ptr or ("*" if item.startswith("*") else "")
                item = item.lstrip("* ")
                am = re.match(r"(\w+)\s*\[(\d+)\]$", item)
                rbase = C_SCALARS[base][0] if base in C_SCALARS else base
                if star:
                    fields.append((item, f"*const {rbase}" if const else f"*mut {rbase}", 8, 8))
                elif am:
                    n = int(am.group(2))
                    fields.append((am.group(1), f"[{rbase}; {n}]", C_SCALARS[base][1] * n, C_SCALARS[base][1]))
                else:
                    fields.append((item, rbase, C_SCALARS[base][1], C_SCALARS[base][1]))
        out[name] = fields
    return out


def rust_structs():
    text = re.sub(r"//[^\n]*", "", open(GPU_RS).read())
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)*pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for decl in [d.strip() for d in re.split(r",\s*\n|,\s*$", body) if d.strip()]:
            m = re.match(r"pub\s+(\w+)\s*:\s*(.+)$", decl.strip().rstrip(","), flags=re.S)
            assert m, (name, decl)
            fields.append((m.group(1), re.sub(r"\s+", " ", m.group(2).strip())))
        out[name] = fields
    return out


def _layout(fields):
    """C layout rules: each field at the next multiple of its alignment; size rounded to the largest alignment."""
    off, amax = 0, 1
    for _n, _t, size, align in fields:
        off = (off + align - 1) // align * align + size
        amax = max(amax, align)
    return (off + amax - 1) // amax * amax


def test_every_repr_c_struct_matches_the_header_field_for_field():
    cs, rs = c_structs(), rust_structs()
    assert set(cs) == {"RtcPattern", "RtcMaterial", "RtcSceneDesc", "RtcCameraDesc", "RtcRenderOpts", "RtcStats"}
    assert set(rs) == set(cs), (sorted(rs), sorted(cs))
    for name, cfields in cs.items():
        rfields = rs[name]
        assert [f[0] for f in cfields] == [f[0] for f in rfields], f"{name}: field names / order differ"
        for (cn, ctype, _s, _a), (rn, rtype) in zip(cfields, rfields):
            assert ctype == rtype, f"{name}.{cn}: header says {ctype}, gpu.rs says {rtype}"


def test_struct_sizes_agree_between_header_rust_and_ctypes():
    cs = c_structs()
    for name, fields in cs.items():
        mirror = getattr(abi, name)
        assert [f[0] for f in fields] == [f[0] for f in mirror._fields_], f"abi.{name}: field names / order differ"
        assert _layout(fields) == C.sizeof(mirror), f"{name}: header layout {_layout(fields)} B, ctypes {C.sizeof(mirror)} B"
        for (fname, _t, size, _a), (_mn, mtype) in zip(fields, mirror._fields_):
            assert C.sizeof(mtype) == size, f"{name}.{fname}"
    assert C.sizeof(abi.RtcStats) == 88 and _layout(cs["RtcStats"]) == 88  # 8 counters + 3 times


def _c_prototypes():
    text = _strip_c(open(HEADER).read())
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(rtc_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = [a.strip() for a in args.split(",")] if args.strip() != "void" else []
        protos[name] = (ret.strip(), args)
    return protos


def _rust_of_c(decl):
    m = re.match(r"(const\s+)?(struct\s+)?(\w+)\s*((?:\*\s*)*)\s*(\w*)$", decl.strip())
    const, _st, base, stars, _name = m.groups()
    r = C_SCALARS[base][0] if base in C_SCALARS else base
    n = stars.count("*")
    if n == 0:
        return r
    if n == 1:
        return f"*const {r}" if const else f"*mut {r}"
    return f"*mut *mut {r}"


def test_extern_block_declares_every_function_with_the_header_signature():
    protos = _c_prototypes()
    assert sorted(protos) == sorted(abi.ABI_SYMBOLS)
    text = re.sub(r"//[^\n]*", "", open(GPU_RS).read())
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    rust = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        rust[name] = ([re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if a.strip()], (ret or "").strip())
    assert sorted(rust) == sorted(protos)
    for name, (cret, cargs) in protos.items():
        rargs, rret = rust[name]
        assert len(rargs) == len(cargs), name
        for ca, ra in zip(cargs, rargs):
            assert _rust_of_c(ca) == ra, f"{name}: header `{ca}` vs gpu.rs `{ra}`"
        want = {"void": "", "int": "c_int", "RtcStatus": "c_int", "uint32_t": "u32", "const char*": "*const c_char"}[re.sub(r"\s*\*", "*", cret)]
        assert rret == want, f"{name}: return {cret} vs {rret}"


def test_constants_agree():
    header = _strip_c(open(HEADER).read())
    rs = open(GPU_RS).read()
    consts = dict(re.findall(r"pub const (\w+): \w+ = (-?\d+);", rs))
    for name, val in re.findall(r"\b(RTC_(?:OK|ERR_\w+|SPHERE|PLANE|CUBE|CYLINDER|CONE|TRIANGLE|PATTERN_\w+))\s*=\s*(-?\d+)", header):
        assert consts.get(name) == val, name
        if hasattr(abi, name):
            assert getattr(abi, name) == int(val), name
    for name, val in re.findall(r"#define (RTC_FLAG_\w+|RTC_ABI_VERSION) (\d+)u?", header):
        assert consts.get(name) == val, name
        assert getattr(abi, name) == int(val), name


def test_integration_md_points_at_the_binding_instead_of_quoting_structs():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert "#[repr(C)]" not in md, "INTEGRATION.md must not carry its own copy of the struct definitions"
    assert "bindings/rust/src/gpu.rs" in md and "bindings/rust/build.rs" in md
