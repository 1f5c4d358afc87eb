"""The Rust binding (bindings/rust/src/lib.rs) against the C headers it binds -- without a Rust toolchain (the image has none).

north_star keeps the host in Rust; the crate under bindings/rust cannot be compiled here, so it is held to the headers by
parsing both sides:
  * every `extern "C"` declaration: the function exists in include/mp3rgain_amd.h / mp3rgain_amd_node.h with the same arity,
    every argument type and the return type translate to the C ones (c_int <-> int, usize <-> size_t, `*const T` <-> `const T *`...);
  * every prototype of the two headers is bound (nothing silently missing);
  * every #[repr(C)] struct with fields: same field names in the same order as the C struct, the same widths, and the
    offsets / size #[repr(C)] gives them (computed here by C's layout rules from the Rust types) equal the offsetof / sizeof a C
    program compiled from the headers prints;
  * the constants the binding restates (ABI version, status codes, flags) equal the headers'.
The reference signatures the wrappers mirror: src/replaygain.rs:57-95, 929-941, 1033-1047, 1119-1132, 1140."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
RUST = ROOT / "bindings" / "rust" / "src" / "lib.rs"
HEADERS = [ROOT / "include" / "mp3rgain_amd.h", ROOT / "include" / "mp3rgain_amd_node.h"]

# Rust scalar -> (C spelling, size, alignment) on the LP64 target the library is built for
SCALARS = {
    "c_int": ("int", 4, 4), "u8": ("uint8_t", 1, 1), "u16": ("uint16_t", 2, 2), "i16": ("int16_t", 2, 2), "u32": ("uint32_t", 4, 4),
    "i32": ("int32_t", 4, 4), "u64": ("uint64_t", 8, 8), "i64": ("int64_t", 8, 8), "usize": ("size_t", 8, 8), "f64": ("double", 8, 8),
    "f32": ("float", 4, 4), "c_char": ("char", 1, 1), "c_void": ("void", 0, 1),
}
STRUCTS = {"RgCtx": "rg_ctx", "RgNode": "rg_node", "RgNodeBackend": "rg_node_backend", "RgTrackDesc": "rg_track_desc",
           "RgTrackResult": "rg_track_result", "RgAlbumResult": "rg_album_result", "RgPeakResult": "rg_peak_result",
           "RgDeviceView": "rg_device_view", "RgWavInfo": "rg_wav_info"}


def _strip_c(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"^\s*#.*$", "", src, flags=re.M)


def _norm_c_type(t: str) -> str:
    return " ".join(t.replace("*", " * ").split())


def c_prototypes():
    out = {}
    for h in HEADERS:
        s = _strip_c(h.read_text())
        for ret, name, args in re.findall(r"^([A-Za-z_][\w\s\*]*?[\s\*])(rg_\w+)\s*\(([^;{}]*?)\)\s*;", s, flags=re.M | re.S):
            args = " ".join(args.split())
            types = []
            if args != "void":
                for a in args.split(","):
                    m = re.match(r"(.*?)(\w+)$", a.strip())
                    types.append((m.group(2), _norm_c_type(m.group(1))))
            out[name] = (_norm_c_type(ret), types)
    return out


def c_structs():
    out = {}
    for h in HEADERS:
        s = _strip_c(h.read_text())
        for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{([^{}]*)\}\s*(\w+)\s*;", s, flags=re.S):
            fields = []
            if "(" in body:  # a table of function pointers (rg_node_backend: a test seam, bound as an opaque type)
                continue
            for decl in body.split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                m = re.match(r"(.*?)(\w+)$", decl)
                fields.append((m.group(2), _norm_c_type(m.group(1))))
            out[name] = fields
    return out


def rust_type_to_c(t: str) -> str:
    """`*const *const c_char` -> `const char * const *` (normalised like _norm_c_type)."""
    t = t.strip()
    quals = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        assert m, t
        quals.append(m.group(1))
        t = m.group(2)
    base = SCALARS[t][0] if t in SCALARS else STRUCTS[t]
    # quals are outermost first: *const *const c_char = pointer (to const) pointer (to const) char
    c = base
    for i, q in enumerate(reversed(quals)):
        if i == 0:
            c = ("const " if q == "const" else "") + c + " *"
        else:
            c = c + (" const" if q == "const" else "") + " *"
    return _norm_c_type(c)


def rust_source():
    return RUST.read_text()


def rust_externs():
    src = rust_source()
    block = re.search(r'extern "C" \{(.*?)\n    \}', src, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", "", block)
    out = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            pname, ptype = a.split(":", 1)
            params.append((pname.strip(), ptype.strip()))
        out[name] = (ret.strip() if ret else None, params)
    return out


def rust_structs():
    src = rust_source()
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+) \{(.*?)\n    \}", src, flags=re.S):
        body = re.sub(r"//[^\n]*", "", body)
        fields = [(n, t.strip()) for n, t in re.findall(r"(?:pub\s+)?(\w+):\s*([^,]+),", body)]
        out[name] = fields
    return out


def test_every_extern_declaration_matches_its_prototype():
    protos, ext = c_prototypes(), rust_externs()
    assert len(ext) >= 60
    for name, (ret, params) in ext.items():
        assert name in protos, f"{name}: declared in the binding, not in the headers"
        c_ret, c_params = protos[name]
        assert len(params) == len(c_params), f"{name}: {len(params)} arguments in Rust, {len(c_params)} in C"
        for (rn, rt), (cn, ct) in zip(params, c_params):
            assert rust_type_to_c(rt) == ct, f"{name}({rn}): Rust {rt} is C `{rust_type_to_c(rt)}`, the header says `{ct}`"
            assert rn == cn, f"{name}: argument named {rn} in Rust, {cn} in C"
        if ret is None:
            assert c_ret == "void", f"{name}: returns {c_ret} in C, nothing in Rust"
        else:
            assert rust_type_to_c(ret) == c_ret, f"{name}: returns {ret} in Rust, `{c_ret}` in C"


def test_every_prototype_of_the_two_headers_is_bound():
    missing = sorted(set(c_prototypes()) - set(rust_externs()))
    assert not missing, f"prototypes without a Rust declaration: {missing}"


def _layout(fields):
    """C / #[repr(C)] layout of a struct of scalars and pointers: [(name, offset, size)], total size."""
    off, align_max, out = 0, 1, []
    for name, t in fields:
        if t.startswith("*"):
            size, align = 8, 8
        else:
            m = re.match(r"\[(\w+);\s*(\d+)\]", t)
            if m:
                _, s1, align = SCALARS[m.group(1)]
                size = s1 * int(m.group(2))
            else:
                _, size, align = SCALARS[t]
        off = (off + align - 1) // align * align
        out.append((name, off, size))
        off += size
        align_max = max(align_max, align)
    return out, (off + align_max - 1) // align_max * align_max


def test_repr_c_structs_have_the_c_structs_fields_and_layout(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    cs, rs = c_structs(), rust_structs()
    checked = 0
    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "mp3rgain_amd.h"', '#include "mp3rgain_amd_node.h"', "int main(void) {"]
    for rname, fields in rs.items():
        if len(fields) == 1 and fields[0][0] == "_p":  # opaque handles
            continue
        cname = STRUCTS[rname]
        assert cname in cs, f"{rname}: no C struct {cname}"
        cf = cs[cname]
        assert [n for n, _ in fields] == [n for n, _ in cf], f"{rname}: field names / order {fields} vs {cf}"
        for (n, rt), (_, ct) in zip(fields, cf):
            assert rust_type_to_c(rt) == ct, f"{rname}.{n}: Rust {rt}, C `{ct}`"
        prog.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        for n, _ in fields:
            prog.append(f'    printf("{cname}.{n} %zu %zu\\n", offsetof({cname}, {n}), sizeof((({cname} *)0)->{n}));')
        checked += 1
    prog += ["    return 0;", "}"]
    assert checked >= 6
    (tmp_path / "layout.c").write_text("\n".join(prog))
    subprocess.run([cc, "-I", str(ROOT / "include"), str(tmp_path / "layout.c"), "-o", str(tmp_path / "layout")], check=True)
    lines = subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout.split("\n")
    c_sizes, c_fields = {}, {}
    for l in lines:
        p = l.split()
        if len(p) == 2:
            c_sizes[p[0]] = int(p[1])
        elif len(p) == 3:
            c_fields[p[0]] = (int(p[1]), int(p[2]))
    for rname, fields in rs.items():
        if len(fields) == 1 and fields[0][0] == "_p":
            continue
        cname = STRUCTS[rname]
        lay, size = _layout(fields)
        assert size == c_sizes[cname], f"{rname}: {size} bytes under #[repr(C)], {c_sizes[cname]} in C"
        for n, off, sz in lay:
            assert (off, sz) == c_fields[f"{cname}.{n}"], f"{rname}.{n}: offset/size {(off, sz)} vs C {c_fields[f'{cname}.{n}']}"


def test_constants_of_the_binding_equal_the_headers():
    src = rust_source()
    hdr = "".join(h.read_text() for h in HEADERS)
    rust_consts = {n: v for n, v in re.findall(r"pub const (RG_\w+):\s*\w+\s*=\s*(-?\d+);", src)}
    assert len(rust_consts) >= 20
    c_vals = {n: v for n, v in re.findall(r"#define\s+(RG_\w+)\s+(-?\d+)u?\b", hdr)}
    c_vals.update({n: v for n, v in re.findall(r"\b(RG_\w+)\s*=\s*(-?\d+)", _strip_c(hdr))})
    for n, v in rust_consts.items():
        assert n in c_vals, f"{n}: not a constant of the headers"
        assert int(v) == int(c_vals[n]), f"{n}: {v} in Rust, {c_vals[n]} in C"


def test_the_wrappers_carry_the_reference_api_names():
    src = rust_source()
    for sig in ("pub fn analyze_track(file_path: &Path) -> Result<ReplayGainResult>",
                "pub fn analyze_track_with_index(file_path: &Path, track_index: Option<u32>) -> Result<ReplayGainResult>",
                "pub fn analyze_album(files: &[&Path]) -> Result<AlbumGainResult>",
                "pub fn analyze_album_with_index(files: &[&Path], track_index: Option<u32>) -> Result<AlbumGainResult>",
                "pub fn find_peak_amplitude(file_path: &Path) -> Result<PeakAmplitudeResult>",
                "pub fn is_available() -> bool",
                "pub fn gain_steps(&self) -> i32",
                "pub fn album_gain_steps(&self) -> i32"):
        assert sig in src, sig
    # braces balance (the cheapest syntax check available without rustc)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', re.sub(r"//[^\n]*", "", src))
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")") and code.count("[") == code.count("]")
