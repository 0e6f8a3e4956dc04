"""ffi/rtm-sys/src/lib.rs against include/rt_mi355x.h.  No Rust toolchain exists
here, so the crate cannot be compiled; this test parses both files and requires: the same structs with the same field
names, order and types -- hence the same sizes and offsets, computed with ctypes for the C side and by the repr(C) rules
for the Rust side --, the same functions with the same argument and return types, the same constants; and that
flat_scene.rs assigns every field of rtm_scene, rtm_model, rtm_camera and rtm_params."""
import ctypes as C
import os
import re

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "rt_mi355x.h")
LIB_RS = os.path.join(ROOT, "ffi", "rtm-sys", "src", "lib.rs")
FLAT_RS = os.path.join(ROOT, "ffi", "raytracer-rs", "flat_scene.rs")

C_SCALARS = {"double": ("f64", 8), "float": ("f32", 4), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8), "uint8_t": ("u8", 1),
             "int": ("c_int", 4), "long long": ("c_longlong", 8), "char": ("c_char", 1), "void": ("c_void", 0)}
RUST_SIZES = {"f64": 8, "f32": 4, "u32": 4, "u64": 8, "u8": 1, "c_int": 4, "c_longlong": 8, "c_char": 1}


def strip_c(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_type_to_rust(t, structs):
    """'const double*' -> '*const f64', 'rtm_scene_dev**' -> '*mut *mut rtm_scene_dev', ..."""
    t = t.strip()
    m = re.fullmatch(r"(const\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s*((?:\*\s*(?:const\s*)?)*)", t)
    assert m, t
    const_base, base, stars = bool(m.group(1)), m.group(2).strip(), m.group(3)
    rust = C_SCALARS[base][0] if base in C_SCALARS else base
    assert base in C_SCALARS or base in structs, base
    ptrs = re.findall(r"\*\s*(const)?", stars)
    for i, pc in enumerate(ptrs):
        pointee_const = const_base if i == 0 else bool(ptrs[i - 1])
        rust = ("*const " if pointee_const else "*mut ") + rust
    return rust


def parse_header():
    text = strip_c(open(HDR).read())
    structs, opaque = {}, set(re.findall(r"typedef struct (\w+) \1;", text))
    for name, body in re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.fullmatch(r"(const\s+)?([A-Za-z_][A-Za-z0-9_]*)\s*(.*)", decl)
            const, base, rest = m.group(1) or "", m.group(2), m.group(3)
            for item in rest.split(","):
                item = item.strip()
                mm = re.fullmatch(r"(\**)\s*(\w+)(?:\[(\d+)\])?", item)
                assert mm, (name, decl)
                stars, fname, arr = mm.group(1), mm.group(2), mm.group(3)
                ty = c_type_to_rust(const + base + stars, set(structs) | opaque)
                fields.append((fname, f"[{ty}; {arr}]" if arr else ty))
        structs[name] = fields
    funcs = {}
    for ret, name, args in re.findall(r"\n((?:const )?[A-Za-z_][A-Za-z0-9_ ]*?\**)\s*(rtm_\w+)\(([^)]*)\);", text):
        ret = " ".join(ret.split())
        alist = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.fullmatch(r"(.*?)(\w+)(\[\d*\])?", a)
                ty = mm.group(1).strip()
                if mm.group(3):              # an array parameter decays to a pointer to its element
                    ty = ty + "*"
                alist.append((mm.group(2), c_type_to_rust(ty, set(structs) | opaque)))
        funcs[name] = (None if ret == "void" else c_type_to_rust(ret, set(structs) | opaque), alist)
    consts = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define (RTM_\w+) (0x[0-9A-Fa-f]+u?|\d+u?)\b", text)}
    consts.update({k: int(v) for k, v in re.findall(r"(RTM_(?:OK|ERR_\w+)) = (-?\d+)", text)})
    return structs, opaque, funcs, consts


def parse_rust():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    structs, opaque = {}, set()
    for attrs, name, body in re.findall(r"((?:#\[[^\]]*\]\s*)+)pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        assert "repr(C)" in attrs, f"{name} is not #[repr(C)]"
        fields = [(f, " ".join(t.split())) for f, t in re.findall(r"(?:pub )?(\w+): ([^,\n]+),", body)]
        if fields == [("_private", "[u8; 0]")]:
            opaque.add(name)
        else:
            structs[name] = fields
    ext = re.search(r'extern "C" \{(.*)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", ext, flags=re.S):
        alist = [tuple(" ".join(x.split()) for x in a.split(":", 1)) for a in args.split(",") if a.strip()]
        funcs[name] = (" ".join(ret.split()) if ret else None, alist)
    consts = {k: int(v.replace("_", ""), 0) for k, v in re.findall(r"pub const (RTM_\w+): \w+ = (-?[0-9A-Fa-fx_]+);", text)}
    return structs, opaque, funcs, consts


def layout(fields, structs):
    """repr(C) / SysV x86-64: (offsets, size, align) of a field list."""
    def size_align(ty):
        m = re.fullmatch(r"\[(.+); (\d+)\]", ty)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if ty.startswith("*"):
            return 8, 8
        if ty in structs:
            _, s, a = layout(structs[ty], structs)
            return s, a
        return RUST_SIZES[ty], RUST_SIZES[ty]
    off, offsets, align = 0, [], 1
    for _, ty in fields:
        s, a = size_align(ty)
        off = (off + a - 1) // a * a
        offsets.append(off)
        off += s
        align = max(align, a)
    return offsets, (off + align - 1) // align * align, align


def test_structs_match_the_header_and_ctypes(rtm):
    hs, ho, _, _ = parse_header()
    rs, ro, _, _ = parse_rust()
    assert set(hs) == set(rs) and ho == ro
    for name in hs:
        assert hs[name] == rs[name], f"{name}: fields differ\n header {hs[name]}\n rust   {rs[name]}"
    # ... and the layout those declarations give is the one the shared library was compiled with (ctypes mirrors it)
    mirrors = {"rtm_camera": rtm.RtmCamera, "rtm_model": rtm.RtmModel, "rtm_scene": rtm.RtmScene, "rtm_params": rtm.RtmParams,
               "rtm_stats": rtm.RtmStats, "rtm_profile": rtm.RtmProfile}
    assert set(mirrors) == set(rs)
    for name, ct in mirrors.items():
        offsets, size, _ = layout(rs[name], rs)
        assert size == C.sizeof(ct), name
        assert [f for f, _ in rs[name]] == [f[0] for f in ct._fields_], name
        assert offsets == [getattr(ct, f[0]).offset for f in ct._fields_], name


def test_functions_and_constants_match_the_header(rtm):
    _, _, hf, hc = parse_header()
    _, _, rf, rc = parse_rust()
    assert set(hf) == set(rf) == set(rtm.EXPORTED_SYMBOLS)
    for name in hf:
        assert hf[name][0] == rf[name][0], f"{name}: return type {hf[name][0]} vs {rf[name][0]}"
        assert [t for _, t in hf[name][1]] == [t for _, t in rf[name][1]], f"{name}: argument types\n {hf[name][1]}\n {rf[name][1]}"
        assert [n for n, _ in hf[name][1]] == [n for n, _ in rf[name][1]], f"{name}: argument names"
    assert hc == rc, (hc, rc)


def test_flat_scene_fills_every_field():
    rs, _, _, _ = parse_rust()
    text = open(FLAT_RS).read()
    for struct in ("rtm_scene", "rtm_model", "rtm_camera", "rtm_params"):
        m = re.search(r"(?<!-> )\b" + struct + r" \{(.*?)\n\s*\}", text, flags=re.S)  # the struct literal, not a return type
        assert m, struct
        assigned = set(re.findall(r"^\s*(\w+)(?::|,)", m.group(1), flags=re.M))
        assert assigned == {f for f, _ in rs[struct]}, (struct, assigned ^ {f for f, _ in rs[struct]})
    # the accessors flat_scene.rs calls exist in the files to be appended to the reference's modules
    acc = open(os.path.join(ROOT, "ffi", "raytracer-rs", "camera_accessors.rs")).read() + \
        open(os.path.join(ROOT, "ffi", "raytracer-rs", "model_accessors.rs")).read()
    for fn in ("position", "forward", "up", "right", "zoom", "vertices", "n_triangles", "offset_array", "triangle_indices"):
        assert re.search(r"pub fn " + fn + r"\(&self\)", acc), fn
        assert "." + fn + "()" in text, fn
