"""bindings/renderer_gpu is Rust source that this image cannot compile; what CAN be checked here is that it
binds the ABI that exists: every #[repr(C)] struct of src/abi.rs has the size and field offsets of its C counterpart (computed from
the Rust text by the repr(C) rules, against sizeof / offsetof printed by a C program built with gcc from the real headers, and
against the committed table abi_layout.json), and the extern "C" block declares the functions include/rt_abi.h declares with the
same argument counts."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "renderer_gpu")
PRIM = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8), "f64": (8, 8)}


def _rust_structs():
    text = open(os.path.join(CRATE, "src", "abi.rs")).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)*pub struct (\w+) \{(.*?)\n\}", text, re.S):
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2))]
        out[m.group(1)] = fields
    return out


def _layout(structs, name, memo):
    if name in memo:
        return memo[name]
    off, align, fields = 0, 1, {}
    for fname, ty in structs[name]:
        arr = re.match(r"\[(\w+); (\d+)\]", ty)
        base, count = (arr.group(1), int(arr.group(2))) if arr else (ty, 1)
        size, al = PRIM[base] if base in PRIM else (_layout(structs, base, memo)["size"], _layout(structs, base, memo)["align"])
        off = (off + al - 1) // al * al
        fields[fname] = off
        off += size * count
        align = max(align, al)
    memo[name] = {"size": (off + align - 1) // align * align, "align": align, "fields": fields}
    return memo[name]


def _c_layout(structs, tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_abi.h"', '#include "rt_checkpoint.h"', 'int main(void) {', 'printf("{");']
    for si, (name, fields) in enumerate(structs.items()):
        lines.append(f'printf("%s\\"{name}\\": {{\\"size\\": %zu, \\"fields\\": {{", "{"," if si else ""}", sizeof({name}));')
        for fi, (fname, _) in enumerate(fields):
            lines.append(f'printf("%s\\"{fname}\\": %zu", "{"," if fi else ""}", offsetof({name}, {fname}));')
        lines.append('printf("}}");')
    lines += ['printf("}\\n"); return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    return json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)


def test_repr_c_structs_match_the_c_headers(tmp_path):
    structs = _rust_structs()
    assert {"rt_node", "rt_prim", "rt_geom", "rt_tri_attrs", "rt_xform", "rt_material", "rt_texture", "rt_light", "rt_camera", "rt_blob_header",
            "rt_render_params", "rt_stats", "rt_pdf_desc", "rt_checkpoint_header"} <= set(structs)
    c = _c_layout(structs, tmp_path)
    memo = {}
    rust = {n: _layout(structs, n, memo) for n in structs}
    for name in structs:
        assert rust[name]["size"] == c[name]["size"], (name, rust[name]["size"], c[name]["size"])
        assert rust[name]["fields"] == c[name]["fields"], name
    assert (c["rt_node"]["size"], c["rt_prim"]["size"], c["rt_geom"]["size"], c["rt_tri_attrs"]["size"], c["rt_checkpoint_header"]["size"], c["rt_pdf_desc"]["size"]) == (64, 32, 80, 120, 64, 64)
    # the committed table (what a maintainer without this repository's headers diffs against)
    table = json.load(open(os.path.join(CRATE, "abi_layout.json")))
    assert table == {n: {"size": c[n]["size"], "fields": c[n]["fields"]} for n in structs}
    # and the ctypes mirrors the tests use agree with it
    import ctypes as C
    from raytracing_amd import _ffi
    assert C.sizeof(_ffi.Stats) == c["rt_stats"]["size"] and C.sizeof(_ffi.RenderParams) == c["rt_render_params"]["size"]
    assert C.sizeof(_ffi.BlobHeader) == c["rt_blob_header"]["size"] and C.sizeof(_ffi.Camera) == c["rt_camera"]["size"]
    for k, off in c["rt_stats"]["fields"].items():
        assert getattr(_ffi.Stats, k).offset == off, k


def _strip(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_extern_block_declares_the_abi():
    header = _strip(open(os.path.join(ROOT, "include", "rt_abi.h")).read())
    c_fns = {m.group(1): (0 if m.group(2).strip() in ("void", "") else m.group(2).count(",") + 1)
             for m in re.finditer(r"\b(rt_\w+)\s*\(([^()]*)\)\s*;", header)}
    rust = open(os.path.join(CRATE, "src", "abi.rs")).read()
    ext = rust[rust.index('extern "C" {'):]
    r_fns = {m.group(1): (0 if not m.group(2).strip() else len([a for a in m.group(2).split(",") if a.strip()]))
             for m in re.finditer(r"pub fn (rt_\w+)\(([^)]*)\)", ext, re.S)}
    assert len(c_fns) >= 20
    assert r_fns == c_fns, (set(c_fns) ^ set(r_fns), {k: (c_fns[k], r_fns.get(k)) for k in c_fns if r_fns.get(k) != c_fns[k]})
    # constants that select behaviour
    for name, val in (("RT_INT_MIS", 0), ("RT_INT_BRDF", 2), ("RT_BG_SKY", 1), ("RT_PRIM_INSTANCE", 4), ("RT_RENDER_LANE_WALK", 8), ("RT_BLOB_VERSION", 2)):
        c = re.search(rf"#define {name} \(?(-?\w+?)u?\)?\s", header)
        r = re.search(rf"pub const {name}: \w+ = (\w+);", rust)
        assert c and r and int(c.group(1), 0) == int(r.group(1).replace("_", ""), 0) == val, name


def test_flattener_follows_the_host_flattener():
    # same table order and the same header fields as raytracing_amd/host/flatten.cpp (the blob layout is positional)
    rs = open(os.path.join(CRATE, "src", "flatten.rs")).read()
    cpp = open(os.path.join(ROOT, "raytracing_amd", "host", "flatten.cpp")).read()
    order = ["off_nodes", "off_prims", "off_geom", "off_tri_attrs", "off_xforms", "off_materials", "off_textures", "off_texels", "off_lights"]
    for text in (rs, cpp):
        pos = [text.index("h." + k + " = ") for k in order]
        assert pos == sorted(pos)
    for hook in ("gpu_describe", "gpu_parts", "classify_background", "RT_NODE_DUP_RIGHT", "RT_NODE_NO_BOX"):
        assert hook in rs
