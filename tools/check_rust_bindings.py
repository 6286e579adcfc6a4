#!/usr/bin/env python3
"""Keeps the Rust declarations of INTEGRATION.md in lock-step with include/rt_amd.h.

No Rust toolchain exists in the build image, so the `#[repr(C)]` structs and the `extern "C"` block a
maintainer would paste into src/gpu.rs cannot be compiled here. This script parses both texts and compares,
struct by struct, field names, order, element type and array length (hence size and offsets), and, function
by function, the argument count and the C type class of every argument and of the return value.
    python tools/check_rust_bindings.py        # exit status 1 and a list of differences when they disagree
"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

C_SCALARS = {"double": ("f64", 8), "float": ("f32", 4), "uint32_t": ("u32", 4), "int32_t": ("i32", 4),
             "uint64_t": ("u64", 8), "uint8_t": ("u8", 1), "int": ("i32", 4), "size_t": ("usize", 8)}
RUST_SIZES = {"f64": 8, "f32": 4, "u32": 4, "i32": 4, "u64": 8, "u8": 1, "usize": 8}
STRUCT_NAMES = {"rt_material": "RtMaterial", "rt_sphere": "RtSphere", "rt_polygon": "RtPolygon", "rt_node": "RtNode",
                "rt_scene_desc": "RtSceneDesc", "rt_camera": "RtCamera", "rt_stats": "RtStats"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_structs(header):
    """name -> [(field, rust element type or '*', array length or 0)]"""
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", strip_comments(header), flags=re.S):
        fields = []
        for decl in filter(None, (d.strip() for d in body.split(";"))):
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)", decl)
            ctype, star, names = m.group(2), m.group(3), m.group(4)
            for nm in (x.strip() for x in names.split(",")):
                ptr = bool(star) or nm.startswith("*")
                nm = nm.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                length = int(arr.group(2)) if arr else 0
                nm = arr.group(1) if arr else nm
                fields.append((nm, "*" if ptr else C_SCALARS[ctype][0], length))
        out[name] = fields
    return out


def rust_structs(text):
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", strip_comments(text), flags=re.S):
        fields = []
        for nm, ty in re.findall(r"pub (\w+):\s*([^,}]+)", body):
            ty = ty.strip()
            arr = re.match(r"\[(\w+);\s*(\d+)\]", ty)
            if ty.startswith("*"):
                fields.append((nm, "*", 0))
            elif arr:
                fields.append((nm, arr.group(1), int(arr.group(2))))
            else:
                fields.append((nm, ty, 0))
        out[name] = fields
    return out


def layout(fields):
    """(offsets, size) under the C rules for these scalar types (alignment = size, pointers 8)."""
    off, offsets, align_max = 0, [], 1
    for _, ty, length in fields:
        size = 8 if ty == "*" else RUST_SIZES[ty]
        align_max = max(align_max, size)
        off = (off + size - 1) // size * size
        offsets.append(off)
        off += size * max(length, 1)
    return offsets, (off + align_max - 1) // align_max * align_max


def c_functions(header):
    out = {}
    for ret, name, args in re.findall(r"\n(rt_status|const char \*|uint32_t)\s*(\w+)\(([^;]*?)\);", strip_comments(header)):
        kinds = []
        for a in (x.strip() for x in args.replace("\n", " ").split(",")):
            if a in ("void", ""):
                continue
            if "*" in a or "[" in a:
                kinds.append("ptr")
            else:
                ctype = a.replace("const ", "").split()[0]
                kinds.append({"rt_traverse_mode": "i32", "rt_shard_mode": "i32"}.get(ctype) or C_SCALARS[ctype][0])
        out[name] = ("ptr" if "*" in ret else {"rt_status": "i32", "uint32_t": "u32"}[ret], kinds)
    return out


def rust_functions(text):
    out = {}
    for name, args, ret in re.findall(r"(?:pub )?fn (rt\w+)\(([^)]*)\)\s*->\s*([^;{]+);", strip_comments(text), flags=re.S):
        kinds = []
        for a in filter(None, (x.strip() for x in args.replace("\n", " ").split(","))):
            ty = a.split(":", 1)[1].strip()
            kinds.append("ptr" if ty.startswith("*") else {"c_int": "i32"}.get(ty, ty))
        ret = ret.strip()
        out[name] = ("ptr" if ret.startswith("*") else {"c_int": "i32"}.get(ret, ret), kinds)
    return out


def differences():
    header = (ROOT / "include" / "rt_amd.h").read_text()
    doc = (ROOT / "INTEGRATION.md").read_text()
    rust_text = "\n".join(re.findall(r"```rust(.*?)```", doc, flags=re.S)) + "\n" + \
                "\n".join(re.findall(r"`(fn rt\w+\([^`]*;)`", doc.replace("\n  ", " "), flags=re.S))
    cs, rs = c_structs(header), rust_structs(rust_text)
    bad = []
    for cname, rname in STRUCT_NAMES.items():
        if cname not in cs or rname not in rs:
            bad.append(f"{cname} / {rname}: missing on one side")
            continue
        if cs[cname] != rs[rname]:
            bad.append(f"{cname} != {rname}:\n    C    {cs[cname]}\n    Rust {rs[rname]}")
        elif layout(cs[cname]) != layout(rs[rname]):
            bad.append(f"{cname}: layouts differ")
    cf, rf = c_functions(header), rust_functions(rust_text)
    for name, sig in rf.items():
        if name not in cf:
            bad.append(f"{name}: declared in INTEGRATION.md, not in rt_amd.h")
        elif cf[name] != sig:
            bad.append(f"{name}: C {cf[name]} != Rust {sig}")
    return bad, {n: layout(cs[n])[1] for n in STRUCT_NAMES if n in cs}, sorted(rf)


if __name__ == "__main__":
    bad, sizes, funcs = differences()
    print("struct sizes:", sizes)
    print("functions declared for Rust:", ", ".join(funcs))
    for b in bad:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)
