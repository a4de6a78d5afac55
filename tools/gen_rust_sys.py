#!/usr/bin/env python3
"""Generates bindings/rust/zkstark_amd_sys.rs from include/zkstark_amd.h.

The Rust `extern "C"` surface is derived from the header, never written by hand, so the two cannot
drift: tests/test_cabi.py regenerates the text and compares it with the committed file, and checks
that every exported function of the header is declared.  The header is written in a regular style
(one declaration per statement, no macros in signatures), which is all this parser understands.

    python tools/gen_rust_sys.py            # rewrite bindings/rust/zkstark_amd_sys.rs
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file is out of date
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zkstark_amd.h")
OUT = os.path.join(ROOT, "bindings", "rust", "zkstark_amd_sys.rs")

SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64",
           "size_t": "usize", "int": "c_int", "double": "f64", "char": "c_char", "void": "c_void"}
KEYWORDS = {"in": "in_", "type": "type_", "ref": "ref_", "box": "box_", "move": "move_", "fn": "fn_", "mod": "mod_", "match": "match_"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_type(ctype, array=False):
    """ctype: e.g. 'const uint32_t *', 'zk_ctx *const *', 'void *'.  array: a `T name[k]` parameter (decays to a pointer)."""
    base, base_const, ptr_const = None, False, []     # ptr_const[k]: the pointer OBJECT of level k is const
    for t in ctype.replace("*", " * ").split():
        if t == "const":
            if ptr_const:
                ptr_const[-1] = True
            else:
                base_const = True
        elif t == "*":
            ptr_const.append(False)
        elif t != "struct":
            base = t
    out = SCALARS.get(base, base)
    levels = len(ptr_const) + (1 if array else 0)
    for k in range(levels):                           # level k points to: the base (k = 0) or the level k-1 pointer object
        pointee_const = base_const if k == 0 else ptr_const[k - 1]
        out = ("*const " if pointee_const else "*mut ") + out
    return out


def parse_params(args):
    params = []
    args = args.strip()
    if args and args != "void":
        for a in args.split(","):
            a = " ".join(a.split())
            am = re.match(r"(.+?)\s*(\w+)\s*(\[\w*\])?$", a)
            ty, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
            params.append((KEYWORDS.get(pname, pname), rust_type(ty, array=bool(arr))))
    return params


def parse(text):
    text = strip_comments(text)
    consts, opaque, structs, fns = [], [], [], []
    for m in re.finditer(r"#define\s+(ZK_[A-Z0-9_]+)\s+(\d+)u?\b", text):
        consts.append((m.group(1), "u32", m.group(2)))
    for m in re.finditer(r"enum\s+\w+\s*\{(.*?)\}\s*;", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [x.strip() for x in item.split("=")]
                nxt = int(val)
            else:
                name = item
            consts.append((name, "c_int", str(nxt)))
            nxt += 1
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = " ".join(decl.split())
            fp = re.match(r"(.+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", decl)
            if fp:                                       # function pointer field: ret (*name)(args)
                fields.append((fp.group(2), "Option<unsafe extern \"C\" fn(" + ", ".join(f"{n}: {t}" for n, t in parse_params(fp.group(3))) + ")"
                               + (f" -> {rust_type(fp.group(1).strip())}" if fp.group(1).strip() != "void" else "") + ">"))
                continue
            fm = re.match(r"(.+?)\s*(\w+)((?:\[\d+\])*)$", decl)
            ty, name, dims = fm.group(1).strip(), fm.group(2), re.findall(r"\[(\d+)\]", fm.group(3))
            rt = rust_type(ty)
            for d in reversed(dims):
                rt = f"[{rt}; {d}]"
            fields.append((name, rt))
        structs.append((m.group(3), fields))
        if m.group(3) in opaque:
            opaque.remove(m.group(3))
    body = text[text.index('extern "C" {'):]
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
    body = re.sub(r"enum\s+\w+\s*\{.*?\}\s*;", "", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(zk_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        fns.append((name, parse_params(args), None if ret == "void" else rust_type(ret)))
    return consts, opaque, structs, fns


def render(consts, opaque, structs, fns):
    L = ["//! Raw bindings to libzkstark_amd.so: every declaration of include/zkstark_amd.h.",
         "//! GENERATED by tools/gen_rust_sys.py from the header -- do not edit; tests/test_cabi.py checks it is current.",
         "//! Not compiled in the build image (no Rust toolchain); see INTEGRATION.md.",
         "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_void};", ""]
    for o in sorted(opaque):
        L.append(f"#[repr(C)] pub struct {o} {{ _private: [u8; 0] }}")
    L.append("")
    for name, ty, val in consts:
        L.append(f"pub const {name}: {ty} = {val};")
    L.append("")
    for name, fields in structs:
        L += ["#[repr(C)]", "#[derive(Clone, Copy)]", f"pub struct {name} {{"]
        L += [f"    pub {f}: {t}," for f, t in fields]
        L += ["}", ""]
        if fields and fields[0][0] == "struct_size":      # caller-allocated struct of the ABI: ZK_STRUCT_INIT
            L += [f"impl {name} {{",
                  "    /// ZK_STRUCT_INIT: all fields zero, struct_size = size_of::<Self>() (the library checks it before every read or write).",
                  "    pub fn new() -> Self {",
                  "        let mut s: Self = unsafe { std::mem::zeroed() };",
                  "        s.struct_size = std::mem::size_of::<Self>() as u32;",
                  "        s",
                  "    }",
                  "}", ""]
    L += ['#[link(name = "zkstark_amd")]', 'extern "C" {']
    for name, params, ret in fns:
        sig = ", ".join(f"{p}: {t}" for p, t in params)
        L.append(f"    pub fn {name}({sig})" + (f" -> {ret};" if ret else ";"))
    L += ["}", ""]
    return "\n".join(L)


def generate():
    with open(HEADER) as f:
        return render(*parse(f.read()))


def declared_functions(rs_text):
    return sorted(set(re.findall(r"pub fn (zk_[a-z0-9_]+)\(", rs_text)))


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("up to date" if ok else "OUT OF DATE: run python tools/gen_rust_sys.py")
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {len(declared_functions(text))} functions")
