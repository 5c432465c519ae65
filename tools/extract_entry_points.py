#!/usr/bin/env python3
"""List every `(shader, entry point)` kernel identity of the reference into tests/golden/reference_entry_points.json.

In the reference a kernel is identified by (WGSL text, @compute fn name): the pipeline-cache key of
GpuDevice::create_compute_pipeline (crates/array/src/gpu_utils/gpu_device.rs:145-168) and what every op macro hands to
apply_{unary,binary,scalar,ternary,broadcast}_function.  This script walks the WGSL files (build container only —
/root/reference is read-only and absent on the GPU box), strips comments, and records for each live `@compute` entry
point: the shader key ("<crate>/<dir>/<file>", the path under crates/ without "compute_shaders/" and ".wgsl"), the
entry-point name, its storage bindings (index, access, element type) and whether any Rust source `include_str!`s the
file (files nobody includes are dead code in the reference: `dead: true`).  Data only — no shader text is copied.

Second output (round 3): the reference does not pass a key, it passes the shader TEXT — every op crate holds
`const X_SHADER: &str = include_str!("…wgsl")` or `concat!(include_str!("…/utils.wgsl"), include_str!("…wgsl"))`
(crates/arithmetic/src/f32.rs:10-15, crates/compare/src/u8.rs:3-12).  For every such constant this script evaluates the
text the way rustc would, and records its 64-bit FNV-1a hash + byte length + the shader key of the LAST included file
(the kernel file; anything before it is a utils prelude) into
  arrow_gpu_amd/csrc/shader_hashes.inc           — the table agpu_shader_key_for_source() looks texts up in
  tests/golden/reference_shader_hashes.json      — the same with constant names and file:line, for the tests.
Hashes and lengths are data; no WGSL text is written anywhere.

Usage: python tools/extract_entry_points.py [/root/reference] [tests/golden/reference_entry_points.json]
"""
from __future__ import annotations

import json
import os
import re
import sys


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def key_of_wgsl(path: str, crates: str):
    parts = os.path.relpath(path, crates).split(os.sep)  # <crate>/compute_shaders/<dir>/<file>.wgsl
    if len(parts) == 4 and parts[1] == "compute_shaders":
        return f"{parts[0]}/{parts[2]}/{parts[3][:-5]}"
    if len(parts) == 3 and parts[1] == "compute_shaders":
        return f"{parts[0]}/-/{parts[2][:-5]}"
    return None


def shader_constants(ref: str):
    """Every `const NAME: &str = include_str!(…) | concat!(include_str!(…), …);` under crates/, evaluated."""
    crates = os.path.join(ref, "crates")
    pat = re.compile(r"(?:pub(?:\([a-z]+\))?\s+)?const\s+(\w+)\s*:\s*&(?:'static\s+)?str\s*=\s*(.*?);", re.S)
    out = []
    for dp, _, fns in sorted(os.walk(crates)):
        for fn in sorted(fns):
            if not fn.endswith(".rs"):
                continue
            path = os.path.join(dp, fn)
            src = open(path).read()
            for m in pat.finditer(src):
                incs = re.findall(r'include_str!\(\s*"([^"]+)"\s*\)', m.group(2))
                if not incs:
                    continue
                rest = re.sub(r'include_str!\(\s*"[^"]+"\s*\)', "", m.group(2))
                assert not re.sub(r"concat!\(|\)|,|\s", "", rest), (path, m.group(1), rest)  # nothing but include_str! / concat!
                files = [os.path.normpath(os.path.join(dp, i)) for i in incs]
                text = b"".join(open(f, "rb").read() for f in files)
                key = key_of_wgsl(files[-1], crates)
                assert key, files
                out.append({"const": m.group(1), "rust": f"{os.path.relpath(path, ref)}:{src[: m.start()].count(chr(10)) + 1}",
                            "includes": [os.path.relpath(f, ref) for f in files], "shader_key": key,
                            "fnv1a64": f"{fnv1a64(text):016x}", "bytes": len(text)})
    return out


def write_shader_hashes(ref: str, root: str):
    consts = shader_constants(ref)
    # One text may be shipped under two file names: compare/u32/min_max.wgsl is byte-identical to compare/i32/min_max.wgsl
    # (it declares array<i32>: the reference's u32 min / max compare as SIGNED).  The program is what the text says, so such a
    # text maps to the first key in sorted order — the i32 one — and the others are listed as aliases in the fixture.
    keys_of = {}
    for c in consts:
        keys_of.setdefault((c["fnv1a64"], c["bytes"]), set()).add(c["shader_key"])
    by_hash = {k: sorted(v)[0] for k, v in keys_of.items()}
    for c in consts:
        c["resolves_to"] = by_hash[(c["fnv1a64"], c["bytes"])]
    with open(os.path.join(root, "tests", "golden", "reference_shader_hashes.json"), "w") as f:
        json.dump({"source": "psvri/arrow-gpu crates/*/src/**/*.rs shader constants (generated by tools/extract_entry_points.py); "
                             "hash = FNV-1a 64 of the constant's text as rustc evaluates include_str! / concat!",
                   "count": len(consts), "distinct_texts": len(by_hash), "constants": consts}, f, indent=1)
    with open(os.path.join(root, "arrow_gpu_amd", "csrc", "shader_hashes.inc"), "w") as f:
        f.write("// GENERATED by tools/extract_entry_points.py from the reference's shader constants — hashes and lengths only, no WGSL.\n"
                "// {FNV-1a 64 of the text, byte length, shader key of the kernel file}  [ref: the constants listed in\n"
                "// tests/golden/reference_shader_hashes.json, e.g. crates/arithmetic/src/f32.rs:10-15, crates/compare/src/u8.rs:3-12]\n")
        for (h, n), key in sorted(by_hash.items(), key=lambda kv: (kv[1], kv[0])):
            f.write(f'{{0x{h}ull, {n}u, "{key}"}},\n')
    print(f"{len(consts)} shader constants ({len(by_hash)} distinct texts) -> shader_hashes.inc, reference_shader_hashes.json")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_entry_points.json")
    crates = os.path.join(ref, "crates")
    rust = {}
    for dp, _, fns in os.walk(crates):
        for fn in fns:
            if fn.endswith(".rs"):
                rust[os.path.join(dp, fn)] = open(os.path.join(dp, fn)).read()
    records = []
    for dp, _, fns in sorted(os.walk(crates)):
        for fn in sorted(fns):
            if not fn.endswith(".wgsl"):
                continue
            path = os.path.join(dp, fn)
            rel = os.path.relpath(path, ref)
            parts = os.path.relpath(path, crates).split(os.sep)  # <crate>/compute_shaders/<dir>/<file>.wgsl
            if len(parts) == 4 and parts[1] == "compute_shaders":
                key = f"{parts[0]}/{parts[2]}/{parts[3][:-5]}"
            elif len(parts) == 3 and parts[1] == "compute_shaders":
                key = f"{parts[0]}/-/{parts[2][:-5]}"
            else:
                continue
            text = strip_comments(open(path).read())
            bindings = []
            for m in re.finditer(r"@binding\((\d+)\)\s*var<storage,\s*(read_write|read)>\s*(\w+)\s*:\s*([^;]+);", text):
                bindings.append({"binding": int(m.group(1)), "access": m.group(2), "name": m.group(3), "type": m.group(4).strip()})
            # who includes it: include_str!("…/compute_shaders/<dir>/<file>.wgsl") anywhere under crates/
            tail = "/".join(parts[1:])
            users = sorted(os.path.relpath(p, ref) for p, s in rust.items()
                           if re.search(r'include_str!\(\s*(?:concat!\([^)]*)?"[^"]*' + re.escape(tail) + '"', s) or (tail in s and "include_str" in s))
            for m in re.finditer(r"@compute\s*@workgroup_size\((\d+)\)\s*fn\s+(\w+)", text):
                line = text[: m.start()].count("\n") + 1
                records.append({
                    "shader_key": key, "entry_point": m.group(2), "workgroup_size": int(m.group(1)),
                    "n_read": sum(1 for b in bindings if b["access"] == "read"), "bindings": bindings,
                    "dead": not users, "ref": f"{rel}:{line}", "included_by": users[:4],
                })
    records.sort(key=lambda r: (r["shader_key"], r["entry_point"]))
    with open(out, "w") as f:
        json.dump({"source": "psvri/arrow-gpu crates/**/compute_shaders/**/*.wgsl (generated by tools/extract_entry_points.py)",
                   "count": len(records), "live": sum(1 for r in records if not r["dead"]), "entry_points": records}, f, indent=1)
    print(f"{len(records)} entry points ({sum(1 for r in records if not r['dead'])} live) -> {out}")
    write_shader_hashes(ref, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


if __name__ == "__main__":
    main()
