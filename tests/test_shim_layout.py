"""CPU: the Rust binding of the shim (shim/src/sys.rs, uncompiled here -- no cargo) is pinned on include/exon_hip.h.

Every `#[repr(C)]` struct of sys.rs is parsed, its C layout (field order, offsets, size) computed from the Rust field
types, and compared with what gcc says about the header (`offsetof` / `sizeof` dump); every `extern "C"` function must be
declared in the header with the same number of parameters, and every constant must equal the header's #define."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS_RS = open(os.path.join(ROOT, "shim", "src", "sys.rs")).read()
HEADER = os.path.join(ROOT, "include", "exon_hip.h")

PRIM = {"i32": (4, 4), "u32": (4, 4), "i64": (8, 8), "u64": (8, 8), "f64": (8, 8), "f32": (4, 4), "u8": (1, 1), "i8": (1, 1),
        "c_char": (1, 1), "c_int": (4, 4), "usize": (8, 8)}


def rust_type_layout(t):
    t = t.strip()
    if t.startswith("*"):
        return 8, 8
    m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", t)
    if m:
        s, a = PRIM[m.group(1)]
        return s * int(m.group(2)), a
    return PRIM[t]


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{([^}]*)\}", SYS_RS):
        name, body = m.group(1), m.group(2)
        fields = re.findall(r"pub (\w+):\s*([^,\n]+),", body)
        if not fields:
            continue  # opaque handles
        off, max_a, lay = 0, 1, []
        for fname, ftype in fields:
            s, a = rust_type_layout(ftype)
            off = (off + a - 1) // a * a
            lay.append((fname, off, s))
            off += s
            max_a = max(max_a, a)
        out[name] = (lay, (off + max_a - 1) // max_a * max_a)
    return out


def test_repr_c_structs_match_the_header(tmp_path):
    structs = rust_structs()
    assert set(structs) == {"exon_hip_plan_desc", "exon_hip_scan_options", "exon_hip_column", "exon_hip_device_info"}
    src = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HEADER}"', "int main(void) {"]
    for name, (lay, _size) in structs.items():
        src.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _, _ in lay:
            src.append(f'  printf("{name}.{fname} %zu %zu\\n", offsetof({name}, {fname}), sizeof((({name}*)0)->{fname}));')
    src += ["  return 0;", "}"]
    c = tmp_path / "dump.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "dump"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", str(c), "-o", str(exe)])  # a missing field fails to compile
    got = dict(line.split(" ", 1) for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, (lay, size) in structs.items():
        assert int(got[name]) == size, name
        for fname, off, fsize in lay:
            assert got[f"{name}.{fname}"] == f"{off} {fsize}", f"{name}.{fname}"
    # and the header has no field the Rust struct lacks: sizes match and the last Rust field ends where padding may begin
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (lay, _) in structs.items():
        body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", h, re.S).group(1)
        c_fields = [f for decl in body.split(";") if decl.strip()
                    for f in re.findall(r"(\w+)\s*(?:\[\d+\])?\s*(?:,|$)", decl.split(None, 1)[1].replace("*", " ").strip())]
        c_fields = [f for f in c_fields if f not in ("const", "char", "uint8_t", "int32_t", "void")]
        assert c_fields == [f for f, _, _ in lay], name


def _params(sig):
    sig = sig.strip()
    return 0 if sig in ("", "void") else sig.count(",") + 1


def test_extern_functions_and_constants_match_the_header():
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {m.group(1): _params(m.group(2)) for m in re.finditer(r"\b(exon_hip_\w+)\s*\(([^;{]*?)\)\s*;", h, re.S)}
    block = SYS_RS[SYS_RS.index('extern "C" {'):]
    rust = {m.group(1): _params(m.group(2)) for m in re.finditer(r"pub fn (exon_hip_\w+)\(([^)]*)\)", block)}
    assert len(rust) >= 25
    for name, n in rust.items():
        assert name in protos, f"{name} is bound in sys.rs but not declared in include/exon_hip.h"
        assert protos[name] == n, f"{name}: {n} parameters in sys.rs, {protos[name]} in the header"
    defines = dict(re.findall(r"#define (EXON_HIP_\w+)\s+\(?(-?\d+)\)?\s*$", h, re.M))
    consts = dict(re.findall(r"pub const (EXON_HIP_\w+): i32 = (-?\d+);", SYS_RS))
    consts.pop("EXON_HIP_ABI_VERSION")
    assert len(consts) >= 15
    for k, v in consts.items():
        assert defines[k] == v, k
    import exon_amd
    assert int(re.search(r"pub const EXON_HIP_ABI_VERSION: i32 = (\d+);", SYS_RS).group(1)) == exon_amd.load().exon_hip_abi_version()


def test_rule_and_exec_are_written():
    """the planner-side half must exist, not just be named."""
    lib = open(os.path.join(ROOT, "shim", "src", "lib.rs")).read()
    rule = open(os.path.join(ROOT, "shim", "src", "rule.rs")).read()
    assert "impl PhysicalOptimizerRule for GpuFilterAggRule" in rule and "pub fn try_new(" in lib
    assert "impl ExecutionPlan for GpuFilterAggExec" in lib and "AggregateMode::Partial" in rule
    for sym in re.findall(r"sys::(exon_hip_\w+)\(", lib + rule):
        assert f"pub fn {sym}(" in SYS_RS, f"{sym} used by the shim but not bound in sys.rs"
