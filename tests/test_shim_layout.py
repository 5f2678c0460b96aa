"""CPU: the Rust binding of the shim (shim/src/sys.rs, uncompiled here -- no cargo) is pinned on include/exon_hip.h.

Every `#[repr(C)]` struct of sys.rs is parsed, its C layout (field order, offsets, size) computed from the Rust field
types, and compared with what gcc says about the header (`offsetof` / `sizeof` dump); every `extern "C"` function must be
declared in the header with the same number of parameters, and every constant must equal the header's #define."""
import os
import re
import subprocess

import pytest

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


C_BASE = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "char": "c_char",
          "void": "c_void", "size_t": "usize", "double": "f64", "float": "f32", "struct ArrowArray": "FFI_ArrowArray",
          "struct ArrowSchema": "FFI_ArrowSchema"}
RUST_BASE = {"c_int": "i32", "c_uint": "u32"}


def c_type(decl, is_param):
    """canonical form of a C parameter / return type: '<ptr levels, outermost first> base', e.g. 'mut const c_char'"""
    d = " ".join(decl.replace("*", " * ").split())
    const = d.startswith("const ")
    if const:
        d = d[6:]
    toks = d.split()
    n_ptr = toks.count("*")
    toks = [t for t in toks if t != "*"]
    if is_param and len(toks) > (2 if toks[0] == "struct" else 1):
        toks = toks[:-1]  # the parameter's name
    base = " ".join(toks)
    base = C_BASE.get(base, base)
    levels = ["mut"] * n_ptr
    if const and n_ptr:
        levels[-1] = "const"
    return " ".join(levels + [base])


def rust_type(t):
    t = t.strip()
    levels = []
    while t.startswith("*"):
        kind, t = t[1:].split(None, 1)
        levels.append(kind)
        t = t.strip()
    return " ".join(levels + [RUST_BASE.get(t, t)])


def split_params(sig):
    sig = sig.strip()
    return [] if sig in ("", "void") else [p.strip() for p in sig.split(",")]


def test_extern_functions_and_constants_match_the_header():
    """Every `extern "C"` function of sys.rs: same return type and the same parameter TYPES, position by position, as the
    prototype in include/exon_hip.h (pointer depth, const-ness of the pointee, integer width) -- not just the count."""
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {m.group(2): (m.group(1), split_params(m.group(3)))
              for m in re.finditer(r"(?m)^\s*([\w\s\*]+?)\s*\b(exon_hip_\w+)\s*\(([^;{]*?)\)\s*;", h, re.S)}
    block = SYS_RS[SYS_RS.index('extern "C" {'):]
    rust = {m.group(1): (m.group(3) or "", split_params(m.group(2)))
            for m in re.finditer(r"pub fn (exon_hip_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block)}
    assert len(rust) >= 27
    for name, (ret, params) in rust.items():
        assert name in protos, f"{name} is bound in sys.rs but not declared in include/exon_hip.h"
        c_ret, c_params = protos[name]
        assert len(c_params) == len(params), f"{name}: {len(params)} parameters in sys.rs, {len(c_params)} in the header"
        assert rust_type(ret) == c_type(c_ret, False), f"{name}: returns {ret!r} in sys.rs, {c_ret!r} in the header"
        for i, (rp, cp) in enumerate(zip(params, c_params)):
            rt = rust_type(rp.split(":", 1)[1])
            assert rt == c_type(cp, True), f"{name} parameter {i}: {rp!r} in sys.rs vs {cp!r} in the header"
    defines = dict(re.findall(r"#define (EXON_HIP_\w+)\s+\(?(-?\d+)\)?\s*$", h, re.M))
    consts = dict(re.findall(r"pub const (EXON_HIP_\w+): i32 = (-?\d+);", SYS_RS))
    consts.pop("EXON_HIP_ABI_VERSION")
    assert len(consts) >= 15
    for k, v in consts.items():
        assert defines[k] == v, k
    import exon_amd
    assert int(re.search(r"pub const EXON_HIP_ABI_VERSION: i32 = (\d+);", SYS_RS).group(1)) == exon_amd.load().exon_hip_abi_version()


def test_type_canonicalisation_catches_a_wrong_binding():
    assert c_type("const char** name", True) == "mut const c_char" == rust_type("*mut *const c_char")
    assert c_type("struct ArrowArray* batch", True) == "mut FFI_ArrowArray" == rust_type("*mut FFI_ArrowArray")
    assert c_type("const exon_hip_plan_desc* desc", True) == rust_type("*const exon_hip_plan_desc")
    assert c_type("int64_t** d_i64", True) == rust_type("*mut *mut i64")
    assert c_type("int32_t partition", True) == rust_type("i32") == rust_type("c_int")
    assert c_type("const uint8_t* id128", True) != rust_type("*mut u8")
    assert c_type("int64_t n", True) != rust_type("i32")


# ---- `use exon::...` paths of the shim against the reference's module visibility ------------------------------------
REF_SRC = "/root/reference/exon/exon-core/src"
SHIM_FILES = ["lib.rs", "rule.rs", "udtf.rs"]


def _shim(name):
    return open(os.path.join(ROOT, "shim", "src", name)).read()


def _strip_comments(src):
    return re.sub(r"//[^\n]*", "", src)


def exon_use_paths():
    """[(file, ['datasources', 'vcf', 'VCFScan']), ...] for every `use exon::...;` (brace groups expanded)"""
    out = []
    for f in SHIM_FILES:
        for m in re.finditer(r"\buse exon::([^;]+);", _strip_comments(_shim(f))):
            path = "".join(m.group(1).split())
            g = re.fullmatch(r"(.*)::\{(.*)\}", path)
            leaves = [f"{g.group(1)}::{x}" for x in g.group(2).split(",") if x] if g else [path]
            out += [(f, leaf.split("::")) for leaf in leaves]
    return out


def _module_file(dirpath, name):
    for cand in (os.path.join(dirpath, name + ".rs"), os.path.join(dirpath, name, "mod.rs")):
        if os.path.exists(cand):
            return cand
    return None


def resolve_public(path):
    """None when `exon::<path>` is importable from another crate, else the reason it is not."""
    cur_file, cur_dir = os.path.join(REF_SRC, "lib.rs"), REF_SRC
    for i, seg in enumerate(path):
        src = _strip_comments(open(cur_file).read())
        last = i == len(path) - 1
        if re.search(r"(?m)^\s*pub mod " + seg + r"\s*;", src):
            nxt = _module_file(cur_dir, seg)
            if nxt is None:
                return f"module file of {seg} not found"
            cur_file, cur_dir = nxt, (os.path.join(cur_dir, seg) if nxt.endswith("mod.rs") else cur_dir)
            if last:
                return None
            continue
        if re.search(r"(?m)^\s*(pub\(\w+\)\s+)?mod " + seg + r"\s*;", src):
            return f"`{seg}` is not a `pub mod` in {os.path.relpath(cur_file, REF_SRC)}"
        if not last:
            return f"`{seg}` is not a module of {os.path.relpath(cur_file, REF_SRC)}"
        if re.search(r"(?m)^\s*pub use [^;]*\b" + seg + r"\b[^;]*;", src) or \
                re.search(r"(?m)^\s*pub (struct|enum|trait|fn|type|const|static) " + seg + r"\b", src):
            return None
        return f"`{seg}` is not a `pub` item or re-export of {os.path.relpath(cur_file, REF_SRC)}"
    return None


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree is only present in the build container")
def test_every_exon_import_of_the_shim_is_public_in_the_reference():
    paths = exon_use_paths()
    assert len(paths) >= 4
    for f, path in paths:
        why = resolve_public(path)
        assert why is None, f"shim/src/{f}: use exon::{'::'.join(path)} -- {why}"
    # the resolver itself: CRAMScan sits in a pub(crate) module (exon-core/src/datasources/cram/mod.rs:15-20)
    assert resolve_public(["datasources", "cram", "scanner", "CRAMScan"]) is not None
    assert resolve_public(["datasources", "cram", "CRAMScan"]) is not None
    assert resolve_public(["datasources", "vcf", "VCFScan"]) is None
    assert resolve_public(["ExonSession"]) is None


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference tree is only present in the build container")
def test_accessors_and_scan_names_the_rule_relies_on_exist_in_the_reference():
    ds = os.path.join(REF_SRC, "datasources")
    rule = _shim("rule.rs")
    # public accessors used through a downcast
    assert "pub fn base_config(&self)" in open(os.path.join(ds, "vcf", "scanner.rs")).read()
    assert "pub fn base_config(&self)" in open(os.path.join(ds, "vcf", "indexed_scanner.rs")).read()
    # the region has NO accessor upstream (hence region_from_debug / the feature) but both scanners derive Debug
    for f in ("vcf/indexed_scanner.rs", "bam/indexed_scanner.rs"):
        src = open(os.path.join(ds, f)).read()
        assert "pub fn region(" not in src and re.search(r"#\[derive\(Debug, Clone\)\]", src) and "region: Arc<Region>" in src
    # scans matched by ExecutionPlan::name(): the strings must be what the reference returns
    names = re.search(r"const ALIGNMENT_SCANS: \[&str; \d+\] = \[([^\]]*)\]", rule).group(1)
    returned = set()
    for root, _, files in os.walk(ds):
        for fn in files:
            if fn.endswith(".rs"):
                returned |= set(re.findall(r'fn name\(&self\) -> &str \{\s*"(\w+)"', open(os.path.join(root, fn)).read()))
    for n in re.findall(r'"(\w+)"', names):
        assert n in returned, f"no ExecutionPlan in the reference is named {n}"
    # mapping_quality really is Utf8 upstream, which is why the Exec converts it
    assert re.search(r'Field::new\("mapping_quality", DataType::Utf8', open("/root/reference/exon/exon-sam/src/schema_builder.rs").read())
    # ExonSession::new returns Self (no `?` on it anywhere in the shim or INTEGRATION.md)
    ext = open(os.path.join(REF_SRC, "session_context", "exon_context_ext.rs")).read()
    assert re.search(r"pub fn new\(session: SessionContext\) -> Self", ext)
    for text in [_shim(f) for f in SHIM_FILES] + [open(os.path.join(ROOT, "INTEGRATION.md")).read()]:
        assert not re.search(r"ExonSession::new\([^;\n]*\)\s*\?", text), "ExonSession::new returns Self, not a Result"


def test_plan_shapes_have_matchers_and_the_known_defects_stay_fixed():
    rule, lib = _strip_comments(_shim("rule.rs")), _strip_comments(_shim("lib.rs"))
    doc = _shim("rule.rs")
    # every matcher the shape table names is defined AND called from rewrite()
    table = re.findall(r"\| `(match_\w+)` \|", doc)
    assert set(table) == {"match_vcf_filter", "match_indexed_vcf", "match_alignment_filter", "match_indexed_bam"}
    body = rule[rule.index("fn rewrite("):]
    for fn in set(table):
        assert re.search(r"\bfn " + fn + r"\(", rule) and re.search(r"\b" + fn + r"\(", body), fn
    # pushed-down region scans are matched WITHOUT a FilterExec above them
    assert "downcast_ref::<IndexedVCFScanner>()" in body and "downcast_ref::<IndexedBAMScan>()" in body
    # get_field is matched structurally, never through Display text
    assert 'downcast_ref::<ScalarFunctionExpr>()' in rule and 'f.name() != "get_field"' in rule
    assert ".to_string()" not in rule[rule.index("fn info_field_of"):rule.index("fn aggregates_are")]
    # no import of the pub(crate) CRAM scanner
    assert "cram::CRAMScan" not in rule and "cram::CRAMScan" not in lib
    # the stream handle crosses `.await` inside a Send newtype; no bare stream pointer lives in execute()
    assert "unsafe impl Send for StreamHandle {}" in lib
    exe = lib[lib.index("fn execute("):]
    assert "*mut sys::exon_hip_stream" not in exe and "input.next().await" in exe
    # Utf8 mapping_quality is converted before the push
    assert "fn mapq_to_u8(" in lib and "mapq_to_u8(&batch, mapq_col)" in exe
    #  the plans DataFusion really emits: a round-robin RepartitionExec / CoalesceBatchesExec / column-only
    # ProjectionExec between the partial aggregate, the filter and the scan is peeled off -- above AND below the filter --
    # and a FilterExec with an embedded projection is no longer a reason to give up
    assert "fn peel(" in rule and "downcast_ref::<RepartitionExec>()" in rule and "Partitioning::RoundRobinBatch(_)" in rule
    assert "peel(agg.input(), true)" in body and "peel(filter.input(), false)" in body
    assert "filter.projection().is_some()" not in body
    peel = rule[rule.index("fn peel("):rule.index("impl GpuFilterAggRule", rule.index("fn peel("))]
    assert "CoalesceBatchesExec" in peel and "ProjectionExec" in peel and "_ => return cur" in peel   # a Hash repartition is NOT skipped
    #  whole-file decode + kernels run inside spawn_blocking, never on a tokio worker; one stream per partition,
    # its keys by value, the region's contig by name
    files = exe[exe.index("Source::Files {"):exe.index("Source::ChildBatches { seed_key } =>")]
    assert "tokio::task::spawn_blocking(move ||" in files and files.index("spawn_blocking") < files.index("exon_hip_stream_consume_scan")
    assert "stream.keys(&plan)" in files and "exon_hip_stream_set_region_contig" in files and "GpuPlan::try_new" not in files
    assert re.search(r'^tokio = ', open(os.path.join(ROOT, "shim", "Cargo.toml")).read().split("[dev-dependencies]")[0], re.M)
    #  an empty / non-positive point interval is not turned into a region string
    pr = rule[rule.index("fn point_region("):rule.index("fn region_text(")]
    assert "a > b || b < 1" in pr
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "RepartitionExec: partitioning=RoundRobinBatch(n), input_partitions=1" in integ and "GpuFilterAggExec: kind=4" in integ
    # the table function of config 5 is registered under its SQL name
    udtf = _shim("udtf.rs")
    assert 'register_udtf("fastq_quality_histogram"' in udtf and "impl TableFunctionImpl for FastqQualityHistogram" in udtf


def test_rule_and_exec_are_written():
    """the planner-side half must exist, not just be named."""
    lib = open(os.path.join(ROOT, "shim", "src", "lib.rs")).read()
    rule = open(os.path.join(ROOT, "shim", "src", "rule.rs")).read()
    assert "impl PhysicalOptimizerRule for GpuFilterAggRule" in rule and "pub fn try_new(" in lib
    assert "impl ExecutionPlan for GpuFilterAggExec" in lib and "AggregateMode::Partial" in rule
    for sym in re.findall(r"sys::(exon_hip_\w+)\(", lib + rule + _shim("udtf.rs")):
        assert f"pub fn {sym}(" in SYS_RS, f"{sym} used by the shim but not bound in sys.rs"
