"""The Rust shim (rust/src/lib.rs) cannot be compiled in this pipeline (no rustc), so nothing would notice its `extern "C"`
block drifting away from the C headers it binds.  This test parses both sides and compares, item by item: name, arity, and
the scalar / pointer type of the return value and of every argument.  It also checks that the shim keeps the serde surface
of the reference (src/layer.rs:41-68, src/index.rs:68, src/geom.rs:83) behind its `serde` feature."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "void": "c_void",
             "char": "c_char", "uint8_t": "u8", "bp_layer_t": "BpLayer", "bp_group_t": "BpGroup", "bp_sharded_t": "BpSharded",
             "bp_filter_fn": "BpFilterFn"}


def _c_type_to_rust(decl, is_param):
    """`const uint32_t **pairs` -> `*mut *const u32`; the parameter name (if any) is dropped"""
    decl = decl.strip()
    stars = decl.count("*")
    toks = [t for t in re.split(r"[\s*]+", decl) if t]
    const = "const" in toks
    toks = [t for t in toks if t != "const"]
    base = toks[0]
    if base not in C_SCALARS:          # structs the shim does not bind (bp_facts, bp_stats, ...): kept by their C name
        C_SCALARS[base] = base
    if is_param:
        assert len(toks) == 2, "parameter without a name (or of a compound type): %r" % decl
    rust = C_SCALARS[base]
    for level in range(stars):
        inner_const = const if level == 0 else False      # only `const T *` / `const T **` forms occur: the const binds to T
        rust = ("*const " if inner_const else "*mut ") + rust
    return rust


def _c_prototypes(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for stmt in text.split(";"):
        m = re.match(r"^\s*(?:extern\s+\"C\"\s*\{)?\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*\b(bp_[a-z0-9_]+)\s*\(([^()]*)\)\s*$", stmt, flags=re.S)
        if not m:
            continue
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        protos[name] = (_c_type_to_rust(ret + " r", True) if "*" in ret else C_SCALARS[ret.strip()],
                        [_c_type_to_rust(a, True) for a in args])
    return protos


def _rust_externs():
    src = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    items = {}
    for m in re.finditer(r"fn\s+(bp_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        types = []
        for a in (x.strip() for x in args.split(",")):
            if a:
                types.append(re.sub(r"\s+", " ", a.split(":", 1)[1].strip()))
        items[name] = (re.sub(r"\s+", " ", ret), types)
    return items


def test_every_rust_extern_matches_its_c_prototype():
    protos = {}
    protos.update(_c_prototypes("bp_layer.h"))
    protos.update(_c_prototypes("bp_sharded.h"))
    assert len(protos) >= 60, len(protos)
    externs = _rust_externs()
    assert len(externs) >= 30
    for name, (ret, args) in externs.items():
        assert name in protos, "rust/src/lib.rs binds %s, which no header declares" % name
        c_ret, c_args = protos[name]
        assert len(args) == len(c_args), "%s: %d arguments in Rust, %d in C" % (name, len(args), len(c_args))
        assert ret == c_ret, "%s: returns %s in Rust, %s in C" % (name, ret, c_ret)
        for k, (r, c) in enumerate(zip(args, c_args)):
            assert r == c, "%s: argument %d is %s in Rust, %s in C" % (name, k, r, c)
    # the calls the reference's hot-path surface needs are all bound
    for need in ("bp_layer_create", "bp_layer_clear", "bp_layer_extend_host", "bp_layer_merge", "bp_layer_sort", "bp_layer_par_sort",
                 "bp_layer_scan", "bp_layer_par_scan", "bp_layer_scan_filtered", "bp_layer_par_scan_filtered", "bp_layer_tree",
                 "bp_layer_set_tree", "bp_layer_min_depth", "bp_sharded_par_scan"):
        assert need in externs, need


def test_the_parser_notices_a_drift():
    protos = _c_prototypes("bp_layer.h")
    assert protos["bp_layer_scan"] == ("c_int", ["*mut BpLayer", "*mut *const u32", "*mut usize"])
    assert protos["bp_layer_set_tree"] == ("c_int", ["*mut BpLayer", "u32", "*const u64", "*const u32", "usize", "c_int"])
    assert protos["bp_layer_last_error"][0] == "*const c_char"


def test_rust_shim_keeps_the_reference_serde_surface():
    src = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    cargo = open(os.path.join(ROOT, "rust", "Cargo.toml")).read()
    assert re.search(r'serde\s*=\s*\{[^}]*optional\s*=\s*true', cargo)                # feature "serde", as the reference's Cargo.toml:27
    assert 'cfg_attr(feature = "serde", derive(Deserialize, Serialize))' in src        # Index* newtypes and Bounds
    assert "impl<Index> Serialize for Layer<Index, u32>" in src and "Deserialize<'de> for Layer<Index, u32>" in src
    # exactly the persistent state, in the reference's order (bincode is positional): min_depth, then tree = (Vec<(Index, ID)>, bool)
    ser = src[src.index("impl<Index> Serialize for Layer"):src.index("impl<'de, Index> Deserialize<'de> for Layer")]
    assert ser.index('serialize_field("min_depth"') < ser.index('serialize_field("tree"') and 'serialize_struct("Layer", 2)' in ser
