"""The Go shims (shim/g2pubs/accel_cgo.go, shim/g1pubs/accel_cgo.go) are source a maintainer drops into the reference's packages; no Go
toolchain exists in this image, so what CAN be checked is checked: every C.blsmi_* call and C.BLSMI_* constant they use exists in
include/blsmi.h with the same number of arguments AND the same C type per argument, in order (the Go argument expressions are typed by a
small inferrer: u8()/u64() helpers, (*C.T)(...) conversions, C.size_t(...), &x of a declared C variable, &slice[0]), the exported Go
functions are the reference's verify surface (names and parameter lists from g2pubs/bls.go:159, 240, 275 and g1pubs/bls.go:165-174,
252-311), nothing newer than the `go` directive of the reference's go.mod is used (unsafe.Slice under go 1.13), a device
error never becomes a verdict, INTEGRATION.md shows the files verbatim, and -- when a `go` binary is present -- gofmt accepts them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = [os.path.join(ROOT, "shim", "g2pubs", "accel_cgo.go"), os.path.join(ROOT, "shim", "g1pubs", "accel_cgo.go")]


def _strip_c_comments(txt):
    return re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)


def _split_args(s):
    """top-level comma split of an argument list"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def header_prototypes():
    txt = _strip_c_comments(open(os.path.join(ROOT, "include", "blsmi.h")).read())
    protos = {}
    for m in re.finditer(r"\b(blsmi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    consts = set(re.findall(r"\b(BLSMI_[A-Z0-9_]+)\b", txt))
    return protos, consts


def go_calls(src):
    """(name, number of arguments) of every C.blsmi_*( ... ) call, with balanced-parenthesis matching"""
    calls = []
    for m in re.finditer(r"C\.(blsmi_[a-z0-9_]+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        calls.append((m.group(1), len(_split_args(src[m.end():i - 1]))))
    return calls


def _go_sources():
    srcs = {p: open(p).read() for p in SHIMS}
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for k, block in enumerate(re.findall(r"```go\n(.*?)```", md, flags=re.S)):
        srcs["INTEGRATION.md#go-block-%d" % k] = block
    return srcs


def test_every_c_call_matches_the_header():
    protos, consts = header_prototypes()
    assert len(protos) >= 70
    seen = set()
    for name, src in _go_sources().items():
        for fn, nargs in go_calls(src):
            assert fn in protos, "%s calls C.%s, which include/blsmi.h does not declare" % (name, fn)
            assert nargs == protos[fn], "%s: C.%s called with %d arguments, the header declares %d" % (name, fn, nargs, protos[fn])
            seen.add(fn)
        for c in re.findall(r"C\.(BLSMI_[A-Z0-9_]+)", src):
            assert c in consts, "%s uses C.%s, which include/blsmi.h does not define" % (name, c)
    # the verify surface of both packages is bound
    # ... through the in-memory-point forms (blsmi 0.6): the shims copy the G?Projective structs, nothing else
    for fn in ("blsmi_g2pubs_verify_batch_jac", "blsmi_g2pubs_verify_aggregate_jac", "blsmi_g2pubs_verify_aggregate_common_jac", "blsmi_g1pubs_verify_batch_jac",
               "blsmi_g1pubs_verify_aggregate_jac", "blsmi_g1pubs_verify_aggregate_common_jac", "blsmi_g1pubs_verify_with_domain_batch_jac",
               "blsmi_g1pubs_verify_aggregate_with_domain_jac", "blsmi_g1pubs_verify_aggregate_common_with_domain_jac", "blsmi_g2_prepared_create_jac",
               "blsmi_g2pubs_verify_batch_prepared_jac", "blsmi_g1_sum_jac", "blsmi_g2_sum_jac", "blsmi_init_devices", "blsmi_prefer_cpu"):
        assert fn in seen, fn


def _go_code(src):
    """Go source without its comments (// to end of line, /* */ blocks)"""
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return "\n".join(line.split("//")[0] for line in src.splitlines())


def test_no_per_point_host_field_arithmetic_in_the_shims():
    """a verify path that runs ToAffine() (an Fq inversion, g1.go:322-340) and SerializeBytes() (MontReduce + byte swap,
    g1.go:157-167) per point on a host core caps a Go caller at ~40 k tuples/s per core.  The shims hand over the G?Projective structs as they
    lie: no ToAffine, no SerializeBytes, no infinity flags anywhere in their code, and every pack* helper is a fixed-size copy."""
    for path in SHIMS:
        code = _go_code(open(path).read())
        assert "ToAffine" not in code and "SerializeBytes" not in code and "IsZero" not in code, path
        for fn in ("packKeys", "packSigs"):
            body = re.search(r"func %s\(.*?\n}\n" % fn, code, flags=re.S).group(0)
            assert re.search(r"copy\(\w+\[(18|36)\*i:(18|36)\*i\+(18|36)\], \(\*\[(18|36)\]C\.uint64_t\)\(unsafe\.Pointer\(\w+\[i\]\.[ps]\)\)\[:\]\)", body), (path, fn)
            assert "C.blsmi" not in body and "bls." not in body
        pkg = os.path.basename(os.path.dirname(path))
        # record sizes: g2pubs keys are G2Projective (36 words) and signatures G1Projective (18); g1pubs the other way round
        kw, sw = (36, 18) if pkg == "g2pubs" else (18, 36)
        assert "pk := make([]C.uint64_t, %d*len(pubs))" % kw in code and "sg := make([]C.uint64_t, %d*len(sigs))" % sw in code


def test_shims_export_the_references_verify_surface():
    want = {"g2pubs": ["func Verify(m []byte, pub *PublicKey, sig *Signature) bool",
                       "func (s *Signature) VerifyAggregate(pubKeys []*PublicKey, msgs [][]byte) bool",
                       "func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool"],
            "g1pubs": ["func Verify(m []byte, pub *PublicKey, sig *Signature) bool",
                       "func VerifyWithDomain(m [32]byte, pub *PublicKey, sig *Signature, domain [8]byte) bool",
                       "func (s *Signature) VerifyAggregate(pubKeys []*PublicKey, msgs [][]byte) bool",
                       "func (s *Signature) VerifyAggregateCommon(pubKeys []*PublicKey, msg []byte) bool",
                       "func (s *Signature) VerifyAggregateCommonWithDomain(pubKeys []*PublicKey, msg [32]byte, domain [8]byte) bool",
                       "func (s *Signature) VerifyAggregateWithDomain(pubKeys []*PublicKey, msgs [][32]byte, domain [8]byte) bool"]}
    for path in SHIMS:
        src = open(path).read()
        pkg = re.search(r"^package (\w+)$", src, flags=re.M).group(1)
        assert os.path.basename(os.path.dirname(path)) == pkg
        assert src.startswith("// +build cgo,blsmi\n") and '#include "blsmi.h"' in src and 'import "C"' in src
        for sig in want[pkg]:
            assert sig + " {" in src, (pkg, sig)


def test_shims_are_formatted_like_gofmt_output():
    """cheap structural stand-in for gofmt when there is no go binary: tabs for indentation, no trailing blanks, one statement per line
    inside braces (no `if x { y }` one-liners, which gofmt rewrites), balanced braces"""
    for path in SHIMS:
        src = open(path).read()
        assert src.count("{") == src.count("}") and src.count("(") == src.count(")")
        in_c = False
        for ln, line in enumerate(src.splitlines(), 1):
            if line.startswith("/*"):
                in_c = True
            if not in_c:
                assert line == line.rstrip(), "%s:%d trailing blank" % (path, ln)
                assert not re.match(r"^ +\S", line), "%s:%d indented with spaces" % (path, ln)
                code = line.split("//")[0]
                assert not re.search(r"\bif\b[^{]*\{[^}]+\}\s*$", code), "%s:%d one-line if block" % (path, ln)
            if line.startswith("*/"):
                in_c = False
    if shutil.which("gofmt"):
        r = subprocess.run(["gofmt", "-l"] + SHIMS, capture_output=True, text=True)
        assert r.returncode == 0 and not r.stdout.strip(), r.stdout + r.stderr
    if shutil.which("go"):
        r = subprocess.run(["go", "vet", "-tags", "blsmi"] + SHIMS, capture_output=True, text=True)
        # outside the upstream module the package's other files are missing; only syntax errors are fatal here
        assert "syntax error" not in r.stderr, r.stderr


def test_integration_md_includes_the_shims_verbatim():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```go\n(.*?)```", md, flags=re.S)
    for path in SHIMS:
        assert open(path).read() in blocks, "INTEGRATION.md does not show %s verbatim (run tools/sync_integration.py)" % os.path.relpath(path, ROOT)


# ---- the boundary has to survive a compiler ---------------------------------------------------------------------
def _c_param_type(decl):
    """'const uint8_t *msgs' -> '*uint8_t', 'size_t n' -> 'size_t', 'void **out' -> '**void' (const dropped, the name dropped)"""
    d = re.sub(r"\bconst\b", " ", decl).strip()
    arr = re.search(r"\[\d*\]\s*$", d)                                   # `const uint64_t sig[18]` is a pointer parameter
    if arr:
        d = d[:arr.start()]
    stars = d.count("*") + (1 if arr else 0)
    d = d.replace("*", " ")
    words = d.split()
    assert words, decl
    base = " ".join(words[:-1]) if len(words) > 1 else words[0]          # the last word is the parameter's name
    return "*" * stars + base


def header_param_types():
    txt = _strip_c_comments(open(os.path.join(ROOT, "include", "blsmi.h")).read())
    out = {}
    for m in re.finditer(r"\b(blsmi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = [] if args in ("", "void") else [_c_param_type(a) for a in _split_args(args)]
    return out


_INT_TYPES = {"int", "size_t", "long long", "unsigned", "uint32_t", "uint64_t", "int64_t"}


def _go_arg_type(expr, body, src):
    """C type of a Go argument expression of the shims ('*uint8_t', 'size_t', 'nil' = any pointer, 'intconst' = any integer)"""
    e = expr.strip()
    if e == "nil":
        return "nil"
    if re.fullmatch(r"\d+", e) or re.fullmatch(r"C\.BLSMI_[A-Z0-9_]+", e):
        return "intconst"
    m = re.match(r"\(\*C\.(\w+)\)\(", e)
    if m:
        return "*" + m.group(1)
    m = re.match(r"C\.(\w+)\(", e)
    if m:
        return m.group(1)
    if e.startswith("u8("):
        return "*uint8_t"
    if e.startswith("u64(") or e.startswith("sigWords("):
        return "*uint64_t"
    m = re.fullmatch(r"&(\w+)\[0\]", e)
    if m:                                                                # first element of a slice declared in this function
        name = m.group(1)
        d = re.search(r"\b%s := make\(\[\]C\.(\w+)," % name, body)
        if d:
            return "*" + d.group(1)
        if re.search(r"\b\w+, %s := packMsgs\(" % name, body):             # packMsgs returns (m []byte, off []C.uint64_t)
            assert re.search(r"func packMsgs\(msgs \[\]\[\]byte\) \(m \[\]byte, off \[\]C\.uint64_t\)", src)
            return "*uint64_t"
        raise AssertionError("cannot type %r" % e)
    m = re.fullmatch(r"&(\w+)", e)
    if m:
        d = re.search(r"\bvar %s (C\.(\w+)|unsafe\.Pointer)\b" % m.group(1), body)
        assert d, "cannot type %r" % e
        return "**void" if d.group(1) == "unsafe.Pointer" else "*" + d.group(2)
    m = re.fullmatch(r"(\w+)\.(\w+)", e)
    if m:                                                                # a struct field: `h unsafe.Pointer`
        d = re.search(r"^\t%s +(unsafe\.Pointer|\*?C\.\w+)$" % m.group(2), src, flags=re.M)
        assert d, "cannot type %r" % e
        return "*void" if d.group(1) == "unsafe.Pointer" else d.group(1).replace("C.", "")
    m = re.fullmatch(r"\w+", e)
    if m:                                                                # a local: `mp := u8(msg)`
        d = re.search(r"\b%s := (.+)" % e, body)
        assert d, "cannot type %r" % e
        return _go_arg_type(d.group(1), body, src)
    raise AssertionError("cannot type %r" % e)


def _go_functions(src):
    """(name, body) of every top-level func"""
    out = []
    for m in re.finditer(r"^func [^\n]*\{\n(.*?)^}\n", src, flags=re.S | re.M):
        out.append(m.group(0))
    return out


def _check_call_types(path, src, protos):
    """raises AssertionError on the first C.blsmi_* call of `src` whose arguments do not have the header's types in the header's order"""
    code = _go_code(src)
    checked = 0
    for fn_src in _go_functions(code):
        for m in re.finditer(r"C\.(blsmi_[a-z0-9_]+)\(", fn_src):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(fn_src[i], 0)
                i += 1
            args = _split_args(fn_src[m.end():i - 1])
            want = protos[m.group(1)]
            assert len(args) == len(want), (path, m.group(1))
            for k, (a, w) in enumerate(zip(args, want)):
                got = _go_arg_type(a, fn_src, code)
                if got == "nil":
                    assert w.startswith("*"), "%s: C.%s argument %d: nil for a %s" % (path, m.group(1), k, w)
                elif got == "intconst":
                    assert w in _INT_TYPES, "%s: C.%s argument %d: integer constant for a %s" % (path, m.group(1), k, w)
                else:
                    assert got == w, "%s: C.%s argument %d is %s (%s), the header declares %s" % (path, m.group(1), k, got, a.strip(), w)
                checked += 1
    # the record widths follow the group: a g2pubs key is a G2Projective (36 words), its signature a G1Projective (18); g1pubs the other way
    # round -- a swapped (pk, sg) pair has the right C types, so the ORDER is checked by name as well
    for m in re.finditer(r"C\.blsmi_g[12]pubs_verify\w*_jac\(([^;]*?)\)\)?, \"", code, flags=re.S):
        args = [a.strip() for a in _split_args(m.group(1))]
        if "u64(pk)" not in args:                                        # (the prepared-key form passes a handle and indices)
            continue
        assert args.index("u64(pk)") < (args.index("u64(sg)") if "u64(sg)" in args else args.index("sigWords(s)")), (path, args)
    return checked


def test_every_c_call_passes_the_headers_types_in_order():
    """Name and arity say nothing about a swapped (sigs, pubs) pair or a *C.uint8_t where the header wants uint64_t*: compare the C
    type of every Go argument expression with the header's parameter type, position by position.  The checker is itself checked on
    three deliberately broken copies of the shim."""
    protos = header_param_types()
    assert protos["blsmi_g2pubs_verify_aggregate_jac"] == ["*uint8_t", "*uint64_t", "*uint64_t", "*uint64_t", "size_t", "*int"]
    total = 0
    for path in SHIMS:
        total += _check_call_types(path, open(path).read(), protos)
    assert total >= 120
    good = open(SHIMS[0]).read()
    for a, b in (("u64(pk), u64(sg), u8(ok), nil, C.size_t(n)), \"g2pubs_verify_batch_jac\"", "u64(sg), u64(pk), u8(ok), nil, C.size_t(n)), \"g2pubs_verify_batch_jac\""),     # swapped records
                 ("u64(pk), sigWords(s), C.size_t(len(msgs)), &ok)", "u64(pk), sigWords(s), &ok, C.size_t(len(msgs)))"),                                                   # swapped tail
                 ("(*C.uint32_t)(unsafe.Pointer(&keyIdx[0]))", "(*C.uint64_t)(unsafe.Pointer(&keyIdx[0]))")):                                                             # wrong pointee
        assert a in good
        with pytest.raises(AssertionError):
            _check_call_types("mutant", good.replace(a, b), protos)


def test_nothing_newer_than_the_references_go_directive():
    """/root/reference/go.mod:15 says `go 1.13`: with -tags blsmi the shim is compiled in that language version.  unsafe.Slice / unsafe.Add /
    unsafe.String (1.17 / 1.20), generics and `any` (1.18), //go:build-only constraints (1.17), min / max / clear (1.21) must not appear."""
    gomod = "/root/reference/go.mod"
    if os.path.exists(gomod):                                            # (not on the GPU box; the directive is quoted in the shims' headers)
        assert re.search(r"^go 1\.13$", open(gomod).read(), flags=re.M)
    deny = [r"unsafe\.Slice", r"unsafe\.Add", r"unsafe\.String", r"unsafe\.SliceData", r"\bany\b", r"func \w+\[", r"\bclear\(", r"\bmin\(", r"\bmax\(",
            r"atomic\.(Int|Uint|Bool|Pointer)\d*\b", r"\bcgo\.Handle\b", r"0[bo][0-9]"]            # (0b / 0o literals are 1.13 itself: kept out for 1.12 readers too)
    srcs = _go_sources()
    for name, src in srcs.items():
        code = _go_code(src)
        for pat in deny:
            assert not re.search(pat, code), "%s uses %s, newer than the reference's `go 1.13`" % (name, pat)
    for path in SHIMS:
        src = open(path).read()
        assert src.startswith("// +build cgo,blsmi\n\n") and "//go:build" not in src     # the pre-1.17 constraint syntax, understood by every version
        assert "go 1.13" in src                                          # the header states the version the file is written for


def test_a_device_error_is_never_a_verdict():
    """`false` from Verify* means the pairing check failed (g2pubs/bls.go:240-270), nothing else: every C.blsmi_* call that returns a code
    goes through must() (panic), and no verdict is computed from a return code."""
    void_or_value = {"blsmi_prefer_cpu", "blsmi_g2_prepared_destroy", "blsmi_trim", "blsmi_host_free", "blsmi_shutdown", "blsmi_version", "blsmi_device_count",
                     "blsmi_held_bytes"}
    for path in SHIMS:
        code = _go_code(open(path).read())
        helper = re.search(r"func must\(rc C\.int, what string\) \{\n\tif rc != 0 \{\n\t\tpanic\([^\n]*\)\n\t\}\n\}\n", code)
        assert helper, path
        rest = code.replace(helper.group(0), "")
        assert not re.search(r"rc\s*==\s*0\s*&&", rest) and not re.search(r"\brc\b", rest), path   # no return code is looked at anywhere else
        for m in re.finditer(r"(must\()?C\.(blsmi_[a-z0-9_]+)\(", code):
            if m.group(2) in void_or_value:
                continue
            assert m.group(1), "%s: C.%s is called without must()" % (path, m.group(2))
        for fn in ("VerifyAggregate", "VerifyAggregateCommon", "VerifyAggregateWithDomain", "VerifyAggregateCommonWithDomain"):
            b = re.search(r"func \(s \*Signature\) %s\(.*?^}\n" % fn, code, flags=re.S | re.M)
            if b:
                assert "return ok != 0" in b.group(0), (path, fn)
        # the struct layouts the *_jac calls rely on are pinned at compile time
        for t, n in (("G2Projective", 288), ("G1Projective", 144), ("FQRepr", 48)):
            assert "var _ = [1]struct{}{}[unsafe.Sizeof(bls.%s{})-%d]" % (t, n) in code, (path, t)
