"""The Go shims (shim/g2pubs/accel_cgo.go, shim/g1pubs/accel_cgo.go) are source a maintainer drops into the reference's packages; no Go
toolchain exists in this image, so what CAN be checked is checked: every C.blsmi_* call and C.BLSMI_* constant they use exists in
include/blsmi.h with the same number of arguments, the exported Go functions are the reference's verify surface (names and parameter
lists from g2pubs/bls.go:159, 240, 275 and g1pubs/bls.go:165-174, 252-311), INTEGRATION.md shows the files verbatim, and -- when a
`go` binary is present -- gofmt accepts them."""
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
