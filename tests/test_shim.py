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
    for fn in ("blsmi_g2pubs_verify_batch", "blsmi_g2pubs_verify_aggregate", "blsmi_g2pubs_verify_aggregate_common", "blsmi_g1pubs_verify_batch",
               "blsmi_g1pubs_verify_aggregate", "blsmi_g1pubs_verify_aggregate_common", "blsmi_g1pubs_verify_with_domain_batch",
               "blsmi_g1pubs_verify_aggregate_with_domain", "blsmi_g1pubs_verify_aggregate_common_with_domain", "blsmi_init_devices", "blsmi_prefer_cpu"):
        assert fn in seen, fn


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
