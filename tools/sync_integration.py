#!/usr/bin/env python3
"""Rewrite the two shim code blocks of INTEGRATION.md (sections 2 and 2b) from shim/g2pubs/accel_cgo.go and shim/g1pubs/accel_cgo.go,
so that the document shows the committed files verbatim (tests/test_shim.py::test_integration_md_includes_the_shims_verbatim)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md_path = os.path.join(ROOT, "INTEGRATION.md")
md = open(md_path).read()
for pkg in ("g2pubs", "g1pubs"):
    src = open(os.path.join(ROOT, "shim", pkg, "accel_cgo.go")).read()
    pat = re.compile(r"```go\n// \+build cgo,blsmi\n(?:(?!```).)*?\npackage %s\n.*?```" % pkg, flags=re.S)
    assert pat.search(md), pkg
    md = pat.sub(lambda m: "```go\n" + src + "```", md, count=1)
open(md_path, "w").write(md)
print("INTEGRATION.md synchronised")
