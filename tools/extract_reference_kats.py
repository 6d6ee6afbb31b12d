#!/usr/bin/env python3
"""Collect the known-answer vectors the reference's own tests hold for the verify path into
tests/golden/reference_kats.json (DATA only: inputs + expected outputs, as decimal/hex strings).

Runs only in the build container (reads /root/reference/*_test.go); the JSON travels.
Sources (file:line under /root/reference):
  pairing_test.go:9-58      e(G1gen, G2gen) -- 12 Fq coefficients ("against Relic")
  hash_test.go:12-26        HashG1("the message to be signed")
  hash_test.go:48-62        HashG2 vector (declared; not enforced upstream because g2.go:141-143)
  hash_test.go:70-82        HashG2WithDomain(0^32, 0^8) compressed
  fq2_test.go:71-246        Fq2 square/mul/inverse/add/sub/neg/double/frobenius/sqrt vectors
  g1_test.go:62-104         G1 doubling / addition vectors
  fqrepr_test.go:136-147    MontReduce vector
  primitivefuncs_test.go:25-262  carry tables + 768-bit product vector
  g2pubs/bls_test.go:323-347, g1pubs/bls_test.go:411-433  DeriveSecretKey, invalid-pubkey hex
  g1.go:25-26, g2.go:26-29  generators
"""
import json, re, pathlib
REF = pathlib.Path("/root/reference")
def rd(p): return (REF / p).read_text()

def func_body(src, name):
    m = re.search(r"func %s\(.*?\n}\n" % name, src, re.S)
    return m.group(0)

def strs(body):
    """all FQReprFromString("..", base) literals in order -> ints"""
    return [int(v, int(b)) for v, b in re.findall(r'FQReprFromString\("([0-9a-fA-F]+)", (\d+)\)', body)]

def reprs(body):
    return [[int(x, 0) for x in re.findall(r"0x[0-9a-f]+|\b\d+\b", g)] for g in re.findall(r"FQRepr\{([^}]*)\}", body)]

def limbs_to_int(l): return sum(v << (64 * i) for i, v in enumerate(l))

K = {}
pt = rd("pairing_test.go")
K["pairing_g1gen_g2gen"] = [str(int(v)) for v in re.findall(r'var c\d\d\d, _ = bls.FQReprFromString\("(\d+)", 10\)', pt)]
assert len(K["pairing_g1gen_g2gen"]) == 12
ht = rd("hash_test.go")
g = dict(re.findall(r'var (expectedG\w+), _ = bls.FQReprFromString\("([0-9a-f]+)", 16\)', ht))
K["hash_g1"] = {"msg": "the message to be signed", "x": g["expectedG1X"], "y": g["expectedG1Y"]}
K["hash_g2"] = {"msg": "the message to be signed", "x_c0": g["expectedG2c0X"], "x_c1": g["expectedG2c1X"], "y_c0": g["expectedG2c0Y"], "y_c1": g["expectedG2c1Y"], "enforced_upstream": False}
K["hash_g2_with_domain"] = {"msg_hex": "00" * 32, "domain_hex": "00" * 8, "compressed_hex": re.search(r'expectedSerializedG2, _ = hex.DecodeString\("([0-9a-f]+)"\)', ht).group(1)}
f2 = rd("fq2_test.go")
def fq2case(name, nin):
    v = strs(func_body(f2, name)); return v
v = strs(func_body(f2, "TestFQ2Squaring")); K["fq2_square"] = {"a": [hex(v[0]), hex(v[1])], "out": [hex(v[2]), hex(v[3])]}
v = strs(func_body(f2, "TestFQ2Mul")); K["fq2_mul"] = {"a": [hex(v[0]), hex(v[1])], "b": [hex(v[2]), hex(v[3])], "out": [hex(v[4]), hex(v[5])]}
v = strs(func_body(f2, "TestFQ2Inverse")); K["fq2_inverse"] = {"a": [hex(v[0]), hex(v[1])], "out": [hex(v[2]), hex(v[3])]}
for nm, key in [("TestFQ2Addition", "fq2_add"), ("TestFQ2Subtraction", "fq2_sub")]:
    v = strs(func_body(f2, nm)); K[key] = {"a": [hex(v[0]), hex(v[1])], "b": [hex(v[2]), hex(v[3])], "out": [hex(v[4]), hex(v[5])]}
for nm, key in [("TestFQ2Negation", "fq2_neg"), ("TestFQ2Doubling", "fq2_double")]:
    v = strs(func_body(f2, nm)); K[key] = {"a": [hex(v[0]), hex(v[1])], "out": [hex(v[2]), hex(v[3])]}
v = strs(func_body(f2, "TestFQ2FrobeniusMap"))
K["fq2_frobenius"] = {"a": [hex(v[0]), hex(v[1])], "steps": [{"power": p, "out": [hex(v[2 + 2 * i]), hex(v[3 + 2 * i])]} for i, p in enumerate([0, 1, 1, 2])]}
v = strs(func_body(f2, "TestFQ2Sqrt"))
K["fq2_sqrt"] = [{"a": [hex(v[0]), hex(v[1])], "out": [hex(v[2]), hex(v[3])]}, {"a": [hex(v[4]), "0x0"], "out": ["0x0", hex(v[5])]}]
g1t = rd("g1_test.go")
r = reprs(func_body(g1t, "TestG1DoublingCorrectness")); K["g1_double"] = {"p": [hex(limbs_to_int(r[0])), hex(limbs_to_int(r[1]))], "out": [hex(limbs_to_int(r[2])), hex(limbs_to_int(r[3]))]}
r = reprs(func_body(g1t, "TestG1AdditionCorrectness")); K["g1_add"] = {"p1": [hex(limbs_to_int(r[0])), hex(limbs_to_int(r[1]))], "p2": [hex(limbs_to_int(r[2])), hex(limbs_to_int(r[3]))], "out": [hex(limbs_to_int(r[4])), hex(limbs_to_int(r[5]))]}
r = reprs(func_body(rd("fqrepr_test.go"), "TestMontReduce")); K["mont_reduce"] = {"hi": [str(x) for x in r[0]], "lo": [str(x) for x in r[1]], "out": [str(x) for x in r[2]]}
pf = rd("primitivefuncs_test.go")
b = func_body(pf, "TestMultiplyFQReprOverflow"); r = reprs(b)
K["multiply_fqrepr"] = {"f0": [str(x) for x in r[0]], "f1": [str(x) for x in r[1]], "lo": re.search(r'expectedLo, _ := new\(big.Int\).SetString\("(\d+)"', b).group(1), "hi": re.search(r'expectedHi, _ := new\(big.Int\).SetString\("(\d+)"', b).group(1)}
def table(name, fields):
    body = func_body(pf, name); out = []
    for blk in re.findall(r"\{\s*((?:\w+:\s*\d+,\s*)+)\}", body):
        d = dict(re.findall(r"(\w+):\s*(\d+)", blk))
        if all(f in d for f in fields): out.append({f: d[f] for f in fields})
    return out
K["sub_with_borrow"] = table("TestSubWithCarry", ["a", "b", "borrow", "out", "outBorrow"])
def ptable(name, fields):
    """positional struct literals: { v, v, v, ... }"""
    body = func_body(pf, name); body = body[body.index("}{"):]
    out = []
    for blk in re.findall(r"\{\s*((?:\d+,\s*)+)\}", body):
        vals = re.findall(r"\d+", blk)
        if len(vals) == len(fields): out.append(dict(zip(fields, vals)))
    return out
K["add_with_carry"] = ptable("TestAddWithCarry", ["a", "b", "carry", "out", "outCarry"])
K["mac_with_carry"] = ptable("TestMACWithCarry", ["a", "b", "c", "carry", "out", "outCarry"])
g2t = rd("g2pubs/bls_test.go"); g1p = rd("g1pubs/bls_test.go")
K["derive_secret_key"] = {"in_ascii": "11223344556677889900112233445566", "fr_hex": re.search(r'FRReprFromString\("([0-9a-f]+)", 16\)', func_body(g2t, "TestDeriveSecretKey")).group(1)}
K["invalid_pubkey_g2pubs_hex"] = re.search(r'unexpectedPub := "([0-9a-f]+)"', func_body(g2t, "TestPubkeyDeserializeInvalid")).group(1)
K["invalid_pubkey_g1pubs_hex"] = re.search(r'unexpectedPub := "([0-9a-f]+)"', func_body(g1p, "TestPubkeyDeserializeInvalid")).group(1)
g1 = rd("g1.go"); g2 = rd("g2.go")
K["g1_generator"] = {"x": re.search(r'g1GeneratorX, _ = FQReprFromString\("(\d+)", 10\)', g1).group(1), "y": re.search(r'g1GeneratorY, _ = FQReprFromString\("(\d+)", 10\)', g1).group(1)}
K["g2_generator"] = {k: re.search(r'g2Generator%s, _ = FQReprFromString\("([0-9a-f]+)", 16\)' % k, g2).group(1) for k in ["XC0", "XC1", "YC0", "YC1"]}
# Montgomery-image samples of hard-coded tables (fq6.go:144-208, fq12.go:122-168, fq2.go:149-152, g1.go:29) to pin derived constants
K["mont_images"] = {
  "frob_fq2_c1_1": reprs(re.search(r"frobeniusCoeffFQ2c1 = .*?\n}\n", rd("fq2.go"), re.S).group(0))[0],
  "frob_fq6_c1": reprs(re.search(r"frobeniusCoeffFQ6c1 = .*?\n}\n", rd("fq6.go"), re.S).group(0)),
  "frob_fq6_c2": reprs(re.search(r"frobeniusCoeffFQ6c2 = .*?\n}\n", rd("fq6.go"), re.S).group(0)),
  "frob_fq12_c1": reprs(re.search(r"frobeniusCoeffFQ12c1 = .*?\n}\n", rd("fq12.go"), re.S).group(0)),
  "b_coeff": reprs(re.search(r"var BCoeff = .*", g1).group(0))[0],
}
K["mont_images"] = {k: ([[str(x) for x in row] for row in v] if isinstance(v[0], list) else [str(x) for x in v]) for k, v in K["mont_images"].items()}
out = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden" / "reference_kats.json"
out.write_text(json.dumps(K, indent=1) + "\n")
print("wrote", out, {k: (len(v) if hasattr(v, "__len__") else 1) for k, v in K.items()})
