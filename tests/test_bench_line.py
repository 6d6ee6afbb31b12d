"""bench.py's stdout contract: ONE JSON line the driver can keep whole (< 4 KB), with the headline, its roofline and CPU baseline
and five compact config entries; everything else goes to bench_detail.json.  The formatter is exercised here with synthetic legs
(worst-case long strings and kernel lists); the -m gpu test runs the real command and parses its last stdout line."""
import json
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _roof(kernel, many=10):
    return {"bound": "hbm", "kernel": kernel, "achieved": 5.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.00064505, "traffic": 190567153984.6667,
            "algorithmic_bytes_per_launch": 234881024, "bytes_per_unit": 224, "units_per_launch": 1 << 20, "kernel_ms": 114.8603, "kernel_launches": 3,
            "all_kernels_ms": {"k_some_long_kernel_name_%d_pair" % i: 1.2345 * i for i in range(many)}, "rocprof_avg_ms": 113.9}


def _cpu(unit):
    return {"value": 34585.34, "unit": unit, "cores": 16, "kind": "port", "single_core_per_s": 3559.02, "sample": "x" * 600, "note": "y" * 400}


def _detail():
    """a detail object shaped like a full one-GPU run, padded well beyond what real runs produce"""
    d = {"metric": "BLS12-381 pairings/sec (batch verify)", "value": 3204526.7, "unit": "pairings/s", "n_gpus": 1, "steps": 10, "warmup": 2, "ms_per_step": 20.451,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 (14 x 28-bit limbs in the pairing / hash / curve kernels, 15 x 27 in the latency programs; int64 accumulate)", "data": "synthetic",
         "launch": "single process: 1 device(s) behind the C ABI (blsmi_init_devices)", "devices": 1, "rccl_ranks": 0, "library": "blsmi 0.5 gfx950:sramecc+:xnack- CUs=256 devices=1 shards=1",
         "config": {"workload": "w" * 500, "pairings_per_gpu": 65536, "parallelism": "shard1", "layout": "lane pair per tuple (one Fq2 coefficient per lane), 2 waves/SIMD"},
         "self_check": {"rows_per_device": 72, "against": "z" * 300, "passed": True},
         "roofline": dict(_roof("k_final_exp_pair"), kernel_ms={"k_miller1h_pair": 9.45, "k_final_exp_pair": 10.973}, traffic_all={"a": 1.0, "b": 2.0}, note="n" * 300),
         "valu": {"bound": "b" * 200, "lane_instructions_per_pairing": 10010953, "achieved": 32.08, "peak": 39.322, "unit": "T lane-instructions/s per GPU", "frac": 0.8158,
                  "nominal": {"x": "q" * 400}},
         "counters": {"file": "profiles/r04_counters.json", "commit": "abcdef0", "source_digest": "0123456789abcdef", "stale": False, "note": "n" * 300},
         "checksum": 123456789, "cpu_baseline": _cpu("pairings/s"),
         "pairing_prepared": {"pairings_per_s": 3553402.6, "ms_per_step": 18.443, "prepare_points_per_s": 16889594.2, "roofline": _roof("k_final_exp_pair"), "note": "n" * 300},
         "verify_bench": {"g2pubs_verifies_per_s": 2463240.7, "g1pubs_verifies_per_s": 2008053.6, "g2pubs_roofline": _roof("k_miller2_pair"), "g1pubs_roofline": _roof("k_miller2_pair"),
                          "g1pubs_with_domain": {"verifies_per_s": 1774708.6, "roofline": _roof("k_cofac2_pair")}, "g2pubs_prepared_keys": {"verifies_per_s": 2645947.8},
                          "g2pubs_cpu_baseline": _cpu("verifies/s"), "g1pubs_cpu_baseline": _cpu("verifies/s"), "note": "n" * 500},
         "msm_bench": {"points": 1 << 20, "note": "n" * 400,
                       **{k: {"value": 23516018.9, "unit": "scalar multiplications/s" if k.endswith("mul") else "points/s", "ms_per_step": 44.59, "roofline": _roof("k_%s_kernel_pair" % k), "cpu_baseline": _cpu("scalar multiplications/s")}
                          for k in ("g1_mul", "g1_msm", "g2_mul", "g2_msm")}},
         "aggregate_bench": {"signatures": 1 << 20, "signatures_per_gpu": 1 << 20, "ms": 153.84, "signatures_per_s": 6815974.1, "exchange": "none (one GPU)", "note": "n" * 300},
         "g2pubs_aggregate_dev_bench": {"signatures": 1 << 20, "ms": 148.9, "signatures_per_s": 7042302.3, "roofline": _roof("k_miller1x2_pair"), "note": "n" * 500, "cpu_baseline": _cpu("signatures/s"),
                                        "prepared_keys": {"ms": 119.49, "signatures_per_s": 8775000.0, "tables_GB": 25.9, "roofline": _roof("k_miller1x2_prep_pair"), "note": "n" * 300}},
         "g1pubs_aggregate_bench": {"signatures": 1 << 18, "ms": 56.14, "signatures_per_s": 4669519.5, "roofline": _roof("k_miller1x2_pair"), "note": "n" * 500, "cpu_baseline": _cpu("signatures/s")},
         "config0": {"workload": "w" * 300, "cpu": dict(_cpu("verifies/s"), wall_s=0.24), "gpu": {"value": 345187.9, "unit": "verifies/s", "ms_one_call": 2.9, "path": "p" * 200},
                     "roofline": _roof("k_lat:verify2"), "verdicts_identical": True, "rejected": 62},
         "inlibrary_bench": {"devices": 1, "shards": 1, "tuples_per_call": 65536, "pairings_per_s": 2.5e6, "g2pubs_verifies_per_s": 2.1e6, "g1pubs_verifies_per_s": 1.8e6, "rccl_ranks": 0},
         "mid_batches": {"pairings_per_s": {str(n): 1234567.8 for n in (8192, 16384, 32768, 65536)}, "note": "n" * 300},
         "reference_shapes": {"x" * 20 + str(i): {"gpu_ms": 1.37, "cpu_ms": 2.9, "note": "n" * 100} for i in range(12)}}
    return d


def test_line_is_under_4k_and_carries_the_contract():
    d = _detail()
    line = bench.compact_line(d, "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in back, k
    assert back["config"]["workload"] and "model" not in back["config"]
    r = back["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "k_final_exp_pair" and r["kernel_ms"] == 10.973 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] == 190567153984 and r["algorithmic_bytes_per_launch"] == 234881024
    c = back["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["single_core_per_s"] == 3559.02 and len(c["sample"]) <= 150
    assert back["valu"]["frac"] == 0.8158 and back["counters"] == {"file": "profiles/r04_counters.json", "stale": False}
    cfg = back["configs"]
    assert sorted(cfg) == ["0", "1", "2", "3", "4"]
    assert sorted(cfg["2"]) == ["g1_msm", "g1_mul", "g2_msm", "g2_mul"]
    for e in [cfg["1"], cfg["3"], cfg["4"]] + list(cfg["2"].values()):
        assert {"value", "unit", "ms", "roofline", "cpu_baseline"} <= set(e)
        assert e["roofline"]["kernel"] and e["roofline"]["frac"] and e["cpu_baseline"]["value"]
    assert cfg["3"]["value"] == 7042302.3 and cfg["3"]["host_buffers_ms"] == 153.84 and cfg["3"]["prepared_keys_ms"] == 119.49
    assert cfg["0"]["cpu_baseline"]["value"] == 34585.34 and cfg["0"]["verdicts_identical"] is True and cfg["0"]["roofline"]["kernel"] == "k_lat:verify2"


def test_line_survives_missing_and_failed_legs():
    d = _detail()
    for k in ("msm_bench", "config0", "g1pubs_aggregate_bench", "pairing_prepared", "cpu_baseline", "mid_batches"):
        d.pop(k)
    d["g2pubs_aggregate_dev_bench"] = {"error": "RuntimeError('x')"}
    line = bench.compact_line(d)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096
    assert sorted(line["configs"]) == ["1", "3"] and line["configs"]["3"]["value"] == 6815974.1          # falls back to the host-buffer leg
    assert line["leg_errors"] == ["g2pubs_aggregate_dev_bench"] and "cpu_baseline" not in line


def test_line_of_a_multi_device_run_names_the_collective_leg():
    d = _detail()
    d["n_gpus"] = d["devices"] = 8
    d["inlibrary_bench"] = dict(d["inlibrary_bench"], devices=8, rccl_ranks=8, tuples_per_call=8 * 65536)
    line = bench.compact_line(d)
    assert line["rccl_ranks"] == 0 and line["inlibrary"]["rccl_ranks"] == 8           # the headline step has no collective; the split host calls do
    assert len(json.dumps(line, separators=(",", ":"))) < 4096


@pytest.mark.gpu
def test_bench_command_prints_one_parsable_line():
    """the driver's command at its smallest: the LAST stdout line parses, is < 4 KB and names every config"""
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
    assert len(lines[0]) < 4096
    j = json.loads(lines[0])
    assert j["metric"].startswith("BLS12-381 pairings/sec") and j["value"] > 1e6 and j["steps"] == 2 and j["warmup"] == 1 and j["n_gpus"] == 1
    assert abs(j["value"] - 65536 / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01
    assert j["roofline"]["kernel"].startswith("k_") and j["roofline"]["kernel_ms"] > 0 and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["value"] > 0
    assert sorted(j["configs"]) == ["0", "1", "2", "3", "4"], j.get("leg_errors")
    assert "leg_errors" not in j, j["leg_errors"]
    for k in ("0", "3", "4"):
        assert j["configs"][k]["cpu_baseline"]["value"] > 0 and j["configs"][k]["roofline"]["kernel"]
    detail = json.load(open(os.path.join(ROOT, j["detail"])))
    assert detail["value"] == j["value"] and "reference_shapes" in detail


@pytest.mark.gpu
def test_two_logical_devices_through_the_bench_command():
    """Multi-device readiness without the hardware: `python bench.py --gpus 2` as the driver would run it on an N-GPU node,
    on ONE GPU under the BLSMI_DEVICE_ALIAS test hook -- two logical devices behind the C ABI, the headline step on both, the split host entry
    points with the bitmap exchange, the sharded 2^20-signature VerifyAggregate with its partial-product exchange.  The line must parse and say
    that these are logical devices (no scaling curve is claimed)."""
    env = dict(os.environ)
    env["BLSMI_DEVICE_ALIAS"] = "0,0"
    env.pop("BLSMI_SHARDS", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["devices"] == 2 and j["aliased_devices"] is True and "ALIASED-DEVICES" in j["library"] and "devices=2" in j["library"]
    assert j["scaling_measured"] is False                                   # the line itself says that no scaling was measured
    assert "NO scaling curve" in j["launch"]
    assert "leg_errors" not in j, j.get("leg_errors")
    assert abs(j["value"] - 2 * 65536 / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01
    il = j["inlibrary"]
    assert il["devices"] == 2 and il["rccl_ranks"] == 2 and il["aliased_devices"] is True and il["tuples_per_call"] == 2 * 65536
    assert il["pairings_per_s"] > 0 and il["g2pubs_verifies_per_s"] > 0 and il["g1pubs_verifies_per_s"] > 0
    assert il["sharded_aggregate"]["signatures"] == 1 << 20 and il["sharded_aggregate"]["ms"] > 0
    detail = json.load(open(os.path.join(ROOT, j["detail"])))
    assert detail["inlibrary_bench"]["shards"] == 2 and "stand-in" in detail["inlibrary_bench"]["collective"]
