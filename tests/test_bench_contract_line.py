"""The driver parses bench.py's LAST stdout line; round 5's 23.7 KB line came back as `parsed: null`.  The line builder is run
here on the canned output of a real run (profiles/r05_bench.json) and on degenerate inputs: strict JSON, < 6000 bytes, every contract key."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline"]


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))


def _strict(text):
    def bad(c):
        raise ValueError("non-strict constant " + c)
    return json.loads(text, parse_constant=bad)


def test_contract_line_of_a_real_run_is_short_and_strict():
    out = _canned()
    assert len(json.dumps(out)) > 20000          # the canned output is the line that failed to parse
    text = bench.contract_line(out)
    assert "\n" not in text and len(text) < 6000, len(text)
    line = _strict(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["value"] == float("%.6g" % out["value"]) and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert abs(line["ms_per_step"] - out["ms_per_step"]) < 1e-4 * out["ms_per_step"]
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") or out["roofline"]["bound"] == "int_mul"    # r05's canned file predates the ADVICE fix
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert "frac_of_binding_roof" in rf and abs(rf["frac_of_binding_roof"] - line["roofline_int"]["frac"]) < 1e-9
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert line["create_proof_k21_pairing_shape"]["equals_committed_oracle_prover_digest"] is True
    assert line["roofline_proof"]["int"]["frac"] > 0 and line["extra"] == "bench_extra.json"


def test_contract_line_survives_nan_errors_and_huge_blocks():
    out = _canned()
    out["value"] = float("nan")
    out["roofline"]["traffic"] = None
    out["cpu_baseline"] = {"error": "x" * 100000}
    out["msm_2_20"] = {"error": "boom"}
    out["create_proof_k21_pairing_shape"] = {"error": "y" * 50000}
    out["config"]["workload"] = "w" * 5000
    out["config"].pop("workload_short", None)
    out["huge"] = ["z" * 100] * 10000
    text = bench.contract_line(out)
    assert len(text) < 6000
    line = _strict(text)
    assert line["value"] is None and line["roofline"]["traffic"] is None and "huge" not in line
    assert len(line["config"]["workload"]) <= 300


def test_contract_line_multi_gpu_fields():
    out = _canned()
    out["n_gpus"] = 8
    out["comm"] = {"transport": "RCCL (ncclAllGather ...)", "ranks": [{"rank": i} for i in range(8)], "distinct_gpus": 8, "sharded_proof_exchanges": {"a": [1] * 1000}}
    out["sharded_bytes_equal_unsharded"] = True
    out["independent_proofs_per_gpu"] = {"what": "w" * 500, "ms_per_step": 13.0, "value": 3.2e8, "unit": "constraints/s", "scaling": "weak"}
    out["create_proof_k21_pairing_shape"]["sharded"] = {"seconds": 0.013, "what": "q" * 900, "equals_committed_oracle_prover_digest": True}
    line = _strict(bench.contract_line(out))
    assert line["comm"] == {"transport": "RCCL (ncclAllGather ...)", "distinct_gpus": 8, "ranks": 8}
    assert line["independent_proofs_per_gpu"]["scaling"] == "weak" and line["sharded_bytes_equal_unsharded"] is True
    assert line["create_proof_k21_pairing_shape"]["sharded"]["seconds"] == 0.013


def test_sig_rounding():
    assert bench._sig(1234567.891) == 1234570.0 and bench._sig(True) is True and bench._sig(7) == 7
    assert bench._sig(float("inf")) is None and bench._sig({"a": [math.pi]}) == {"a": [3.14159]}
