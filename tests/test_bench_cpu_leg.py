"""bench.py's cpu_baseline leg (the oracle timed on host cores) on a tiny sample: keys, units and a positive rate."""
import importlib.util
import os


def test_cpu_baseline_leg_runs_and_reports():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(2048, seconds_target=0.3)
    assert out["kind"] == "port" and out["unit"] == "agent-steps/s" and out["cores"] >= 1
    assert out["value"] > 0 and out["env_steps_per_s"] > 0 and out["single_thread_env_steps_per_s"] > 0
    assert abs(out["value"] - 4 * out["env_steps_per_s"]) < 1e-3 * out["value"]   # level 6: 4 agents
    assert "oracle/lle_oracle.c" in out["sample"]
    assert bench.ALGO_BYTES_PER_ENV_STEP == 12 * 12 * 13 + 2 * (8 + 3 + 1 + 12) + 4 + 4 + 9
