"""bench.py's multi-GPU control flow rehearsed on the CPU with a stand-in worker (DQMC_BENCH_FAKE_WORKER: no GPU, no
library): `python bench.py --gpus N` must start by itself -- the way the driver starts the N = 1 line -- and the
torch.distributed.run launch (one rank per GPU, gloo here) must print the same line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, DQMC_BENCH_FAKE_WORKER="1", DQMC_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    return json.loads(lines[0])


def test_bench_starts_by_itself_for_two_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workers", "2",
                          "--batch", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=120, env=_env())
    assert out.returncode == 0, out.stderr
    r = _line(out)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert len(r["per_gpu"]) == 2 and r["config"]["launch"] == "self"
    # whole-job value: all chains of both GPUs over the slowest participant's time
    assert abs(r["value"] - 2 * 2 * 4 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]
    assert r["value"] <= sum(r["per_gpu"]) * (1 + 1e-9)
    assert r["one_context_sweeps_per_s"] > 0 and r["single_chain_sweeps_per_s"] > 0
    assert r["metric"].startswith("DQMC sweeps/sec") and r["unit"] == "sweeps/s" and r["dtype"] == "f64" and r["vs_baseline"] is None


def test_bench_single_gpu_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=120, env=_env())
    assert out.returncode == 0, out.stderr
    r = _line(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "per_gpu"):
        assert k in r, k
    assert r["n_gpus"] == 1 and "workload" in r["config"] and "model" not in r["config"]


def test_bench_under_torch_distributed_run_gloo():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workers", "1",
                          "--batch", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=_env(DQMC_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and len(r["per_gpu"]) == 2 and r["config"]["launch"] == "torch.distributed.run"


def test_roofline_block_reports_the_binding_roof():
    sys.path.insert(0, ROOT)
    import bench
    n, m, B = 512, 100, 128
    prof = {"bmult": (100.0, 210), "gemm": (100.0, 40), "decomp": (300.0, 1800), "decide": (150.0, 800), "other": (1.0, 20),
            "gather": (50.0, 800), "flush": (250.0, 800), "jacobi": (300.0, 1800), "svd_calls": 0, "svd_sweeps_total": 0, "svd_sweeps_max": 0,
            "qr_calls": 20, "gemm_flops": 40 * 8.0 * n ** 3 * B, "decomp_round_ms": 200.0, "decomp_rounds": 900,
            "blocks_nonempty": 500 * B, "updates_accepted": 500 * B * 28, "chains": B}
    _, roofs, whole = bench.rooflines(prof, n, m, B, {"flush": {"hbm_bytes_per_launch": 1.0e9, "note": "pmc"}})
    by = {r["family"]: r for r in roofs}
    assert roofs[0]["device_ms"] >= roofs[1]["device_ms"]
    fl = by["flush"]
    # K = 56 per block: 8 n^2 K flop against 32 n^2 bytes -> the matrix cores are the binding roof
    assert fl["bound"] == "mfma" and fl["frac"] == fl["mfma_frac"] > fl["hbm_frac"] and fl["traffic"] == 1.0e9
    assert by["gemm"]["bound"] == "mfma" and by["bmult"]["bound"] == "hbm" and by["bmult"]["traffic"] is None
    assert by["decide"].get("latency_bound") and by["qr_rest"].get("latency_bound")
    assert 0 < whole["hbm_frac"] < 1 and 0 < whole["mfma_frac"] < 1
    for r in roofs:
        assert r["frac"] == max(r["hbm_frac"], r["mfma_frac"]) and r["peak"] in (bench.HBM_PEAK_GBS, bench.MFMA_F64_PEAK_TF)


def test_bench_exits_nonzero_when_a_worker_dies():
    """A worker process that dies in the timed region (here: the stand-in worker of GPU 1 exits with status 7) must end
    `python bench.py --gpus 2` with a non-zero status and the worker's exit code in the message -- and without a JSON line
    (a scaling run must not report the surviving GPUs' rate as the job's)."""
    env = _env(DQMC_BENCH_FAKE_DIE="1:7")
    env.pop("DQMC_BENCH_ONE_DEVICE")           # the stand-in workers must see their real device ordinals
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0
    assert "exit code 7" in out.stderr, out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_config_selects_the_other_baseline_workloads():
    """`--config` runs the other BASELINE.json workloads with the same JSON shape (config 5 without the flux the reference rejects
    for opdim = 3, /root/reference/src/detsdwparams.cpp:57-60); the default stays the configuration the metric is quoted on."""
    for cfg, label, chains in (("o3_L24_b20", "SDW-O3 L=24 beta=20", 8), ("o2_L16_b20", "SDW-O2 L=16 beta=20", 256), (None, "SDW-O2 L=16 beta=10", 512)):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + (["--config", cfg] if cfg else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=_env())
        assert out.returncode == 0, out.stderr
        r = _line(out)
        assert label in r["metric"] and r["config"]["replicas_per_gpu"] == chains and "workload" in r["config"]
        assert r["dtype"] == "f64" and r["vs_baseline"] is None
