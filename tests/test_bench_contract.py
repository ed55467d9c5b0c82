"""The bench line the driver parses (committed in profiles/ from the last GPU run of `python bench.py`) carries every field of the
contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config{workload}, `roofline` of the dominant kernel and the `cpu_baseline` leg.  CPU-only: checks the committed line and the CLI."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_committed_dag_bench_line_has_the_contract_fields():
    d = json.loads(open(_latest("r0*_bench_headline.json")).read().strip().split("\n")[-1])     # `python bench.py`, the driver's command
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes"]          # PMC bytes cannot be below the algorithmic ones
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # the headline is BASELINE.json's two-part metric: S2ST utterances/s (value) and the dag_loss fwd+bwd time of C2 beside it;
    # ms_per_step and value describe the SAME step (B utterances per step on one GPU)
    assert d["unit"] == "utt/s" and d["dag_loss_fwd_bwd_ms_per_batch"] > 0
    assert abs(d["value"] - 1e3 * d["config"]["batch_per_gpu"] / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["c1"]["dag_loss_fwd_bwd_ms"] > 0 and d["c1"]["launch_status"] == 0         # BASELINE configs[0] on the HIP ops, same run


def test_bench_cli_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def _json_line(stdout):
    lines = [l for l in stdout.strip().split("\n") if l.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_bench_self_spawn_world2_gloo_rehearsal():
    """`bench.py --gpus 2` started WITHOUT a launcher re-executes itself under torch.distributed.run (127.0.0.1 rendezvous), builds the
    process group, runs the probe all-reduce, the barriers and the max-over-ranks reduction, shards a pool by length and prints ONE line
    from rank 0.  No kernels (`--workload plumbing`): this is the launcher path the driver's 2/4/8-GPU runs take, rehearsed on CPU."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["DSP_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "plumbing"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["shard_spread"] <= 0.05
    # r06: per-rank record for the scaling runs — device, shard sizes, cost and its spread, each rank's own clock
    sd = d["scaling_diag"]
    assert [r["rank"] for r in sd["ranks"]] == [0, 1] and sd["backend"] == "gloo"
    assert all(r["utterances_per_step"] == [32] and r["src_frames_per_step"][0] > 0 and r["dag_cells_per_step"][0] > 0 for r in sd["ranks"])
    assert sd["src_frames_spread"] <= 0.05 and sd["dag_cells_spread"] <= 0.10 and 0.0 <= sd["local_time_spread"] <= 1.0
    assert all(r["local_ms_per_step"] > 0 for r in sd["ranks"])
    assert out.stderr.count("all-reduce ok") == 2
    assert len([l for l in out.stdout.split("\n") if l.startswith("{")]) == 1          # rank 0 only


def test_bench_under_a_launcher_world2_gloo_rehearsal():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (RANK / WORLD_SIZE from the env)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env["DSP_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "plumbing"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _json_line(out.stdout)["n_gpus"] == 2
    # a world that does not match --gpus is refused, not silently re-labelled
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "plumbing"], capture_output=True, text=True,
                         timeout=300, env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_live_traffic_reads_the_per_dispatch_counters(tmp_path, monkeypatch):
    """bench.live_dp_traffic: two `rocprofv3 --pmc` child passes, per-dispatch mean of the DP kernel, FETCH doubled + WRITE, in bytes.  A stand-in
    rocprofv3 on PATH writes the counter CSV the real one writes (no GPU here); a failing pass gives None (the committed record is used then)."""
    import stat
    import types
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]; c = a[a.index("--pmc") + 1]
if os.environ.get("FAKE_FAIL") == c: sys.exit(3)
os.makedirs(os.path.join(d, "host", "1"), exist_ok=True)
v = {"FETCH_SIZE": (1000.0, 3000.0), "WRITE_SIZE": (500.0, 700.0)}[c]
with open(os.path.join(d, "host", "1", "p_counter_collection.csv"), "w") as f:
    f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
    k = '"void dsp::dag_strip4g_kernel<256, 0, false>(dsp::GStripParams)"'
    f.write(f"1,{k},{c},{v[0]}\\n2,{k},{c},{v[1]}\\n")
    f.write(f'3,"void dsp::lsg_fwd_regl_kernel<float, 8>(P)",{c},999999\\n')
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.delenv("DSP_BENCH_CHILD", raising=False)
    args = types.SimpleNamespace(tr=32, dag_batch=2, graph_len=64, tgt_len=8, vocab=16)
    r = bench.live_dp_traffic(args, 32)
    assert r is not None and r["FETCH_SIZE_KB_raw"] == 2000.0 and r["WRITE_SIZE_KB"] == 600.0
    assert r["hbm_bytes_per_launch"] == int((2 * 2000.0 + 600.0) * 1024) and "this run" in r["source"]
    monkeypatch.setenv("FAKE_FAIL", "WRITE_SIZE")
    assert bench.live_dp_traffic(args, 32) is None
    monkeypatch.delenv("FAKE_FAIL")
    monkeypatch.setenv("DSP_BENCH_CHILD", "1")                 # a child pass never recurses
    assert bench.live_dp_traffic(args, 32) is None
