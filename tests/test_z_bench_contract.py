"""GPU tests (-m gpu) of the MEASUREMENT TOOL: bench.py's JSON line, its N > 1 step as two ranks, the RCCL leg.

This file sorts behind every parity file (test_gpu_configs.py, test_gpu_parity.py, test_lut_integral_pin.py, test_reference_pins.py) on purpose: under the driver's
`pytest -x` a regression of bench.py must not hide the kernel tests (GPUTEST_r04: a counter-file commit turned test 5 of 201 red and 196 parity tests never ran).
"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("launcher", ["driver", "self"])
def test_two_rank_bench_step_gathers_the_unsharded_frame(tmp_path, launcher):
    """bench.py's N > 1 step end to end -- tile-sharded render, pack, gather to rank 0, fh_unpack_shards -- as two fresh processes (one per
    rank) sharing this one GPU, with gloo in place of RCCL: started by torch.distributed.run exactly as the driver starts them ("driver"), and by bench.py itself from
    `python bench.py --gpus 2` with no launcher and no WORLD_SIZE ("self": the parent never touches the GPU, the ranks are its children).  --check-frame
    makes rank 0 compare the gathered frame bit for bit with an unsharded render of as many samples."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "4", "--check-frame", "--no-cpu-baseline"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    if launcher == "self":
        cmd = [sys.executable] + tail
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    assert "frame gathered from 2 ranks bit-identical to the unsharded render: True" in run.stderr
    assert "rank-0 shard bit-identical to the unsharded render: True" in run.stderr
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0 and "gather" in out["config"]["gather"]


@pytest.mark.parametrize("cfg,spp", [(1, 8), (2, 12)])
def test_bench_line_contract(tmp_path, cfg, spp):
    """one short run of bench.py per kind of dominant kernel (shade on the Cornell box, traversal on the soup): ONE JSON line with the metric, the roofline record
    (no fraction above 1, a bound that is named, launch times from HIP events), parity against the checker (bit-identical crop), the small-launch latencies and the
    whole-frame figures"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(cfg), "--spp", str(spp), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "parity", "latency", "whole_frame", "rates"):
        assert k in out, k
    assert out["unit"] == "Msamples/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["dtype"] == "f32" and out["vs_baseline"] is None and "workload" in out["config"]
    assert abs(out["value"] - 1920 * 1080 * spp / (out["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * out["value"]
    r = out["roofline"]
    assert r["bound"] in ("valu_issue", "hbm") and r["unit"] in ("G SIMD issue cycles/s", "GB/s") and r["peak"] > 0 and r["avg_launch_ms"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    sys.path.insert(0, root)
    import bench
    assert "fractions_refused" not in out and bench.refuse_bad_fracs(json.loads(lines[0])) == []  # no fraction of a roof outside [0, 1] anywhere in the line
    # these short runs submit passes of 8 / 4 samples; every committed counter file is about passes of 43 and more: the line must carry NO counter-derived field and
    # must say which file it refused (round 4 took the nearest file and printed frac 2.86)
    cu = r["counters_unusable"]
    assert abs(cu["then"] - cu["now"]) > 0.02 * cu["now"] and cu["then"] > 2 * cu["now"] and cu["file"].startswith("profiles/")
    for k in ("traffic", "frac_hbm_measured", "hbm_traffic_frac", "valu", "vl1d", "counters_from", "counters_stale"):
        assert r.get(k) is None, k
    if cfg == 1:
        assert r["bound"] == "hbm" or r["kernel"].startswith("k_trace")  # the shade kernel's issue fraction needs its counters: without them, the SURVEY 8(d) bytes
    if cfg == 2:
        assert r["bound"] == "valu_issue" and r["kernel"].startswith("k_trace") and 1.0 < r["clock_ghz_in_kernel"] < 2.6
        assert 0.0 < r["lane_utilisation"]["node_tests"] <= 1.0 and 0.0 < r["lane_utilisation"]["triangle_tests"] <= 1.0
    # round 6: the fixed-name fields and the shade kernels' own record, whichever kernel dominates
    for k in ("frac_hbm_counters", "frac_survey_8d_over_hbm_peak", "ta_busy", "valu_busy", "bound_verdict"):
        assert k in r, k
    assert r["frac_hbm_counters"] is None and r["frac_survey_8d_over_hbm_peak"] == (r["frac"] if r["bound"] == "hbm" else r["algorithmic_gbs_over_hbm_peak"]) and isinstance(r["bound_verdict"], str)
    sh = out["shade"]
    assert sh["ms_per_step_alone"] > 0 and sh["shaded_ghits_per_s_alone"] > 0 and sh["kernels"] and all(k["vgprs"] > 0 and 1 <= k["waves_per_simd"] <= 8 for k in sh["kernels"])
    assert (sh.get("counters_from") and "valu_lane_utilisation" in sh) or "valu_busy" not in sh
    p = out["parity"]
    assert p["rmse"] == 0.0 and p["bit_identical_pixels"] == 1.0 and p["pixels"] == 8 * 1920
    lat = out["latency"]
    assert 0.0 < lat["spp1"]["min_ms"] <= lat["spp1"]["median_ms"] < lat["spp16"]["median_ms"] * 4
    if cfg == 2:  # the default configuration appends the general-scene leg (configs[3]) outside the headline's timed region
        g = out["general_scene"]
        assert g["workload"].startswith("configs[3]") and g["msamples_per_s"] > 0 and g["parity"]["bit_identical_pixels"] == 1.0 and g["parity"]["rmse"] == 0.0
        assert g["roofline"]["kernel"].startswith("k_trace") and 0.0 < g["roofline"]["frac"] <= 1.0 and g["roofline"]["kernel_info"]["vgprs"] > 0
        assert 0.0 < g["latency"]["spp1"]["median_ms"] < g["latency"]["spp16"]["median_ms"]
        assert r["issue_model"]["stale"] in (True, False)
        gr = g["roofline"]  # the general-scene leg of this short run is 512 spp in five passes, which is what its counter file saw: used, and then it says from where
        assert (gr.get("counters_from") and "counters_stale" in gr and gr.get("traffic")) or (gr.get("counters_unusable") and gr.get("traffic") is None and "frac_hbm_measured" not in gr)


def test_rccl_leg_on_two_gpus(tmp_path):
    """The RCCL leg itself -- dist.init_process_group("nccl", device_id=...) and dist.gather of device tensors, bench.py's N > 1 path as the driver's multi-GPU run
    takes it -- needs one GPU per rank: runs wherever two or more are visible (the one-GPU boxes of the development pool skip it; the two-rank test above covers
    everything but the transport there).  First the collective alone (tools/rccl_gather_probe.py), then the whole bench step with --check-frame."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("the RCCL leg needs two visible GPUs (one process per GPU)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FH_BENCH_BACKEND", None)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    run = subprocess.run(launch + ["--master-port", str(port()), os.path.join(root, "tools", "rccl_gather_probe.py")], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    probe = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert probe["backend"] == "nccl" and probe["world"] == 2 and probe["data_ok"]
    run = subprocess.run(launch + ["--master-port", str(port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "4", "--check-frame", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    assert "frame gathered from 2 ranks bit-identical to the unsharded render: True" in run.stderr
    out = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["gather"].startswith("RCCL")
