"""CPU unit test of bench.py's self-launch: `python bench.py --gpus N` from a bare shell must
start the N ranks as child processes with the driver's own torch.distributed.run command, relay
exactly rank 0's JSON line and propagate a failure."""
import importlib.util
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def test_launcher_command_is_the_drivers():
    bench = load_bench()
    cmd = bench.launcher_command(8, 29511, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[0] == sys.executable
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py")
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_launch_relays_rank0_line_and_status(capsys, avr_lib):
    bench = load_bench()
    args = types.SimpleNamespace(gpus=4)
    seen = {}

    def fake_run(command, env=None, stdout=None, text=None):
        seen["command"], seen["env"] = command, env
        line = json.dumps({"metric": "Mray-samples/s", "value": 1.0, "n_gpus": 4})
        return types.SimpleNamespace(returncode=0, stdout="rank noise\n" + line + "\n")

    assert bench.launch_ranks(args, ["--gpus", "4"], run=fake_run) == 0
    out = capsys.readouterr()
    assert out.out.count("\n") == 1 and json.loads(out.out)["n_gpus"] == 4
    assert "rank noise" in out.err
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "--nproc-per-node=4" in seen["command"]

    def failing_run(command, env=None, stdout=None, text=None):
        return types.SimpleNamespace(returncode=3, stdout="")

    assert bench.launch_ranks(args, ["--gpus", "4"], run=failing_run) == 3

    def silent_run(command, env=None, stdout=None, text=None):
        return types.SimpleNamespace(returncode=0, stdout="no result\n")

    assert bench.launch_ranks(args, ["--gpus", "4"], run=silent_run) == 1


def test_bare_multi_gpu_start_spawns_children_before_touching_the_gpu(tmp_path):
    """End to end without a GPU: the parent must get as far as starting the ranks (which then fail
    here, on a machine without a HIP device) and exit non-zero -- not die in its own process with
    'launch with torch.distributed.run', and not import torch before the launch."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["PYTHONPATH"] = ROOT
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, runpy\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--config', 'tiny',"
        " '--no-cpu-baseline']\n"
        "import subprocess\n"
        "real = subprocess.run\n"
        "def spy(cmd, **kw):\n"
        "    print('SPAWN', 'torch' in sys.modules, cmd[1:3], flush=True)\n"
        "    raise SystemExit(7)\n"
        "subprocess.run = spy\n"
        f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    done = subprocess.run([sys.executable, str(probe)], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, cwd=str(tmp_path))
    assert done.returncode == 7, done.stderr
    assert "SPAWN False ['-m', 'torch.distributed.run']" in done.stdout


def test_profiled_traffic_is_only_quoted_for_the_kernels_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC summary: bench.py quotes it only when the
    summary's digest equals the sha256 of this tree's kernel and driver sources -- an edited kernel
    makes the line say "stale" instead of quoting the old bytes."""
    bench = load_bench()
    args = types.SimpleNamespace(config="config4", field="smooth", transparency=0.97, width=0,
                                 height=0, antialiasing=1, orbit=0, fly_through=False,
                                 cache_classification=False)
    monkeypatch.setattr(bench, "PMC_DIR", str(tmp_path))
    summary = tmp_path / "pmc_config4_translucent.txt"
    body = ("classify_kernel\n  FETCH_SIZE    n=5 mean=1000\n  WRITE_SIZE    n=5 mean=100\n"
            "render_runs_kernel\n  FETCH_SIZE    n=5 mean=200\n  WRITE_SIZE    n=5 mean=30\n")
    assert bench.profiled_traffic(args, 1)[0] is None              # no summary at all
    summary.write_text("# sources sha256: %s  (x)\n" % bench.kernel_sources_sha256() + body)
    traffic, source = bench.profiled_traffic(args, 1)
    assert traffic == int((2 * 1000 + 100 + 200 + 30) * 1024) and source == str(summary)
    summary.write_text("# sources sha256: %s  (x)\n" % ("0" * 64) + body)
    traffic, source = bench.profiled_traffic(args, 1)
    assert traffic is None and source.startswith("stale")
    summary.write_text(body)                                       # a summary without a digest
    assert bench.profiled_traffic(args, 1)[0] is None
    # every workload the sheet quotes has a summary of its own, found by its key
    assert bench.workload_key(args, 1) == "config4_translucent"
    args.transparency = 0.0
    assert bench.workload_key(args, 1) == "config4_opaque"
    args.config, args.transparency = "config2", 0.97
    assert bench.workload_key(args, 1) == "config2_translucent"
    assert bench.profiled_traffic(args, 1)[0] is None              # (none written for it here)
    args.config, args.antialiasing = "config5", 4
    assert bench.workload_key(args, 1) == "config5_translucent"
    args.orbit = 16                                                # not a profiled workload
    assert bench.profiled_traffic(args, 1) == (None, None)
    assert bench.workload_key(args, 2) is None
