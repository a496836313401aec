"""-m gpu: the bench.py line itself, in the launch forms a driver may use -- small sizes, subprocesses with
their own library instance, so this module can run at any point of the session.

* `python bench.py --gpus 2` with NO launcher (WORLD_SIZE unset): one process over two devices (logical devices of GPU 0 when only
  one GPU is visible) must exit 0 and print ONE JSON line with n_gpus = 2, the weak-scaling value, `strong` (one proof by both routes
  equal to the single-device proof, the 2^22-term MSM equal to its naive-loop golden), `rccl` (ranks seen, collectives) and `devices`
  (ordinals, peer-access matrix).
* the driver's N-rank command line (`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`) with both ranks on the
  one GPU (GS_BENCH_SHARE_GPU=1, gloo): the same keys in the line.
* N = 1 stays what it was: no `strong`, `roofline` and `cpu_baseline` present."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd, env):
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_multi(d):
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["unit"] == "constraints/s"
    s = d["strong"]
    key = "prove_sharded_2^12"
    assert s[key]["px_route"]["proof_equals_single_device"] is True and s[key]["px_route"]["ms_per_step"] > 0
    assert s[key]["values_route"]["proof_equals_single_device"] is True
    assert s["msm_sharded_2^22"]["equals_naive_loop_golden"] is True and s["msm_sharded_2^22"]["terms"] == 1 << 22
    assert "watchdog" not in s
    assert "ranks_seen" in d["rccl"] and "collectives" in d["rccl"] and "mode" in d["rccl"]
    assert d["devices"]["logical_to_physical"] and isinstance(d["devices"]["can_access_peer"], list)
    assert "accepted" in d["proof_verified"]


def test_gpus_2_without_a_launcher_prints_the_whole_multi_gpu_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _line([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "1", "--log2n", "12"], env)
    _check_multi(d)
    assert d["launch"].startswith("one process") and d["rccl"]["local"] is True
    assert d["rccl"]["ranks_seen"] >= 1 and d["rccl"]["collectives"] >= 3          # the records of every sharded step went through ncclAllGather
    assert d["strong"]["prove_sharded_2^12"]["px_route"]["used_rccl"] is True


def test_two_ranks_under_torch_distributed_run_print_the_same_keys():
    env = dict(os.environ, GS_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    port = str(29600 + os.getpid() % 300)
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
               "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "1", "--log2n", "12", "--cpu-log2n", "0"], env)
    _check_multi(d)
    assert d["launch"].startswith("torch.distributed.run") and "gloo" in d["strong"]["exchange"]


def test_multi_ranks_reexecutes_the_plain_command_line_under_the_launcher():
    """`python bench.py --gpus 2 --multi ranks` with no WORLD_SIZE: bench.py re-executes itself under torch.distributed.run (one rank per
    GPU; both on GPU 0 here) and the line comes from rank 0 of that job."""
    env = dict(os.environ, GS_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    d = _line([sys.executable, "bench.py", "--gpus", "2", "--multi", "ranks", "--steps", "2", "--warmup", "1", "--reps", "1", "--log2n", "12", "--cpu-log2n", "0",
               "--no-strong"], env)
    assert d["n_gpus"] == 2 and d["launch"].startswith("torch.distributed.run") and "strong" not in d and d["value"] > 0


def test_single_gpu_line_keeps_its_contract():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--reps", "1", "--log2n", "12", "--cpu-log2n", "8", "--no-extras"], dict(env, GS_BENCH_NO_LIVE_PMC="1"))
    assert d["n_gpus"] == 1 and "strong" not in d and d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["vs_baseline"] is None
    assert d["config"]["workload"] == "groth16_prove_2^12_constraints_per_gpu" and "proof_check" in d
