"""Two-rank run of bench.py through its own launcher (counterpart of tests/run_train_2rank.py).  Not collected by
pytest (a GPU-initialised pytest process must not spawn GPU children on the pool's boxes); launched directly:

    python tests/run_bench_2rank.py

Backend: "nccl" (= RCCL over xGMI) when at least two GPUs are visible, else "gloo" with both ranks on the one card (RCCL
refuses two ranks on one device).  Checks the JSON line (n_gpus = 2, the backend the communicator really used, rccl_world /
rccl_version when it is RCCL) and that the LAUNCHING parent never mapped the HIP runtime or torch: on this pool a process that
has initialised the GPU must not fork + exec children."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor HIP)


def main():
    ngpu = bench.visible_gpus()
    backend = "nccl" if ngpu >= 2 else "gloo"
    with tempfile.TemporaryDirectory() as td:
        maps = os.path.join(td, "parent_maps.txt")
        env = dict(os.environ, PRIORFLOW_BENCH_BACKEND=backend, PRIORFLOW_BENCH_PARENT_MAPS=maps)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, timeout=900, check=True).stdout.decode()
        line = json.loads(out.strip().splitlines()[-1])
        mapped = open(maps).read()
    assert line["n_gpus"] == 2 and line["config"]["collective_backend"] == backend, line["config"]
    if backend == "nccl":
        assert line["config"]["rccl_world"] == 2 and line["config"]["rccl_version"], line["config"]
    assert "libamdhip64" not in mapped and "libtorch" not in mapped, "the launching parent mapped the GPU runtime:\n" + mapped
    print(f"bench.py --gpus 2 over {backend} ({ngpu} GPU(s) visible): {line['value']} pairs/s, parent maps clean; "
          f"rccl_world={line['config']['rccl_world']} rccl_version={line['config']['rccl_version']}")


if __name__ == "__main__":
    main()
