"""bench.py's multi-process paths, rehearsed on one GPU: --gpus N starts N ranks of the script (all on GPU 0
with BENCH_REHEARSAL_ONE_GPU=1) and rank 0 prints one JSON line for the whole job."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = dict(os.environ, BENCH_REHEARSAL_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # one JSON line and nothing else on standard output
    return json.loads(lines[0]), r.stderr


def test_gpus_2_starts_two_ranks_batch():
    out, _ = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "1024", "--height", "512", "--batch", "8",
                       "--resident", "4", "--per-launch", "4", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16
    assert out["roofline"]["kernel"] == "k_coded" and out["roofline"]["launches_timed"] == 4
    assert out["value"] > 0


@pytest.mark.parametrize("extra,protocol,gather", [((), "chain", "device"), (("--gather", "host"), "chain", "host"),
                                                   (("--stripe-protocol", "host"), "host", "device")])
def test_gpus_3_stripes_equal_the_whole_image(extra, protocol, gather):
    """Three processes, one stripe each: the stripe chain (the ranks' kernels hand the scalars on through a mailbox of
    rank 0 opened by the others; the pieces meet in rank 0's device memory, or in its page-locked host memory, the
    fallback where GPUs cannot open each other's memory) and the earlier protocol with host exchanges between the
    steps; the stitched scan is compared with the whole-image encode."""
    out, err = run_bench("--gpus", "3", "--config", "5", "--steps", "2", "--warmup", "1", "--width", "2048", "--height",
                         "1008", "--passes", "3", "--no-cpu-baseline", "--verify", *extra)
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["config"]["stripes"] == 3
    assert out["config"]["stripe_protocol"] == protocol and out["config"]["gather"] == gather
    assert out["config"]["control_plane"] == "gloo"
    assert "verified" in err


def test_gpus_5_stripes_with_ranks_that_have_no_rows():
    out, err = run_bench("--gpus", "5", "--config", "5", "--steps", "1", "--warmup", "1", "--width", "1024", "--height",
                         "40", "--passes", "2", "--no-cpu-baseline", "--verify")
    assert out["config"]["stripes"] == 5 and "verified" in err


def test_one_rank_line_has_the_roofs_and_figures():
    out, _ = run_bench("--steps", "2", "--warmup", "1", "--width", "1024", "--height", "512", "--batch", "32",
                       "--resident", "8", "--per-launch", "8", "--no-cpu-baseline")
    assert out["n_gpus"] == 1
    rf = out["roofline"]
    assert rf["launches_timed"] == 8 and rf["kernel_ms_min"] <= rf["kernel_ms_median"]
    assert rf["fits_in_step"], rf
    assert out["e2e"]["page_locked_host_to_host"] > 0 and out["e2e"]["pageable_host_to_host"] > 0
