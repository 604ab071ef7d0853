"""Two DatasetTrainer ranks on one card (gloo exchange), started as FRESH processes before anything in this pytest process has
initialised the GPU (this file sorts first; conftest only counts devices): the N > 1 step — two captured graphs around one
eager all-reduce — must equal the six-eager-launch step bit for bit, on both ranks, in both precisions."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd"))

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_two_rank_step_graph_equals_eager_bitwise(tmp_path, prec):
    from tnerf import launch                       # no torch import, no GPU call
    out = tmp_path / f"two_rank_{prec}.json"
    rc = launch.spawn_ranks(2, [os.path.join(ROOT, "tests", "workers", "two_rank_graph_worker.py"), str(out), prec], timeout=600,
                            extra_env={"TNERF_SHARE_GPU": "1"})
    assert rc == 0, f"rank processes failed with {rc}"
    r = json.loads(out.read_text())
    assert r["world"] == 2 and r["rays_local"] == 45
    assert r["same_losses"] and r["same_weights"] and r["same_packed"] and r["same_across_ranks"], r
    # the sharded run follows the one-rank run on the whole batch (shard sums in another order: rounding only)
    for a, b in zip(r["losses_graph"], r["losses_one_rank"]):
        assert abs(a - b) <= (1e-5 if prec == "fp32" else 2e-3) * abs(b), (a, b)
    assert r["max_dev_vs_one_rank"] <= (3 if prec == "fp32" else 6) * 5e-4 * 1.01
    # the parity path (torch-drawn batch, FusedTrainer.step_camera) sharded the same way
    assert r["parity_same_across_ranks"] and r["parity_max_dev_vs_one_rank"] <= (5e-5 if prec == "fp32" else 2e-3), r

