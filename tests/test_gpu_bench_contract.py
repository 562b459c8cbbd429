"""The driver's bench.py contract, checked end to end on the GPU box: one JSON line with the agreed keys, launched both
plainly and through torch.distributed.run (the N>1 launcher, here with one rank)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def run(cmd, **env):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", **env})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_and_roofline_fields():
    d = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--clips", "1", "--cpu-frames", "1", "--cpu-runs", "2"])
    assert KEYS <= set(d)
    assert d["metric"].startswith("video-clips/sec") and d["unit"] == "clips/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1.0 and abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["clips_per_step_per_gpu"]) < 1e-2
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    # `achieved` = MFMA FLOP/s issued for the algorithmic work, so frac is a true roofline fraction (<= 1); the
    # direct-convolution-equivalent rate (which a Winograd kernel can push past the MFMA peak) sits under its own key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_tflops"] * r["mfma_issue_ratio"]) < 0.05
    if r["kernel"].startswith("conv3x3_wino24"):
        assert abs(r["mfma_issue_ratio"] - 1 / 3.0) < 1e-3
    elif r["kernel"].startswith("conv3x3_wino"):
        assert abs(r["mfma_issue_ratio"] - 1 / 2.25) < 1e-3
    else:
        assert r["mfma_issue_ratio"] == 1.0
    assert r["traffic"] is None or (r["traffic"] > 0 and r["pmc_source"]["file"].startswith("profiles/")
                                    and r["pmc_source"]["measured_in_this_run"] is False
                                    and r["pmc_source"]["matches_current_sources"] is True)
    if r["pmc_source"] is not None and not r["pmc_source"]["matches_current_sources"]:
        assert r["traffic"] is None and r["mfma_busy_frac_pmc"] is None and "stale" in r["pmc_source"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "clips/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert len(c["runs_s"]) == 3 and c["cpu"]              # 1 warm-up + min of 2 here (--cpu-runs 2)
    assert d["value"] > 10 * c["value"]                # north_star's floor: >= 10x the same-host CPU path
    # the timed step's outputs are checked against the oracle's inside the bench run itself
    p = d["parity"]
    assert p["checked"] is True and p["ok"] is True and p["max_err_of_scale"] <= 1e-3 and p["topk_set_overlap"] >= 0.99
    assert set(p["err_of_scale"]) == {"roi_features", "x3_1b", "match_logits", "bank_rows"}
    # the bank is extractor output (the shop-side path, once, before the timed region), and its first rows were re-derived by the oracle
    assert d["bank"]["source"].startswith("extractor output") and d["bank"]["rows_this_rank"] == d["config"]["gallery"]
    assert d["value_clips1"] > 1.0 and d["full_forward_ms_per_clip"] > 0
    # allocator activity inside the timed region is part of the line (a hipMalloc there stalls the device: DESIGN 3.4)
    assert d["hbm_gb"]["device_allocs_in_timed_region"] >= 0 and d["hbm_gb"]["reserved_growth_in_timed_region"] >= 0.0


def test_timed_region_does_not_grow_the_allocator():
    """Default line (two body streams, side-stream tensors record_stream-ed): with the host held two steps ahead of the device and three
    warm-up steps the caching allocator has reached its plateau before timing starts -- no hipMalloc inside the timed region (before
    the pacing: 16 calls / +47.7 GB over 20 steps, and 100+ ms stalls in a box's first process)."""
    d = run([sys.executable, "bench.py", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-extras"])
    assert d["hbm_gb"]["device_allocs_in_timed_region"] == 0 and d["hbm_gb"]["reserved_growth_in_timed_region"] == 0.0, d["hbm_gb"]
    assert d["hbm_gb"]["allocator_retries"] == 0
    # ... and not because the bench paces its loop: the model's side-stream sections are fork / join (no record_stream), so a host
    # that launches the whole timed region ahead of the device needs no fresh blocks either
    d = run([sys.executable, "bench.py", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-extras"],
            SEAM_BENCH_QUEUE_DEPTH="100000")
    assert d["hbm_gb"]["device_allocs_in_timed_region"] == 0 and d["hbm_gb"]["reserved_growth_in_timed_region"] == 0.0, d["hbm_gb"]


def test_bench_large_gallery_workload_c3():
    """configs[2]: 20 000-product gallery ranked without the [S,G,2] logits in HBM; parity of the ranking vs the oracle."""
    d = run([sys.executable, "bench.py", "--workload", "c3", "--steps", "2", "--warmup", "1", "--clips", "1", "--cpu-frames", "1",
             "--cpu-runs", "1", "--no-roofline"])
    assert d["config"]["gallery"] == 20000 and "20k gallery" in d["metric"] and "configs[2]" in d["config"]["workload"]
    assert d["parity"]["ok"] is True and d["parity"]["topk_set_overlap"] >= 0.99 and d["parity"]["topk_equal"] is True
    m = d["match_stage"]                              # the MFMA similarity + top-k stage, timed on its own
    assert m["queries"] == 32 and m["gallery"] == 20000 and 0 < m["us_per_call"] < 500 and m["queries_redone_with_the_direct_form"] == 0


def test_bench_under_the_multi_gpu_launcher():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--clips", "2",
             "--workload", "c4", "--no-cpu-baseline", "--no-roofline"])
    assert d["config"]["gallery"] == 50000 and "configs[3]" in d["config"]["workload"]
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["config"]["clips_per_step_per_gpu"] == 2 and d["value"] > 1.0


def test_data_parallel_code_path_on_one_gpu():
    """--force-collective: RCCL process group, bank all-gather on the side stream bracketed by HIP events, barriers and the rank
    reductions of the N > 1 path, with one rank (the driver's 8-GPU run is the first time more than one exists)."""
    d = run([sys.executable, "bench.py", "--force-collective", "--workload", "c4", "--steps", "3", "--warmup", "1", "--clips", "1",
             "--no-cpu-baseline", "--no-roofline", "--no-extras"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["config"]["gallery"] == 50000 and d["value"] > 1.0
    a = d["allgather"]
    assert a["median_us_slowest_rank"] > 0 and a["bytes_sent_per_peer"] == 50000 * 256 * 4 and a["bound_us_ring"] == 0.0


def test_self_launch_branch_on_the_gpu():
    """`python bench.py --gpus N` starts its own ranks (child torch.distributed.run, before any HIP call in the parent) and relays
    rank 0's line; with one GPU the same branch is taken through SEAM_BENCH_SELF_LAUNCH=1.  --force-collective makes the single
    rank run the N > 1 step (RCCL group, side-stream all-gather, rank reductions, roofline leg on the collective path)."""
    d = run([sys.executable, "bench.py", "--gpus", "1", "--force-collective", "--workload", "c4", "--steps", "2", "--warmup", "1",
             "--clips", "1", "--no-cpu-baseline", "--no-extras"], SEAM_BENCH_SELF_LAUNCH="1")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["data"] == "synthetic"
    assert d["bank_identical_on_all_ranks"] is True and len(d["ms_per_step_per_rank"]) == 1
    assert d["allgather"]["median_us_slowest_rank"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0.3 < d["roofline"]["frac"] < 1.0        # the N > 1 line carries the roofline too


def test_more_gpus_than_the_box_has_is_a_clean_error():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--steps", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=300, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    assert r.returncode == 2 and "only" in r.stderr and r.stdout.strip() == ""


def test_bank_rows_do_not_depend_on_the_shard_split():
    """bench.py's shop-side bank (round 6: every rank runs the shop images of ITS rows through the extractor): a row is the same bits
    whichever rank computes it -- rows [0, 300) in one piece == [0, 100) + [100, 300) (shard borders that cut through the 128-image
    batches) == eight ragged shards; and the rows are the aggregator's shop descriptors of the synthetic shop images."""
    import torch
    import bench
    from seam_match_rcnn_amd import retrieval
    dev = torch.device("cuda:0")
    model, _ = bench.build_model(dev)
    model.roi_heads.roi_features_contiguous = False
    ta = model.roi_heads.temporal_aggregator
    whole = bench.compute_bank_rows(model, ta, 0, 300, dev)
    assert tuple(whole.shape) == (300, 256) and bool(torch.isfinite(whole).all())
    two = torch.cat([bench.compute_bank_rows(model, ta, 0, 100, dev), bench.compute_bank_rows(model, ta, 100, 300, dev)])
    assert torch.equal(two, whole)
    parts = [bench.compute_bank_rows(model, ta, *retrieval.shard_range(300, r, 8), dev) for r in range(8)]
    assert torch.equal(torch.cat(parts), whole)
    assert (model.transform.min_size, model.transform.max_size) == (800, 1333)            # the shop pass restores the transform
    # products differ: no two of the first rows are (nearly) the same descriptor
    d = torch.cdist(whole[:64], whole[:64]) + torch.eye(64, device=dev) * 1e9
    assert float(d.min()) > 1e-3 * float(whole.abs().max())
