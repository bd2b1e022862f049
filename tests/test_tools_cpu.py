"""CPU: the small evidence tools that post-process rocprofv3 output (no GPU, synthetic traces)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
                    "Start_Timestamp", "End_Timestamp"])
        for i, (name, start, dur) in enumerate(rows):
            w.writerow(["KERNEL_DISPATCH", "Agent 2", 1, 1, 1, i + 1, 1, name, i + 1, start, start + dur])


def test_memset_split_separates_setup_fills_from_keyframe_fills(tmp_path):
    fill, hist, fold = "__amd_rocclr_fillBufferAligned", "void k_shadow_hist<0>(float const*, int)", "void k_semb_fold_tasks<HvSemVoxel, float, 2>(HvTable)"
    rows, t = [], 1000
    rows += [(fill, t, 5_000_000), (fill, t + 6_000_000, 1_000_000)]  # pool zeroing at volume creation
    t += 8_000_000
    for _ in range(3):  # three keyframes: hist ... one fill inside ... fold, then the fill that follows the fold
        rows.append((hist, t, 20_000))
        rows.append(("k_shadow_mask(float const*)", t + 21_000, 7_000))
        rows.append((fill, t + 30_000, 4_000))
        rows.append(("void k_sem_assoc_vote<HvSemVoxel, false>(HvTable)", t + 35_000, 100_000))
        rows.append((fold, t + 140_000, 4_000))
        rows.append((fill, t + 150_000, 3_000))
        t += 200_000
    d = tmp_path / "kt"
    d.mkdir()
    _trace(d / "sem_kernel_trace.csv", rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "memset_split.py"), str(d)], capture_output=True, text=True, check=True).stdout
    r = json.loads(out)
    assert r["keyframes"] == 3
    assert r["fills_inside_keyframes"] == 3 and r["fills_inside_per_keyframe"] == 1.0
    assert r["fills_inside_mean_us"] == 4.0
    assert r["fills_directly_after_a_keyframe"] == 3
    assert r["fills_outside_keyframes"] == 2 + 3  # set-up + the one after every keyframe
    assert r["fills_outside_max_us"] == 5000.0
    assert r["launches_per_keyframe_incl_fills"] == 5.0


def test_pmc_summary_ranges_and_second_entries(tmp_path):
    """tools/pmc_summary.py: --range keeps a kernel's launches LO .. HI-1 (bench.py's incremental extraction ticks), --also adds a second
    entry "<kernel><suffix>" over another range (its forced full pass); bench.pmc_kernel tells the two apart."""
    d = tmp_path / "pmc_FETCH_SIZE"
    d.mkdir()
    with open(d / "pmc_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        values = [1000.0, 10.0, 20.0, 30.0, 1200.0]  # full (cold), three ticks, forced full
        for i, v in enumerate(values):
            w.writerow([i * 2 + 1, "k_unit_masks(HvTable, char const*)", "FETCH_SIZE", v])
            w.writerow([i * 2 + 2, "k_mc_vertices(HvTable)", "FETCH_SIZE", 500.0])
    out = tmp_path / "pmc.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--json", str(out), "--command-key", "K", "--also", "k_unit_masks:4:5:@full",
                    "--range", "k_unit_masks:1:4", str(d)], capture_output=True, text=True, check=True)
    z = json.load(open(out))
    assert z["kernels"]["k_unit_masks"]["FETCH_SIZE"] == 20.0 and z["kernels"]["k_unit_masks"]["launches"] == 3
    assert z["kernels"]["k_unit_masks@full"]["FETCH_SIZE"] == 1200.0 and z["kernels"]["k_unit_masks@full"]["launches"] == 1
    assert z["kernels"]["k_mc_vertices"]["launches"] == 5
    sys.path.insert(0, ROOT)
    import bench

    assert bench.pmc_kernel(z, "k_unit_masks", "K")["FETCH_SIZE"] == 20.0
    assert bench.pmc_kernel(z, "k_unit_masks@full", "K")["FETCH_SIZE"] == 1200.0
    assert bench.pmc_kernel(z, "k_mc_vertices@full", "K") is None
