"""bench.py's host-side helpers (CPU only): the algorithmic-work table, the synthetic workloads, the attribution of PMC rows
to C-ABI launches by marker kernels (profiles/pmc_to_traffic.py) and the staleness check of profiles/traffic.json."""
import csv
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_formulas_match_survey_8d():
    # ball query 12 B (N+M) + 4 B M (ns+1); group_point 4 B (N C + M ns + M ns C); three_nn 12 B (n+m) + 24 B n
    assert bench.algorithmic("pasnl_query_ball_point", (64, 1024, 512, 32))[0] == 64 * (12 * (1024 + 512) + 4 * 512 * 33)
    assert bench.algorithmic("pasnl_group_point", (64, 512, 128, 128, 64))[0] == 4 * 64 * (512 * 128 + 128 * 64 + 128 * 64 * 128)
    by, fl, bound = bench.algorithmic("pasnl_three_nn", (16, 8192, 1024))
    assert by == 16 * (12 * (8192 + 1024) + 24 * 8192) and fl == 8 * 16 * 8192 * 1024 and bound == "valu"
    assert bench.algorithmic("pasnl_knn_batch", (64, 1024, 1024, 32, 0))[2] == "valu"
    assert bench.algorithmic("pasnl_farthest_point_sample_gather", (64, 1024, 512))[2] == "latency"
    for sym in ("pasnl_as_cell_narrow", "pasnl_as_cell_wide_ld", "pasnl_sa_cell_centre0", "pasnl_nl_attention"):
        dims = {"pasnl_as_cell_narrow": (32768, 12, 32, 9, 6), "pasnl_as_cell_wide_ld": (8192, 12, 65, 134, 131, 224),
                "pasnl_sa_cell_centre0": (64, 512, 128, 128, 64, 128, 128), "pasnl_nl_attention": (64, 512, 1024, 32, 0)}[sym]
        by, fl, bound = bench.algorithmic(sym, dims)
        assert by > 0 and fl > 0 and bound == "mfma"


def test_workloads_are_seeded_and_shaped_like_the_baseline_configs():
    shapes = {1: (64, 1024, 3), 2: (64, 1024, 3), 3: (16, 8192, 6), 4: (8, 10240, 3)}
    for ci, spec in bench.WORKLOADS.items():
        small = dict(spec, batch=2, points=256)
        a, b = bench.make_input(ci, small, 0), bench.make_input(ci, small, 0)
        assert a.dtype == np.float32 and a.shape == (2, 256, shapes[ci][2]) and np.array_equal(a, b)
        assert not np.array_equal(a, bench.make_input(ci, small, 1)), "ranks draw different clouds"
        assert (spec["batch"], spec["points"]) == shapes[ci][:2]
    noisy = bench.make_input(2, dict(bench.WORKLOADS[2], batch=2, points=256), 0)
    clean = bench.synth_clouds(1234 + 2, 2, 256)
    assert not np.array_equal(noisy[:, :10], clean[:, :10]) and np.array_equal(noisy[:, 10:], clean[:, 10:])


def test_pmc_rows_are_attributed_to_launches_by_markers(tmp_path):
    """Two launches: the first starts two pasnl kernels and a vendor GEMM, the second one kernel; an unmeasured launch in
    front.  The script must sum per launch, skip what is not pasnl::, skip "_unmeasured", double FETCH_SIZE, and take KiB."""
    seq = [["_unmeasured", [1]], ["pasnl_knn_batch_ws", [2, 3]], ["pasnl_group_point", [4]]]
    (tmp_path / "pmc_FETCH_SIZE.json").write_text("noise\n" + json.dumps({"launch_sequence": seq}) + "\n")
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    trace = ["at::cuda::(anonymous namespace)::spin_kernel(long)", "void pasnl::warm_kernel(int)",
             "at::cuda::(anonymous namespace)::spin_kernel(long)", "void pasnl::knn_grid_build_kernel(int)", "Cijk_vendor_gemm",
             "void pasnl::knn_grid_query_kernel<1, true, int>(int)",
             "at::cuda::(anonymous namespace)::spin_kernel(long)", "void pasnl::group_point_kernel<4>(int)"]
    values = {"FETCH_SIZE": [0, 9, 0, 10, 99, 30, 0, 5], "WRITE_SIZE": [0, 9, 0, 1, 99, 2, 0, 7]}
    for counter, vals in values.items():
        d = tmp_path / f"pmc_{counter}"
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=cols)
            w.writeheader()
            for i, (name, v) in enumerate(zip(trace, vals)):
                w.writerow({"Dispatch_Id": i + 1, "Kernel_Name": name, "Counter_Name": counter, "Counter_Value": v})
    out = tmp_path / "traffic.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "pmc_to_traffic.py"), str(tmp_path), str(out)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    t = json.load(open(out))
    assert t["pasnl_knn_batch_ws:2,3"] == int((2 * 40 + 3) * 1024) and t["pasnl_group_point:4"] == int((2 * 5 + 7) * 1024)
    assert not any(k.startswith("_unmeasured") for k in t) and set(t["_source"]) == {"commit", "csrc_sha256"}


def test_stale_traffic_is_not_reported(tmp_path, monkeypatch):
    """measured_traffic refuses a figure whose kernel source changed since the PMC pass."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    digests = bench.csrc_digests()
    key = "pasnl_query_ball_point:64,1024,512,32"
    json.dump({key: 123, "_source": {"commit": "abc", "csrc_sha256": digests}}, open(prof / "traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.symlink(os.path.join(ROOT, "pointasnl_amd"), tmp_path / "pointasnl_amd")
    assert bench.measured_traffic("pasnl_query_ball_point", [64, 1024, 512, 32])[0] == 123
    stale = dict(digests, **{"grouping.hip": "0" * 64})
    json.dump({key: 123, "_source": {"commit": "abc", "csrc_sha256": stale}}, open(prof / "traffic.json", "w"))
    val, why = bench.measured_traffic("pasnl_query_ball_point", [64, 1024, 512, 32])
    assert val is None and "stale" in why and "grouping.hip" in why


def test_asm_scan_flags_inline_assembly_next_to_matrix_instructions():
    """tools/asm_scan.py: a vector instruction inside ;;#ASMSTART .. ;;#ASMEND that shares a register with a
    v_mfma close by is reported (the hazard recogniser cannot see it); the same instruction far enough away, or outside inline
    assembly, is not."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("asm_scan", os.path.join(ROOT, "tools", "asm_scan.py"))
    asm_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(asm_scan)
    bad = """_Zk:
	v_mfma_f32_32x32x2_f32 a[0:15], v0, v1, a[0:15]
	;;#ASMSTART
	v_max_f32 v0, v0, v2
	;;#ASMEND
	s_nop 7
	s_nop 7
	s_nop 7
	;;#ASMSTART
	v_max_f32 v5, v6, v7
	;;#ASMEND
	v_mfma_f32_32x32x2_f32 a[0:15], v5, v1, a[0:15]
	v_mfma_f32_16x16x4_f32 v[8:11], v5, v1, v[8:11]
	v_add_f32 v3, v3, v3
	;;#ASMSTART
	s_nop 1
	v_max_f32_dpp v9, v9, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
	;;#ASMEND
""".split("\n")
    found = asm_scan.asm_mfma_hazards(bad, 0, len(bad))
    assert len(found) == 3 and "v0" in found[0] and "v5" in found[1] and "v9" in found[2]
    good = """_Zk:
	v_mfma_f32_32x32x2_f32 a[0:15], v0, v1, a[0:15]
	v_max_f32 v0, v0, v2
	;;#ASMSTART
	v_max_f32 v20, v21, v22
	;;#ASMEND
	s_nop 7
	s_nop 7
	s_nop 7
	;;#ASMSTART
	v_max_f32 v1, v1, v1
	;;#ASMEND
""".split("\n")
    assert asm_scan.asm_mfma_hazards(good, 0, len(good)) == []
    assert asm_scan.regs_of("v[4:7]") == {"v4", "v5", "v6", "v7"} and asm_scan.regs_of("a3") == {"a3"} and asm_scan.regs_of("s[0:1]") == set()


def test_asm_scan_checks_wait_states_in_front_of_inline_dpp():
    """tools/asm_scan.py `asm_dpp`: a DPP instruction in inline assembly needs two wait states behind the vector
    instruction that wrote the register it reads from other lanes (ball_grid.hip's v_min_i32_dpp reductions carry s_nop 1)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("asm_scan", os.path.join(ROOT, "tools", "asm_scan.py"))
    asm_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(asm_scan)
    bad = """_Zk:
	v_xor_b32_e32 v4, v5, v6
	;;#ASMSTART
	v_min_i32_dpp v4, v4, v4 row_shr:1 row_mask:0xf bank_mask:0xf
	s_nop 0
	v_min_i32_dpp v4, v4, v4 row_shr:2 row_mask:0xf bank_mask:0xf
	;;#ASMEND
""".split("\n")
    found = asm_scan.asm_mfma_hazards(bad, 0, len(bad))
    assert len(found) == 2 and "0 wait" in found[0] and "1 wait" in found[1]
    good = """_Zk:
	v_xor_b32_e32 v4, v5, v6
	;;#ASMSTART
	s_nop 1
	v_min_i32_dpp v4, v4, v4 row_shr:1 row_mask:0xf bank_mask:0xf
	s_nop 1
	v_min_i32_dpp v4, v4, v4 row_shr:2 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	v_add_u32_e32 v7, v8, v9
	v_add_u32_e32 v7, v8, v9
	;;#ASMSTART
	v_min_i32_dpp v4, v4, v4 row_shr:4 row_mask:0xf bank_mask:0xf
	;;#ASMEND
""".split("\n")
    assert asm_scan.asm_mfma_hazards(good, 0, len(good)) == []
