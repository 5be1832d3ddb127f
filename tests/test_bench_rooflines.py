"""bench.py's roofline arithmetic on the build container (no GPU): the two-class issue-time floors are computed from the committed PMC
passes (a property of the code) and whatever issue rates the run measured; the batch generator is position-independent; the shipped
kernel's instruction counts are there after a build."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ROOFS = {"mad": {"T_lane_ops_per_s": 30.0}, "valu": {"T_lane_ops_per_s": 53.0}}


def test_derive_scalars_is_position_independent():
    a = bench.derive_scalars(b"tag", 1000, 64, bench.R_BANDERSNATCH)
    assert len(a) == 64 * 32 and bench.derive_scalars(b"tag", 1010, 5, bench.R_BANDERSNATCH) == a[320:480]
    assert bench.derive_scalars(b"other", 1000, 64, bench.R_BANDERSNATCH) != a
    vals = [int.from_bytes(a[32 * i: 32 * i + 32], "little") for i in range(64)]
    assert all(0 < v < 1 << 248 for v in vals) and len(set(vals)) == 64


def test_issue_roofline_from_committed_counts():
    r = bench.issue_roofline("k_ped_verify<SuiteBandersnatch>", ("r6_pmc_per_item.json",), 65536, 25e6, ROOFS, "unit test")
    assert r["int64_class_instructions_per_item"] > 3e5 and r["valu_instructions_per_item"] > r["int64_class_instructions_per_item"]
    floor = r["int64_class_instructions_per_item"] / 30e12 + (r["valu_instructions_per_item"] - r["int64_class_instructions_per_item"]) / 53e12
    assert abs(r["issue_time_floor_us_per_item"] - floor * 1e6) < 1e-9 and abs(r["frac"] - floor * 25e6) < 1e-9
    assert "error" in bench.issue_roofline("k_no_such_kernel", ("r6_pmc_per_item.json",), 65536, 1e6, ROOFS, "x")
    assert "error" in bench.issue_roofline("k_ped_verify<SuiteBandersnatch>", ("r6_pmc_per_item.json",), 65536, 1e6, None, "x")


def test_step_issue_floor_sums_the_kernels_of_one_batch():
    sf = bench.step_issue_floor(("r6_pmc_thin.json",), ROOFS)
    assert sf and any("k_accumulate" in k for k in sf["kernels"]) and any("k_thin_prepare" in k for k in sf["kernels"])
    assert 0.2 < sf["issue_time_floor_ms"] < 0.6 and sf["int64_class_instructions_per_batch"] < sf["valu_instructions_per_batch"]
    d = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_thin.json")))
    assert sf["counts_source"].startswith("profiles/r6_pmc_thin.json") and str(d.get("head"))[:12] in sf["counts_source"]


def test_shipped_kernel_counts_after_build():
    c = bench.shipped_kernel_counts(("k_accumulate", "TeCurve", "17SuiteBandersnatchE"))
    if not os.path.exists(os.path.join(ROOT, "ark_vrf_amd", "kernel_counts.json")):
        assert c is None or "error" in c
        return
    lp = c["largest_loop"]
    assert lp["multiply_add"] >= 8 * 153 and lp["vector_alu"] > lp["multiply_add"] and lp["instructions"] >= lp["vector_alu"]
