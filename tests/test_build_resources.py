"""The compiler's per-kernel resource report of the built library (de6d_amd/_build.py writes libdet6d_hip.usage.json from
-Rpass-analysis=kernel-resource-usage).  The latency-chain samplers and the GEMM family must not touch scratch memory: in
round 4 a computed array index moved the 16384-point sampler's coordinates to scratch (144 bytes per lane, re-read every
sweep) without a warning — 0.78 instead of 0.57 us per pick."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USAGE = os.path.join(ROOT, "de6d_amd", "csrc", "libdet6d_hip.usage.json")

#: substrings of the mangled names of kernels that must hold everything in registers / LDS
NO_SCRATCH = ("fps_seq_kernel", "fps_coop_multi_kernel", "fps_fat_kernel", "cell_sort_kernel", "mlp_group_kernel", "mlp_group_stream_kernel",
              "mlp_rows_kernel", "mlp_chain_reg_kernel", "mlp_chain_wide_kernel", "ball_query_pair_kernel", "bq_grid_query_kernel",
              "compact_place_kernel", "post_select_kernel", "post_mask_kernel")


@pytest.fixture(scope="module")
def usage():
    if not os.path.exists(USAGE):
        pytest.skip("library not built by de6d_amd._build in this tree")
    with open(USAGE) as f:
        per_file = json.load(f)
    flat = {}
    for kernels in per_file.values():
        flat.update(kernels)
    return flat


def test_report_covers_the_hot_kernels(usage):
    for key in NO_SCRATCH:
        assert any(key in name for name in usage), key


def test_hot_kernels_hold_no_scratch_and_spill_nothing(usage):
    bad = {name: u for name, u in usage.items() if any(k in name for k in NO_SCRATCH)
           and (u.get("ScratchSize", 0) or u.get("VGPRs Spill", 0) or u.get("Dynamic Stack") == "True")}     # (SGPR spills go to VGPR lanes)
    assert not bad, bad


def test_samplers_fit_a_full_workgroup(usage):
    """1024-thread workgroups: 16 waves on 4 SIMDs leave 128 registers per lane"""
    for name, u in usage.items():
        if "fps_seq_kernel" in name or "fps_coop_multi_kernel" in name:
            assert u["VGPRs"] + u.get("AGPRs", 0) <= 128 and u["Occupancy"] >= 4, (name, u)


def test_resource_report_parser():
    """de6d_amd/_build.py: the remarks of -Rpass-analysis=kernel-resource-usage -> per-kernel fields; everything else passes through"""
    from de6d_amd._build import _parse_usage
    text = (
        "/x/a.hip:10:1: remark: Function Name: _Z3foov [-Rpass-analysis=kernel-resource-usage]\n"
        "   10 | __global__ void foo() {\n"
        "      | ^\n"
        "/x/a.hip:10:1: remark:     TotalSGPRs: 20 [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:10:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:10:1: remark:     ScratchSize [bytes/lane]: 144 [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:10:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:10:1: remark:     LDS Size [bytes/block]: 34888 [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:22:3: warning: something else\n"
        "/x/a.hip:30:1: remark: Function Name: _Z3barv [-Rpass-analysis=kernel-resource-usage]\n"
        "/x/a.hip:30:1: remark:     VGPRs Spill: 3 [-Rpass-analysis=kernel-resource-usage]\n")
    usage, rest = _parse_usage(text)
    assert usage == {"_Z3foov": {"TotalSGPRs": 20, "VGPRs": 128, "ScratchSize": 144, "Dynamic Stack": "False", "LDS Size": 34888},
                     "_Z3barv": {"VGPRs Spill": 3}}
    assert rest == "/x/a.hip:22:3: warning: something else\n"
