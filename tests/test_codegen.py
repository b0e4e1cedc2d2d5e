"""The generated gfx950 code of libggq has the properties the design relies on (tests/isa_checks.py says which and
why).  CPU only: hipcc cross-compiles the device side to assembly, no GPU is touched.  The last two tests remove one
compiler flag each from the Makefile's set and expect the same checks to complain, so that a toolchain or Makefile
change that loses a flag's effect cannot pass unnoticed."""
import os
import shutil

import pytest

import isa_checks as c

pytestmark = pytest.mark.skipif(shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="needs make and hipcc")


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    s = str(tmp_path_factory.mktemp("isa") / "ggq_api.s")
    remarks = c.compile_device(s)
    funcs = c.functions(open(s).read())
    res = c.resources(remarks)
    names = c.demangle(sorted(set(funcs) | set(res)))
    return funcs, res, names


def test_flags_present():
    flags = c.makefile_flags()
    assert "-ffp-contract=off" in flags
    assert "-Xclang -target-feature -Xclang -packed-fp32-ops" in flags
    assert "--offload-arch=gfx950" in flags


def test_no_packed_f32_no_mixed_fma_and_fma_only_in_divisions(shipped):
    funcs, _, names = shipped
    assert len(funcs) > 100
    assert c.check_arithmetic(funcs, names) == []
    # the tagged steps of the quantizers' short divisions were seen at all
    assert sum(1 for n in funcs if names[n].startswith("void ggq::quant_one<")) == 21


def test_hand_placed_instructions_keep_their_wait_states(shipped):
    funcs, _, _ = shipped
    assert c.check_hand_placed_hazards(funcs) == []
    # the check sees what it is meant to see
    broken = {"f": ["v_rcp_f32 v1, v2", "v_fma_f32 v3, -v2, v1, 1.0"],
              "g": ["v_max3_f32 v8, v8, |v7|, |v6|", "v_max_f32_dpp v8, v8, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"],
              "h": ["v_rcp_f32 v1, v2", "s_nop 0", "v_fma_f32 v3, -v2, v1, 1.0", "v_mov_b32_e32 v8, v3", "s_nop 1",
                    "v_max_f32_dpp v8, v8, v8 row_mirror row_mask:0xf bank_mask:0xf"]}
    broken["k"] = ["v_cmp_gt_f32_e64 s[6:7], |v12|, |v11|", "v_cndmask_b32 v11, v11, v12, s[6:7]"]
    broken["l"] = ["v_cmp_gt_f32_e64 s[6:7], |v12|, |v11|", "s_or_b64 s[6:7], s[8:9], s[6:7]", "s_nop 0",
                   "v_cndmask_b32 v11, v11, v12, s[6:7]"]
    broken["m"] = ["v_cndmask_b32 v11, v11, v12, s[4:5]", "s_nop 0", "v_permlane16_swap_b32_e32 v11, v12"]
    broken["n"] = ["v_and_b32 v1, 0, v1", "v_readfirstlane_b32 s4, v1"]
    broken["o"] = ["v_and_b32 v1, 0, v1", "s_nop 0", "v_readfirstlane_b32 s4, v1"]
    broken["p"] = ["v_dot4c_i32_i8_e32 v4, 0x1010101, v3", "s_nop 1", "v_add_u32_dpp v4, v4, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"]
    broken["q"] = ["v_dot4c_i32_i8_e32 v4, 0x1010101, v3", "s_nop 2", "v_add_u32_dpp v4, v4, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"]
    found = c.check_hand_placed_hazards(broken)
    assert [x[0] for x in found] == ["f", "g", "k", "m", "n", "p"]


def test_group_start_is_a_scalar_probe_ahead_of_the_tile_loads(shipped):
    funcs, _, names = shipped
    assert c.check_group_start(funcs, names) == []


def test_decode_kernels_resources(shipped):
    _, res, names = shipped
    assert c.check_resources(res, names) == []


@pytest.mark.parametrize("dropped, expect", [
    ("-ffp-contract=off", "v_fma_mix"),
    ("-Xclang -target-feature -Xclang -packed-fp32-ops", "v_pk_"),
])
def test_checks_notice_a_missing_flag(tmp_path, dropped, expect):
    flags = c.makefile_flags()
    assert dropped in flags
    s = str(tmp_path / "sabotaged.s")
    c.compile_device(s, flags_override=flags.replace(dropped, ""))
    complaints = c.check_arithmetic(c.functions(open(s).read()))
    assert complaints and any(expect in x for x in complaints)
