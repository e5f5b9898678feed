"""The occupancy figures DESIGN.md and the launch logic (hipnlp.hip: `wide`, `fused`) assume, checked on the BUILT library's own code
objects (tools/kernel_resources.py): a source change that makes the compiler unroll one loop can cost 70 VGPRs and a workgroup per
CU without failing any numerical test (seen once: 8.3 instead of 7.9 us per 100-knot launch, x 64 30 % slower).  CPU only."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

LIB = os.path.join(ROOT, "hippopt_amd", "lib", "libhipnlp.so")
LDS_PER_CU = 160 * 1024
VGPR_FILE = 512          # per SIMD lane (unified VGPR + AGPR file of gfx950)


def _granule(v):         # the allocation granularity of the register file is 8
    return (v + 7) // 8 * 8


@pytest.fixture(scope="module")
def res():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    if not os.path.exists(os.path.join(kernel_resources.LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf")
    return kernel_resources.kernel_resources(LIB)


def test_every_kernel_is_there_and_none_needs_scratch(res):
    expect = {"hipnlp_knot_kernel<0,4>", "hipnlp_knot_kernel<1,4>", "hipnlp_knot_kernel<0,8>", "hipnlp_knot_kernel<1,8>", "hipnlp_knot_hess_kernel<0,0>", "hipnlp_knot_hess_kernel<0,1>",
              "hipnlp_knot_hess_kernel<1,0>", "hipnlp_knot_hess_kernel<1,1>", "hipnlp_pose_kernel<0>", "hipnlp_pose_kernel<1>", "hipnlp_pose_hess_kernel<0>", "hipnlp_pose_hess_kernel<1>",
              "hipnlp_reduce_kernel", "hipnlp_reassemble_kernel", "hipnlp_peer_push_kernel", "hipnlp_peer_signal_kernel", "hipnlp_peer_wait_kernel"}
    assert expect <= set(res), sorted(expect - set(res))
    for name, r in res.items():
        assert r["scratch"] == 0, "%s spills %d B per lane to scratch" % (name, r["scratch"])


@pytest.mark.parametrize("peers", ["", ",true", ",false,true"])     # plain, peer exchange, VARY (the constants of jac g are not staged or stored)
@pytest.mark.parametrize("terrain", [0, 1])
def test_four_wave_callback_kernel_fits_four_workgroups_per_cu(res, terrain, peers):
    r = res["hipnlp_knot_kernel<%d,4%s>" % (terrain, peers)]
    assert r["wg"] == 256
    assert 4 * r["lds"] <= LDS_PER_CU, r
    assert _granule(r["vgpr"] + r["agpr"]) * 4 <= VGPR_FILE, r      # four waves per SIMD


@pytest.mark.parametrize("peers", ["", ",true", ",false,true"])
@pytest.mark.parametrize("terrain", [0, 1])
def test_eight_wave_callback_kernel_fits_two_workgroups_per_cu(res, terrain, peers):
    """hipnlp.hip picks it for (knots + 1) x batch <= 512 = 256 CUs x 2"""
    r = res["hipnlp_knot_kernel<%d,8%s>" % (terrain, peers)]
    assert r["wg"] == 512
    assert 2 * r["lds"] <= LDS_PER_CU, r
    assert _granule(r["vgpr"] + r["agpr"]) * 4 <= VGPR_FILE, r      # 2 workgroups x 8 waves over 4 SIMDs


def test_hessian_and_pose_kernels(res):
    # exact Hessian of the kinodynamic NLP: full layout two workgroups per CU, compact layout three (both terrains; the smooth terrain's
    # compact kernel sits at the 168 cap without scratch since its point tasks were split and the copy-out tables are fetched late)
    for t in (0, 1):
        h = res["hipnlp_knot_hess_kernel<%d,0>" % t]
        assert 2 * h["lds"] <= LDS_PER_CU and _granule(h["vgpr"] + h["agpr"]) * 2 <= VGPR_FILE and h["scratch"] == 0, h
        hc = res["hipnlp_knot_hess_kernel<%d,1>" % t]
        # (LDS is handed out in granules of 1 280 B: three workgroups need 3 x ceil(lds / 1280) granules of the CU's 128)
        assert 3 * (-(-hc["lds"] // 1280)) * 1280 <= LDS_PER_CU and _granule(hc["vgpr"] + hc["agpr"]) * 3 <= VGPR_FILE and hc["scratch"] == 0, hc
    # the planar Hessian kernel that stores its entries straight into a device destination (no LDS staging): four workgroups per CU
    hd = res["hipnlp_knot_hess_kernel<0,1,true>"]
    assert 4 * (-(-hd["lds"] // 1280)) * 1280 <= LDS_PER_CU and _granule(hd["vgpr"] + hd["agpr"]) * 4 <= VGPR_FILE and hd["scratch"] == 0, hd
    # pose kernels on the compact (static) scratch with the lite tables: planar terrain five per CU, smooth steps four (callbacks) / three (Hessian)
    for t in (0, 1):
        p = res["hipnlp_pose_kernel<%d>" % t]
        # round 6: the planar callback kernel at FIVE per CU (static scratch without velocity arrays, Jacobian staging cut to the pieces the pose
        # program touches, static kinematics: 31.8 KB, 90 VGPRs); the smooth terrain's stays at four (123 VGPRs)
        pcu = 5 if t == 0 else 4
        assert pcu * (-(-p["lds"] // 1280)) * 1280 <= LDS_PER_CU and _granule(p["vgpr"] + p["agpr"]) * pcu <= VGPR_FILE and p["scratch"] == 0, p
        ph = res["hipnlp_pose_hess_kernel<%d>" % t]
        # round 6: the planar Hessian kernel at FIVE per CU (no staging of g / grad f in its scratch, the (q_b, q_b) block in three small groups)
        per_cu = 5 if t == 0 else 3
        assert per_cu * (-(-ph["lds"] // 1280)) * 1280 <= LDS_PER_CU and _granule(ph["vgpr"] + ph["agpr"]) * per_cu <= VGPR_FILE and ph["scratch"] == 0, ph


@pytest.mark.parametrize("terrain", [0, 1])
def test_four_wave_vary_kernels_fit_five_workgroups_per_cu(res, terrain):
    """the four-wave VARY kernels (destinations that hold the constant entries of jac g): trimmed scratch (only the Jacobian slots that
    may depend on x, the other horizon end's periodicity variables read from global memory, one pad word per record) -> 30.9 KB of LDS
    on the planar terrain, exactly 32 KB on the smooth steps; copy-out tables fetched behind the last barrier but one -> 96 VGPRs,
    no scratch: FIVE workgroups per CU"""
    r = res["hipnlp_knot_kernel<%d,4,false,true>" % terrain]
    assert r["wg"] == 256 and r["scratch"] == 0
    assert 5 * r["lds"] <= LDS_PER_CU, r
    assert _granule(r["vgpr"] + r["agpr"]) * 5 <= VGPR_FILE, r
