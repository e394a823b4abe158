"""CPU logic check of the per-element kernels: the SAME csrc/pf_elem.h device functions,
compiled for the host (tests/emu/pf_emu.cpp), run through the SAME ctypes wrappers and the
SAME parity cases as the GPU tests.  This validates index / wrap / zero-padding logic in the
GPU-less build container; it is not a product path and proves nothing about the HIP build
(tests/test_hip_kernels.py does, on the GPU)."""
import os
import shutil
import subprocess

import pytest
import torch

import golden_cases as gc
import kernel_cases as kc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libpf_emu.so")
CSRC = os.path.join(ROOT, "prior-flow_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    srcs = [os.path.join(EMU_DIR, "pf_emu.cpp"), os.path.join(CSRC, "pf_elem.h"),
            os.path.join(CSRC, "pf_api_elem.inc"), os.path.join(CSRC, "pf_common.h")]
    if not os.path.exists(EMU_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMU_SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
                               "-I", CSRC, srcs[0], "-o", EMU_SO])
    from prior_flow_amd._lib import PfLib
    return PfLib(EMU_SO, require_cuda=False, optional=("pf_debug_dirty_lds", "pf_conv2d", "pf_conv2d_tile", "pf_conv2d_stats_blocks", "pf_conv2d_roles", "pf_corr_pyramid", "pf_corr_pyramid_bf16x3", "pf_conv2d_wgrad",
                                                            "pf_dccl_combine_conv1x1", "pf_conv2d_wgrad_small", "pf_conv2d_wgrad_small_ws",
                                                            "pf_conv2d_wgrad_small_ws_floats", "pf_enc_stem"))


@pytest.fixture(scope="module")
def params():
    from prior_flow_amd.modules import state_dict_shapes
    return gc.det_state_dict(state_dict_shapes())


@pytest.mark.parametrize("case", kc.ELEMENTWISE_CASES, ids=lambda c: c.__name__)
def test_emu_case(emu, case):
    case(emu, torch.device("cpu"))


def test_emu_direct_conv(emu, params):
    kc.case_direct_conv(emu, torch.device("cpu"), params)


def test_pymod_matches_aten_remainder_bitwise(tmp_path):
    """pf_pymod replaces fmodf by a floor/FMA step + correction: it must reproduce ATen's `%` bit for bit
    (exact multiples incl. the signed zero, neighbours of multiples, tiny negatives that round to W)."""
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "pf_pymod_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "prior-flow_amd", "csrc"),
                           os.path.join(ROOT, "tests", "emu", "pf_pymod_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-500:]
