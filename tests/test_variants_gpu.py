"""The A/B kernel variants kept behind RS_GRAD_V=1 / RS_ROLLOUT_V=1 (64-sample-per-wave gradient kernel, 64-env-per-wave
rollout) must stay correct: the fused-path tests are re-run in a child process with the toggles set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_v1_kernel_variants_pass_the_fused_path_tests():
    env = dict(os.environ, RS_GRAD_V="1", RS_ROLLOUT_V="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ppo_gpu.py"), "-m", "gpu", "-x", "-q", "-k",
           "fused or device_side or reference_update_rada2c", "-p", "no:cacheprovider"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_v4_six_term_split_bf16_gradient_kernel_is_f32_accurate():
    """RS_GRAD_V=4 (three bf16 pieces per operand, six cross terms): as accurate as the f32 instruction, so it must pass
    every fused-path test including the Adam trajectory."""
    env = dict(os.environ, RS_GRAD_V="4")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ppo_gpu.py"), "-m", "gpu", "-x", "-q", "-k",
           "fused or device_side or reference_update_rada2c", "-p", "no:cacheprovider"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_v3_split_bf16_gradient_kernel_within_the_stated_tolerance():
    """RS_GRAD_V=3: the 64x64 GEMMs of the gradient pass on split-bf16 matrix instructions (hi*hi + hi*lo + lo*hi in
    float32, ~4e-6 relative error).  It must pass the same gradient / loss / reference-golden tests at the stated
    fp32 tolerance (rtol 1e-4); the 8-step Adam trajectory and the full-size linearity checks, whose absolute
    tolerances are tuned to the exact f32 path, are left to the default kernel."""
    env = dict(os.environ, RS_GRAD_V="3")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ppo_gpu.py"), "-m", "gpu", "-x", "-q", "-k",
           "fused_ppo_grad or fused_collector or reference_update_rada2c", "-p", "no:cacheprovider"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
