"""The sanitizer / bounds-checking builds of the library (`make asan`, `make debug`), when they are built:
* librevrand_hip_asan.so  -- host side of the C ABI under AddressSanitizer: the ABI checks (CPU) and the ragged-shape
  parity tests (GPU) run against it in a subprocess with the ASan runtime preloaded;
* librevrand_hip_debug.so -- -DRR_BOUNDS: guard bands around every device allocation + index assertions in the feature,
  SYRK, feature-matrix and FastFood kernels; the ragged-shape tests run against it on the GPU, and a deliberate overrun
  shows that the guards catch one."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

LIBDIR = os.path.join(ROOT, "revrand_amd", "lib")
ASAN_LIB = os.path.join(LIBDIR, "librevrand_hip_asan.so")
DEBUG_LIB = os.path.join(LIBDIR, "librevrand_hip_debug.so")
RAGGED = ["tests/test_gpu_rff.py::test_tiny_and_ragged_shapes_end_to_end", "tests/test_gpu_rff.py::test_transform_shapes_vs_oracle",
          "tests/test_gpu_rff.py::test_gram_vs_oracle", "tests/test_gpu_rff.py::test_gram_f64_vs_oracle",
          "tests/test_gpu_fastfood.py", "tests/test_gpu_slm.py::test_concat_second_pass_and_predict_vs_oracle",
          "tests/test_gpu_slm.py::test_second_pass_and_predict_vs_oracle", "tests/test_gpu_large_xdim.py",
          "tests/test_gpu_parity_r2.py::test_gram_with_a_ragged_last_column_block",
          "tests/test_gpu_parity_r2.py::test_predictive_variance_is_a_sum_of_squares_for_badly_scaled_covariances",
          # round 3: the products fused with their consumers (partial row tiles, Xdim below / above 32)
          "tests/test_gpu_slm.py::test_second_pass_product_fused_with_its_contraction_equals_the_two_pass_route",
          "tests/test_gpu_glm.py::test_edphi_product_fused_with_its_contraction_equals_the_two_pass_route",
          "tests/test_gpu_glm.py::test_first_product_with_the_likelihood_terms_as_its_epilogue_equals_the_three_pass_route",
          # round 4: the posterior's panel pipeline at config 3's width (65 panels, ragged last one) and its failure path
          "tests/test_gpu_posterior.py::test_posterior_vs_oracle_solve_posdef[8257-default]",
          "tests/test_gpu_posterior.py::test_not_positive_definite_in_a_late_panel_is_reported_and_leaves_nothing_in_flight[8257-8256-default]",
          # round 4: `predict` from the feature kernel alone (no feature-major output), ragged row counts; the paired
          # triangular product with the diagonal blocks' zero quarters skipped runs under the two predict tests above
          "tests/test_gpu_slm.py::test_predict_of_a_random_kernel_basis_comes_from_the_feature_kernel_alone",
          # round 5: the resident SVI loop (per-child tables, two feature matrices, the second stream), ragged minibatches
          "tests/test_gpu_resident_sgd.py::test_concatenation_of_fourier_and_linear_children",
          "tests/test_gpu_resident_sgd.py::test_resident_loop_equals_host_loop",
          # round 6: the in-process device group (every launch checked for "current device == the stream's device"), on
          # distinct GPUs where the box has them; the reference-pinned GLM fits through both loops
          "tests/test_gpu_multigpu.py::test_sharded_elbo_equals_the_one_context_elbo",
          "tests/test_gpu_multigpu_devices.py::test_one_member_group_under_rccl_runs_the_grouped_calls",
          "tests/test_gpu_multigpu_devices.py::test_elbo_with_devices_for_every_fit_state",
          "tests/test_gpu_glm_fit.py::test_fit_equals_the_references_fit[gaussian_cat_bs10_ns5-fused loop]",
          "tests/test_gpu_glm_fit.py::test_fit_equals_the_references_fit[binomial_cat_bs10_ns3-fused loop]",
          "tests/test_gpu_glm_fit.py::test_fit_equals_the_references_fit[poisson_ard_bs64f_ns5-resident loop]",
          "tests/test_gpu_fused_svi.py::test_shapes_across_the_tiles_of_the_matrix_core_products",
          "tests/test_gpu_resident_group.py::test_a_member_without_rows_of_a_minibatch_follows_the_others[two streams]",
          "tests/test_gpu_resident_group.py::test_group_resident_fit_equals_the_one_context_fit[two streams-gaussian-cat-devices2-host]",
          "tests/test_gpu_resident_group.py::test_group_resident_fit_equals_the_one_context_fit[one stream-binomial-iso-devices1-device]",
          "tests/test_gpu_resident_group.py::test_group_resident_fit_equals_the_one_context_fit[two streams-poisson-gm-devices4-host]",
          "tests/test_gpu_resident_sgd.py::test_spectral_mixture_children_run_resident[two streams-12-gaussian]"]


def _asan_runtime():
    for cc in ("/opt/rocm/bin/hipcc", "/opt/rocm/lib/llvm/bin/clang"):
        if os.path.exists(cc):
            p = subprocess.run([cc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
            if os.path.isabs(p) and os.path.exists(p):
                return p
    return None


def _pytest_with(lib, args, preload=None, timeout=1500):
    env = dict(os.environ, REVRAND_HIP_LIB=lib)
    if preload:
        env.update(LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=1")
        # The (uninstrumented) HIP runtime must tolerate a preloaded ASan runtime: the copy bundled with the torch wheel
        # does; /opt/rocm's aborts inside its own initialisation under the preload (ROCm ships separate ASan builds of its
        # libraries for that).  What is under test is the host side of librevrand_hip_asan.so, not the runtime.
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                env["RR_HIP_RUNTIME"] = "torch"
        except (ImportError, ValueError):
            pass
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_asan_build_passes_the_abi_checks():
    rt = _asan_runtime()
    if not os.path.exists(ASAN_LIB) or rt is None:
        pytest.skip("make -C revrand_amd/csrc asan has not been run")
    # the ABI checks, and the one host-only entry point with real work in it (rr_legacy_randn: worker threads, scratch)
    r = _pytest_with(ASAN_LIB, ["tests/test_abi.py", "tests/test_host_logic.py::test_library_generator_reproduces_numpy_legacy_randn",
                                "tests/test_host_logic.py::test_library_generator_reproduces_numpy_legacy_permutation",
                                "-m", "not gpu"], preload=rt, timeout=900)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_asan_build_runs_the_ragged_shape_tests():
    rt = _asan_runtime()
    if not os.path.exists(ASAN_LIB) or rt is None:
        pytest.skip("make -C revrand_amd/csrc asan has not been run")
    r = _pytest_with(ASAN_LIB, RAGGED[:3] + RAGGED[4:5] + ["-m", "gpu"], preload=rt)
    # every test passed, and no ASan report has a frame of this library in it.  (The HSA runtime bundled with torch
    # occasionally trips ASan inside libhsa-runtime64.so while the process exits, after the summary line: not ours.)
    import re
    assert re.search(r"\b\d+ passed\b", r.stdout) and not re.search(r"\b(failed|error)\b", r.stdout), \
        (r.stdout[-2500:], r.stderr[:3000], r.stderr[-1500:])
    reports = (r.stdout + r.stderr).split("ERROR: AddressSanitizer")[1:]
    assert not any("librevrand_hip" in rep for rep in reports), (r.stdout + r.stderr)[-4000:]


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_bounds_build_runs_the_ragged_shape_tests():
    if not os.path.exists(DEBUG_LIB):
        pytest.skip("make -C revrand_amd/csrc debug has not been run")
    r = _pytest_with(DEBUG_LIB, RAGGED + ["-m", "gpu"])
    assert r.returncode == 0 and "RR_BOUNDS" not in (r.stdout + r.stderr), (r.stdout[-2500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_bounds_build_catches_an_overrun():
    if not os.path.exists(DEBUG_LIB):
        pytest.skip("make -C revrand_amd/csrc debug has not been run")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from revrand_amd import _hip\n"
        "dev = _hip.get_device()\n"
        "assert dev.lib.rr_build_flags() & 1\n"
        "buf = dev.malloc(1000)\n"
        "dev.memset(buf, 1000); dev.sync()            # in bounds: fine\n"
        "_hip._check(dev.lib, dev.lib.rr_memset(dev.ctx, buf.ptr, 0, 1016))   # 16 bytes past the end\n"
        "try:\n"
        "    dev.sync()\n"
        "except _hip.HipError as e:\n"
        "    assert 'RR_BOUNDS' in str(e) and 'offset 1000' in str(e), str(e)\n"
        "    print('caught')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REVRAND_HIP_LIB=DEBUG_LIB), capture_output=True,
                       text=True, timeout=600)
    assert "caught" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_bounds_build_checks_the_device_of_every_launch():
    """-DRR_BOUNDS wraps every kernel launch of the library in a check that the calling thread's current HIP device is the
    device of the stream launched on (rr_internal.h) -- what the in-process device group relies on and a one-GPU box cannot
    show broken.  A sharded `_elbo` (members on distinct GPUs when the box has two) makes thousands of such launches from
    several host threads: all checked, none in violation (a violation fails the next synchronisation)."""
    if not os.path.exists(DEBUG_LIB):
        pytest.skip("make -C revrand_amd/csrc debug has not been run")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from revrand_amd import _hip, multigpu\n"
        "import revrand_amd.basis_functions as bs\n"
        "from revrand_amd.slm import StandardLinearModel\n"
        "lib = _hip.load_library()\n"
        "assert lib.rr_build_flags() & 1 and lib.rr_debug_launch_checks() == 0\n"
        "v = multigpu.visible_devices()\n"
        "devices = list(range(min(v, 4))) if v >= 2 else [0, 0, 0]\n"
        "rs = np.random.RandomState(0)\n"
        "X = rs.randn(20000, 5).astype(np.float32); y = np.sin(X[:, 0]).astype(np.float32)\n"
        "slm = StandardLinearModel(bs.RandomRBF(nbases=64, Xdim=5, random_state=1) + bs.LinearBasis(), nstarts=0, maxiter=3,\n"
        "                          devices=devices).fit(X, y)\n"
        "slm.predict_moments(X[:3000])\n"
        "multigpu.get_group(devices).sync()\n"
        "n = lib.rr_debug_launch_checks()\n"
        "assert n > 100, n\n"
        "print('checked', n, 'launches on', devices)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REVRAND_HIP_LIB=DEBUG_LIB), capture_output=True,
                       text=True, timeout=900)
    assert "checked" in r.stdout and "RR_BOUNDS" not in (r.stdout + r.stderr), (r.stdout[-1500:], r.stderr[-3000:])
