"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/slam_engine.h declares (no compute calls without a GPU)."""
import ctypes as C
import os

from slamkit_amd import engine as E


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    names = E.header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/slam_engine.h but not exported"
    # and the binding covers exactly the declared functions
    assert sorted(lib._slam_signatures) == names


def test_engine_layout_without_gpu():
    lib = E.load_library()
    assert b"gfx950" in lib.slam_version()
    d = E.SlamModelDesc(24, 896, 14, 2, 64, 4864, 502, 0, 1e-6, 10000.0)
    eng = E.Engine(d)
    # 358,347,904 reference parameters + 10 zero pad rows of the 512-row embedding image
    assert eng.n_params == 358_347_904 + 10 * 896
    t = eng.tensors
    assert t["embed"].rows == 512 and t["embed"].offset == 0
    assert t["layers.0.wqkv"].rows == 1152 and t["layers.0.wgu"].rows == 9728
    assert all(s.offset % 8 == 0 for s in t.values())
    assert eng.workspace_bytes(8192) > 7 * 2**30
    # bad descriptions are rejected with an error code, not a crash
    h = C.c_void_p()
    bad = E.SlamModelDesc(2, 100, 4, 2, 32, 512, 502, 0, 1e-6, 10000.0)
    assert lib.slam_engine_create(C.byref(bad), C.byref(h)) == -1
    assert lib.slam_backward(None, 1.0, 0, E.BUCKET_CB(0), None, None) == -1
    eng.close()


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(OSError):
        E.load_library(str(tmp_path / "nope.so"))


def test_gemm_tn_workspace_covers_every_contraction_length():
    """The wgrad planners re-plan from the runtime token count (packed micro-batches vary): the workspace sized for Mmax must
    cover the plan of every M <= Mmax (round-2 advisor finding: H 256, I 1024, Mmax 2048 - at M = 1600 the gate|up plan
    needed 10.0 MiB of an 8.0 MiB allocation). Host-only: the sizing function does not touch the device."""
    from slamkit_amd import engine as E
    lib = E.load_library()
    assert lib.slam_op_gemm_tn_workspace(2048, 2048, 256) >= 10 * 2 ** 20
    for N, K in ((2048, 256), (9728, 896), (896, 4864), (1152, 896), (896, 896)):
        cap = lib.slam_op_gemm_tn_workspace(8192, N, K)
        for M in range(64, 8192 + 1, 64):
            assert lib.slam_op_gemm_tn_workspace(M, N, K) <= cap, (M, N, K)


def test_c_consumer_compiles_against_the_header(tmp_path):
    """include/slam_engine.h is plain C and tools/examples/c_dp_consumer.c - one data-parallel optimizer step of a consumer without
    torch: forward, backward with the bucket callback calling slam_allreduce_grads_async, slam_comm_finish, clip, AdamW - still
    matches it (compile only: gcc, C99, warnings as errors)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        import pytest
        pytest.skip("no C compiler on this box")
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), "-c",
                        os.path.join(root, "tools", "examples", "c_dp_consumer.c"), "-o", str(tmp_path / "c.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_library_cannot_skip_output_stores():
    """Round-5 verdict / advisor: bit 1 of gemm_nt_store made every NT GEMM drop its output stores and was reachable from the
    environment (SLAM_GEMM_NT_STORE). It is compiled out of the product library now (-DSLAM_PROBES builds only): the option is
    rejected beyond {0, 1}, for the process default and for an engine, and UnitLM no longer forwards it from the environment.
    Host-only calls."""
    lib = E.load_library()
    assert lib.slam_set_option(None, b"gemm_nt_store", 2) == -1
    assert lib.slam_set_option(None, b"gemm_nt_store", 3) == -1
    assert lib.slam_set_option(None, b"gemm_nt_store", 1) == 0
    assert lib.slam_set_option(None, b"gemm_nt_store", 0) == 0
    eng = E.Engine(E.SlamModelDesc(2, 64, 4, 2, 64, 128, 502, 0, 1e-6, 10000.0))
    import pytest
    with pytest.raises(E.EngineError, match="out of range"):
        eng.set_option("gemm_nt_store", 2)
    eng.set_option("gemm_nt_store", 0)
    with pytest.raises(E.EngineError):
        eng.set_option("grad_final_next", 3)
    eng.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "gemm_nt_store" not in open(os.path.join(root, "slamkit_amd", "model", "unit_lm.py")).read()
    src = open(os.path.join(root, "slamkit_amd", "csrc", "build.py")).read()
    assert "-DSLAM_PROBES" in src and "libslam_engine_probes.so" in src


def test_address_arithmetic_of_the_32x32x16_gemm_paths():
    """tools/layout_sim.py: the index arithmetic of gemm.hip's 32x32x16 main loops restated on labels - LDS-DMA placement ->
    swizzled fragment reads -> MFMA lane maps -> epilogue32 columns - for the 128 x 128, the eight-wave 256 x 256 and the four-wave
    256 x 256 kernel: every stored element is the product sum it should be, every ds_read_b128 lane group is conflict-free."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("layout_sim", os.path.join(root, "tools", "layout_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.sim_128x128()
    m.sim_256_8wave()
    m.sim_256_4wave()
