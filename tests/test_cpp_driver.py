"""The C++ driver tests/cpp/test_HSS_seq.cpp (reference-shaped classes, reference's command line and
pass criteria) over a subset of the reference's CTest lines (test/CMakeLists.txt:57-143).
CPU: linked against the emulator build; GPU (-m gpu): linked against the product library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "strumpack_amd", "csrc", "host")
SRC = os.path.join(ROOT, "tests", "cpp", "test_HSS_seq.cpp")

LINES = [
    "L 10 --hss_leaf_size 3 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 32 --hss_dd 4",
    "T 200 --hss_leaf_size 128 --hss_rel_tol 1 --hss_abs_tol 1e-10 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 128 --hss_dd 8",
    "U 200 --hss_leaf_size 16 --hss_rel_tol 1e-1 --hss_abs_tol 1e-10 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 128 --hss_dd 4",
    "T 10 --hss_leaf_size 16 --hss_rel_tol 1e-10 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm original --hss_d0 128 --hss_dd 4",
    "U 300 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_compression_algorithm stable --hss_d0 16 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo perm --hss_nnz0 2 --hss_nnz 2",
    "T 500 --hss_leaf_size 128 --hss_rel_tol 1e-10 --hss_abs_tol 1e-13 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 16 --hss_dd 4",
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo chunk --hss_nnz0 2 --hss_nnz 2",
]
GPU_LINES = LINES + [
    "T 500 --hss_leaf_size 128 --hss_rel_tol 1e-10 --hss_abs_tol 1e-13 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 16 --hss_dd 4",
    "L 500 --hss_leaf_size 16 --hss_rel_tol 1e-10 --hss_abs_tol 1e-10 --hss_disable_sync --hss_compression_algorithm original --hss_d0 128 --hss_dd 8",
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch Gaussian",
    "T 4096",
    "T 8192 --hss_leaf_size 256 --hss_rel_tol 1e-4",
    # HSS_seq_23 .. 26: the SJLT sketch
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo chunk --hss_nnz0 2 --hss_nnz 2",
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo chunk --hss_nnz0 4 --hss_nnz 4",
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo perm --hss_nnz0 2 --hss_nnz 2",
    "T 1000 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 8 --hss_dd 8 --hss_compression_sketch SJLT --hss_SJLT_algo perm --hss_nnz0 4 --hss_nnz 4",
]


KRR_SRC = os.path.join(ROOT, "tests", "cpp", "KernelRegression.cpp")
KRR_DATA = os.path.join(ROOT, "tests", "golden", "data", "susy_10Kn")


def build(libdir, libname, out, src=SRC):
    cmd = ["g++", "-O2", "-std=c++17", "-I" + HOST, "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True)
    return out


def run(exe, line):
    r = subprocess.run([exe] + line.split(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "# exiting" in r.stdout


@pytest.fixture(scope="module")
def emu_exe(tmp_path_factory):
    import emu_lib
    emu_lib.build()
    return build(os.path.dirname(emu_lib.PATH), "strumpack_amd_emu", str(tmp_path_factory.mktemp("cpp") / "test_HSS_seq_emu"))


@pytest.mark.parametrize("line", LINES)
def test_cpp_driver_emulator(emu_exe, line):
    run(emu_exe, line)


@pytest.mark.gpu
@pytest.mark.parametrize("line", GPU_LINES)
def test_cpp_driver_gpu(tmp_path_factory, line):
    from strumpack_amd import _loader
    exe = build(os.path.dirname(_loader.lib_path()), "strumpack_amd", str(tmp_path_factory.mktemp("cpp") / "test_HSS_seq"))
    run(exe, line)


def run_krr(exe, points, min_score, extra=()):
    env = dict(os.environ, KRR_MAX_POINTS=str(points), KRR_MIN_SCORE=str(min_score))
    r = subprocess.run([exe, KRR_DATA, "8", "1.3", "3.11", "1", "Gauss", "test", *extra], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "# prediction score:" in r.stdout and "compression succeeded" in r.stdout


def test_kernel_regression_driver_emulator(tmp_path_factory):
    """examples/dense/KernelRegression.cpp-shaped driver on a prefix of its shipped data set (the reference scores 78 %
    on these 400 points, tests/golden/kernel_golden.json: regression_gauss_400)."""
    import emu_lib
    emu_lib.build()
    exe = build(os.path.dirname(emu_lib.PATH), "strumpack_amd_emu", str(tmp_path_factory.mktemp("cpp") / "krr_emu"), KRR_SRC)
    run_krr(exe, 150, 65.0, ["--hss_leaf_size", "32", "--hss_approximate_neighbors", "150"])


@pytest.mark.gpu
def test_kernel_regression_driver_gpu(tmp_path_factory):
    """The reference's example with its defaults on the full shipped data set: the reference scores 79.0 %."""
    from strumpack_amd import _loader
    exe = build(os.path.dirname(_loader.lib_path()), "strumpack_amd", str(tmp_path_factory.mktemp("cpp") / "krr"), KRR_SRC)
    run_krr(exe, 10000, 78.0)


WR_SRC = os.path.join(ROOT, "tests", "cpp", "test_write_read.cpp")


def test_write_read_round_trip_emulator(tmp_path_factory):
    import emu_lib
    emu_lib.build()
    d = tmp_path_factory.mktemp("cpp")
    exe = build(os.path.dirname(emu_lib.PATH), "strumpack_amd_emu", str(d / "wr_emu"), WR_SRC)
    r = subprocess.run([exe, "200", str(d / "h.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_write_read_round_trip_gpu(tmp_path_factory):
    from strumpack_amd import _loader
    d = tmp_path_factory.mktemp("cpp")
    exe = build(os.path.dirname(_loader.lib_path()), "strumpack_amd", str(d / "wr"), WR_SRC)
    r = subprocess.run([exe, "3000", str(d / "h.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr


UK_SRC = os.path.join(ROOT, "tests", "cpp", "test_user_kernel.cpp")


def test_user_defined_kernel_emulator(tmp_path_factory):
    """a subclass of kernel::Kernel<double> that only overrides the virtual evaluation (kernel/Kernel.hpp:73-170): compressed
    through HSSMatrix(K, opts) with its blocks evaluated on the host, fit_HSS, predict"""
    import emu_lib
    emu_lib.build()
    d = tmp_path_factory.mktemp("cpp")
    exe = build(os.path.dirname(emu_lib.PATH), "strumpack_amd_emu", str(d / "uk_emu"), UK_SRC)
    r = subprocess.run([exe, "500", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_user_defined_kernel_gpu(tmp_path_factory):
    from strumpack_amd import _loader
    d = tmp_path_factory.mktemp("cpp")
    exe = build(os.path.dirname(_loader.lib_path()), "strumpack_amd", str(d / "uk"), UK_SRC)
    r = subprocess.run([exe, "3000", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr


# ---- the reference's OWN drivers, compiled unmodified from where they lie (build container only: /root/reference does not
# exist on the GPU box) against include/{dense,HSS,structured,kernel,misc}/*.hpp and linked with the emulator library ----
REF = "/root/reference"
REF_LINES = [
    "T 1000",
    "T 200 --hss_leaf_size 128 --hss_rel_tol 1 --hss_abs_tol 1e-10 --hss_disable_sync --hss_compression_algorithm stable --hss_d0 128 --hss_dd 8",
    "L 300 --hss_leaf_size 16 --hss_rel_tol 1e-10 --hss_abs_tol 1e-10 --hss_compression_algorithm original --hss_d0 128 --hss_dd 8",
    "U 500 --hss_leaf_size 32 --hss_rel_tol 1e-5 --hss_abs_tol 1e-10 --hss_enable_sync --hss_compression_algorithm stable --hss_d0 16 --hss_dd 8",
]


def build_ref(src, out, libdir, libname):
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True)
    return out


@pytest.fixture(scope="module")
def ref_exe(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "test", "test_HSS_seq.cpp")):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    return build_ref(os.path.join(REF, "test", "test_HSS_seq.cpp"), str(tmp_path_factory.mktemp("cpp") / "ref_test_HSS_seq"),
                     os.path.dirname(emu_lib.PATH), "strumpack_amd_emu")


@pytest.mark.parametrize("line", REF_LINES)
def test_reference_own_driver_unmodified(ref_exe, line):
    """/root/reference/test/test_HSS_seq.cpp itself -- HSSMatrix(A, opts), child(c)->apply, apply_HSS on a child, get, extract,
    DenseMatrix::extract, free gemm, factor / solve, partial_factor / Schur_update -- with its own pass criteria"""
    run(ref_exe, line)


# the reference's BLR CTest lines (test/CMakeLists.txt:162-184)
REF_BLR_LINES = [
    "300 --blr_factor_algorithm RL",
    "300 --blr_factor_algorithm LL",
    "300 --blr_factor_algorithm Star --blr_compression_kernel full",
    "300 --blr_factor_algorithm Star --blr_compression_kernel half",
    "300 --blr_factor_algorithm Comb --blr_compression_kernel full",
    "300 --blr_factor_algorithm Comb --blr_compression_kernel half",
    "700 --blr_leaf_size 64 --blr_rel_tol 1e-6 --blr_low_rank_algorithm ACA",
]


@pytest.fixture(scope="module")
def ref_blr_exe(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "test", "test_BLR_seq.cpp")):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    return build_ref(os.path.join(REF, "test", "test_BLR_seq.cpp"), str(tmp_path_factory.mktemp("cpp") / "ref_test_BLR_seq"),
                     os.path.dirname(emu_lib.PATH), "strumpack_amd_emu")


@pytest.mark.parametrize("line", REF_BLR_LINES)
def test_reference_own_blr_driver_unmodified(ref_blr_exe, line):
    """/root/reference/test/test_BLR_seq.cpp itself -- BLROptions::set_from_command_line, ClusterTree::leaf_sizes,
    BLRMatrix(m, tiles, m, tiles), compress_and_factor, solve -- with its own pass criterion, on its six CTest lines (Star and
    Comb run the RL schedule here: BLRMatrix.hpp) and one more with ACA tiles"""
    run(ref_blr_exe, line)


def test_reference_structured_example_unmodified(tmp_path_factory):
    """/root/reference/examples/dense/testStructured.cpp itself: the structured:: interface over all construction routes
    (dense, elements, blocks, partially matrix-free), factor / solve, shift, and the compressed matrix as the preconditioner of
    iterative::GMRes / BiCGStab; BLR stops at factor() and LOSSY / LOSSLESS at construction with the exceptions the example
    itself catches (the reference's BLR has no factor() behind this interface either)"""
    src = os.path.join(REF, "examples", "dense", "testStructured.cpp")
    if not os.path.exists(src):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    exe = build_ref(src, str(tmp_path_factory.mktemp("cpp") / "ref_testStructured"), os.path.dirname(emu_lib.PATH), "strumpack_amd_emu")
    r = subprocess.run([exe, "300", "--structured_leaf_size", "64"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    import re
    assert out.count("HSS\n") == 5 and "HSS failed" not in out          # five HSS constructions, all carried through
    assert out.count("BLR\n") == 3 and out.count("BLR failed: Operation factor not supported") == 3   # (one per try block)
    errs = [float(x) for x in re.findall(r"\|\|X-A\\\(A\*X\)\|\|_F/\|\|X\|\|_F = (\S+)", out)]
    assert len(errs) == 10 and max(errs) < 1e-8, errs                    # GMRes and BiCGStab with each HSS matrix as preconditioner
    comp = [float(x) for x in re.findall(r"\|\|A-H\|\|_F/\|\|A\|\|_F = (\S+)", out)]
    assert comp and max(comp) < 1e-3, comp


def test_reference_kernel_regression_example_unmodified(tmp_path_factory):
    """/root/reference/examples/dense/KernelRegression.cpp itself on a prefix of its shipped data set"""
    src = os.path.join(REF, "examples", "dense", "KernelRegression.cpp")
    if not os.path.exists(src):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    d = tmp_path_factory.mktemp("cpp")
    exe = build_ref(src, str(d / "ref_krr"), os.path.dirname(emu_lib.PATH), "strumpack_amd_emu")
    for part, cnt in (("train", 300), ("train_label", 300), ("test", 100), ("test_label", 100)):
        with open(KRR_DATA + "_%s.csv" % part) as f, open(str(d / ("s_%s.csv" % part)), "w") as g:
            g.writelines(f.readlines()[:cnt])
    r = subprocess.run([exe, str(d / "s"), "8", "1.3", "3.11", "1", "Gauss", "test", "--hss_leaf_size", "64", "--hss_quiet"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    score = [ln for ln in r.stdout.splitlines() if "prediction score" in ln]
    assert score and float(score[0].split(":")[1].strip().rstrip("%")) >= 70.0, r.stdout[-500:]


def test_reference_c_example_unmodified(tmp_path_factory):
    """/root/reference/examples/dense/dstructured.c itself (plain C against include/structured/StructuredMatrix.h):
    SP_d_struct_default_options, _from_elements, _mult, _factor, _solve, _destroy"""
    src = os.path.join(REF, "examples", "dense", "dstructured.c")
    if not os.path.exists(src):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    exe = str(tmp_path_factory.mktemp("c") / "ref_dstructured")
    libdir = os.path.dirname(emu_lib.PATH)
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + libdir, "-lstrumpack_amd_emu",
                    "-Wl,-rpath," + libdir, "-lm"], check=True)
    r = subprocess.run([exe, "500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    comp = float(re.search(r"\|\|T-H\|\|_F/\|\|T\|\|_F = (\S+)", r.stdout).group(1))
    sol = float(re.search(r"\|\|X-H\\\(H\*X\)\|\|_F/\|\|X\|\|_F = (\S+)", r.stdout).group(1))
    assert comp < 1e-6 and sol < 1e-10, r.stdout          # (rel_tol 1e-8 in the example)


def test_reference_python_example_unmodified(tmp_path_factory):
    """/root/reference/examples/dense/KernelRegression.py itself: `import STRUMPACKKernel as sp` resolves to
    include/python/STRUMPACKKernel.py (the reference installs its module under the same name), the scikit-learn style classifier
    on this library -- here on the emulator build and a prefix of the example's data set"""
    src = os.path.join(REF, "examples", "dense", "KernelRegression.py")
    if not os.path.exists(src):
        pytest.skip("the reference tree is only present in the build container")
    import emu_lib
    emu_lib.build()
    d = tmp_path_factory.mktemp("py")
    for part, cnt in (("train", 300), ("train_label", 300), ("test", 100), ("test_label", 100)):
        with open(KRR_DATA + "_%s.csv" % part) as f, open(str(d / ("s_%s.csv" % part)), "w") as g:
            g.writelines(f.readlines()[:cnt])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "include", "python"), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_with_emulator.py"), emu_lib.PATH, src, str(d / "s"), "1.3", "3.11", "1",
                        "--hss_leaf_size", "64", "--hss_quiet"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    q = [ln for ln in r.stdout.splitlines() if "HSS KernelRR quality" in ln]
    assert q and float(q[0].split("=")[1].strip().rstrip("%")) >= 70.0, r.stdout[-500:]


BLRF_SRC = os.path.join(ROOT, "tests", "cpp", "test_BLR_front.cpp")


def test_blr_front_cpp_emulator(tmp_path_factory):
    """BLRMatrix<double>::construct_and_partial_factor / trsmLNU_gemm / gemm_trsmUNN with the reference's signatures, driven
    as sparse/fronts/FrontBLR.cpp drives them"""
    import emu_lib
    emu_lib.build()
    d = tmp_path_factory.mktemp("cpp")
    exe = build_ref(BLRF_SRC, str(d / "blr_front_emu"), os.path.dirname(emu_lib.PATH), "strumpack_amd_emu")
    r = subprocess.run([exe, "96", "16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_blr_front_cpp_gpu(tmp_path_factory):
    from strumpack_amd import _loader
    d = tmp_path_factory.mktemp("cpp")
    exe = build_ref(BLRF_SRC, str(d / "blr_front"), os.path.dirname(_loader.lib_path()), "strumpack_amd")
    r = subprocess.run([exe, "1500", "128"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# exiting" in r.stdout, r.stdout + r.stderr
