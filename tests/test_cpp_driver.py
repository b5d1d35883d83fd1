"""The C++ driver tests/cpp/test_HSS_seq.cpp (reference-shaped classes, reference's command line and
pass criteria) over a subset of the reference's CTest lines (test/CMakeLists.txt:57-143).
CPU: linked against the emulator build; GPU (-m gpu): linked against the product library."""
import os
import subprocess

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
