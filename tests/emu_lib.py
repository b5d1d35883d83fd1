"""TEST INFRASTRUCTURE ONLY: builds (once) and loads tests/emu/_build/libstrumpack_amd_emu.so -- the
product's kernel/host sources compiled with g++ on the CPU SIMT emulator.  The product package
never loads this library; it exists so `-m "not gpu"` tests can exercise host logic and kernel
index arithmetic in the GPU-less container."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "emu", "_build", "libstrumpack_amd_emu.so")


def build():
    subprocess.run(["make", "-C", os.path.join(HERE, "emu"), "-j8"], check=True,
                   stdout=subprocess.DEVNULL)
    return PATH
