"""TEST INFRASTRUCTURE: runs a Python script with the package's library path pointed at the CPU emulator build of the
kernel sources (tests/emu), for the GPU-less build container:  python tests/run_with_emulator.py <emu .so> <script> [args ...]"""
import runpy
import sys

from strumpack_amd import _loader

_loader.LIB_PATH = sys.argv[1]
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
