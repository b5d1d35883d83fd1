# Instantiates the reference's own StrumpackConfig.h.in template with CMake's
# configure_file (the same mechanism its build uses), for a CPU-only, MPI-free,
# OpenMP build of the HSS path.  Run:  cmake -DREF=/root/reference -DOUT=<dir> -P gen_config.cmake
# Test infrastructure only -- nothing produced here ships in the product.
set(STRUMPACK_USE_OPENMP ON)
set(STRUMPACK_USE_GETOPT ON)
set(STRUMPACK_COUNT_FLOPS ON)
set(STRUMPACK_TASK_TIMERS OFF)
set(STRUMPACK_USE_OPENMP_TASKLOOP ON)
set(STRUMPACK_USE_OPENMP_TASK_DEPEND ON)
set(STRUMPACK_PBLAS_BLOCKSIZE 32)
set(STRUMPACK_VERSION_MAJOR 8)
set(STRUMPACK_VERSION_MINOR 0)
set(STRUMPACK_VERSION_PATCH 0)
configure_file(${REF}/src/StrumpackConfig.h.in ${OUT}/StrumpackConfig.h)
