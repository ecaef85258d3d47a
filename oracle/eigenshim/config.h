// oracle/eigenshim/config.h -- TEST INFRASTRUCTURE ONLY.
// Stands in for Thirdparty/g2o/config.h, which the reference's CMake generates from config.h.in
// (configure_file, Thirdparty/g2o/CMakeLists.txt): no OpenMP (G2O_USE_OPENMP is OFF by default),
// shared library.  Found through -I oracle/eigenshim/inc/a/b + the sources' "../../config.h".
#ifndef G2O_CONFIG_H
#define G2O_CONFIG_H
#define G2O_SHARED_LIBS 1
#endif
