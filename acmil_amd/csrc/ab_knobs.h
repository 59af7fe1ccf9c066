// ab_knobs.h -- A/B measurement switches (environment variables) exist ONLY in builds made with -DACMIL_AB_KNOBS:
// libacmil_hip_ab.so (Makefile target; tools/ and the tests that compare kernel variants load it through ACMIL_HIP_LIB).
// The product library libacmil_hip.so reads NO environment variable: a stray ACMIL_* in a user's environment cannot change which
// kernels run or in which order they sum (tests/test_cabi_cpu.py asserts that no such name is even present in the binary).
#pragma once
#include <stdlib.h>
#ifdef ACMIL_AB_KNOBS
#define ACMIL_AB_ENV(name) getenv(name)
#else
#define ACMIL_AB_ENV(name) ((const char*)nullptr)
#endif
