// TEST-ONLY shim: lets the plain-C++ maths headers of theseus_amd/csrc (lie.cuh, dual.cuh, unroll_se3.cuh) compile for the
// HOST with g++, so that tests can check the device formulas against torch autograd through the oracle without a GPU.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
