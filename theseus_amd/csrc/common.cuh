// Shared host-side helpers for the C ABI (error string, dtype dispatch, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/theseus_hip.h"
#include "lie.cuh"

namespace thx {

std::string& last_error();

inline int fail(const char* what) {
  last_error() = what;
  return -1;
}

inline int fail(const char* who, const char* what) {
  last_error() = std::string(who) + what;
  return -1;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    last_error() = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
  }
  return 0;
}

template <typename T>
inline Eps<T> make_eps(const thx_lie_eps* e) {
  Eps<T> r;
  r.nz = (T)e->near_zero;
  r.dnz = (T)e->d_near_zero;
  r.npi = (T)e->near_pi;
  return r;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace thx

#define THX_DISPATCH(dtype, CALL_F32, CALL_F64) \
  do {                                          \
    if ((dtype) == THX_F32) {                   \
      CALL_F32;                                 \
    } else if ((dtype) == THX_F64) {            \
      CALL_F64;                                 \
    } else {                                    \
      return thx::fail("bad dtype");            \
    }                                           \
  } while (0)
