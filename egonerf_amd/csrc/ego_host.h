// Host-side error plumbing shared by the translation units of libegonerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/egonerf_hip.h"

inline char* ego_err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

inline int ego_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(ego_err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define EGO_REQUIRE(cond, msg) \
  do {                         \
    if (!(cond)) return ego_fail(EGO_E_BADARG, "%s", msg); \
  } while (0)

// csrc/ego_shade.hip: can ego_shade_composite serve this scene and sample count? (asked by ego_render_forward)
bool ego_can_fold_composite(const ego_scene* sc, int32_t S);

inline int ego_launch_status(const char* kernel) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ego_fail((int)e, "%s: launch failed: %s", kernel, hipGetErrorString(e));
  return EGO_OK;
}
