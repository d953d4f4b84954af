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
bool ego_fold_is_balanced(int64_t N, int32_t S);

inline int ego_launch_status(const char* kernel) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ego_fail((int)e, "%s: launch failed: %s", kernel, hipGetErrorString(e));
  return EGO_OK;
}

// ---- tracing ("tracing" row) ---------------------------------------------------------------------------
// EGO_ROCTX=1 in the environment: every entry point of the library that queues device work opens a roctx range named after itself
// (roctxRangePushA / roctxRangePop of librocprofiler-sdk-roctx.so, dlopen'ed on first use - no link-time dependency, nothing loaded
// and one predictable branch per call when the switch is off).  `rocprofv3 --marker-trace --kernel-trace -- python bench.py ...`
// then shows which library call (= which row of SURVEY 8(a)) every kernel of every config belongs to, without the stage probes.
struct EgoRoctx {
  int (*push)(const char*);
  int (*pop)();
};
const EgoRoctx* ego_roctx();   // csrc/ego_ops.hip; nullptr when tracing is off or the roctx library is absent

struct EgoRange {
  const EgoRoctx* r;
  explicit EgoRange(const char* name) : r(ego_roctx()) { if (r) r->push(name); }
  ~EgoRange() { if (r) r->pop(); }
  EgoRange(const EgoRange&) = delete;
  EgoRange& operator=(const EgoRange&) = delete;
};
#define EGO_TRACE(name) EgoRange ego_trace_range_(name)
