// Host-side glue of libmvpops: ABI version and launch-error reporting.
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"

namespace mvp {

static thread_local char g_last_err[256] = "";

int check_launch(const char *what) {
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) return MVP_OK;
  std::snprintf(g_last_err, sizeof(g_last_err), "%s: %s", what,
                hipGetErrorString(err));
  return MVP_ELAUNCH;
}

}  // namespace mvp

extern "C" int mvp_abi_version(void) { return MVP_ABI_VERSION; }

extern "C" const char *mvp_last_hip_error(void) { return mvp::g_last_err; }
