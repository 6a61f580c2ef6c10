// C-ABI plumbing shared by every entry point: error string, launch check, version, device sync.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void lct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lct_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    lct_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return LCT_OK;
}

extern "C" {
int lctvqa_version(void) { return 100; }
const char* lctvqa_last_error(void) { return g_err; }
int lctvqa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
}
