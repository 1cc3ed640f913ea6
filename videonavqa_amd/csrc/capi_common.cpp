// Error plumbing shared by every entry point of libvnqa_hip.so.
#include "vnqa_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void vnqa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vnqa_last_error(void) { return g_err; }
extern "C" int vnqa_version(void) { return VNQA_ABI_VERSION; }
