// Error channel + version of libpc3d_hip.so.
#include <stdarg.h>
#include "pc3d_common.h"

namespace pc3d {
static thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pc3d

extern "C" int pc3d_version(void) { return 100; }  // 0.1.0
extern "C" const char* pc3d_last_error(void) { return pc3d::g_err; }
