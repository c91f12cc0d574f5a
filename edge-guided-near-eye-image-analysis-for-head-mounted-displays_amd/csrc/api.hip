// Error buffer + version of the C-ABI.
#include "common.h"
namespace egne {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace egne
extern "C" const char* egne_last_error(void) { return egne::err_buf(); }
extern "C" int egne_version(void) { return 100; }
