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

// sizeof() of the descriptor structs, so that the ctypes mirrors can be checked against the compiler
extern "C" int egne_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(egne_conv_desc);
    case 1: return (int)sizeof(egne_loss_desc);
    case 2: return (int)sizeof(egne_bdcn_tail_desc);
    case 3: return (int)sizeof(egne_dst);
    case 4: return (int)sizeof(egne_conv_query);
    case 5: return (int)sizeof(egne_conv_choice);
    default: return -1;
  }
}
