// Weight gradient of the 3x3 "same" convolutions over one bf16 slice (training plans with bf16 activation storage).
#include "common.h"

namespace egne {

bool wgrad3x3_bf16_supported(const egne_conv_desc& d, long long gzs) { (void)d; (void)gzs; return false; }
int wgrad3x3_bf16_splits(const egne_conv_desc& d) { (void)d; return 1; }
int wgrad3x3_bf16_launch(const egne_conv_desc& d, const egne_bf16* gz, long long gzs, int gzo, float* ws, hipStream_t st) {
  (void)d; (void)gz; (void)gzs; (void)gzo; (void)ws; (void)st;
  return fail(EGNE_ERR_ARG, "wgrad3x3_bf16: not supported");
}

}  // namespace egne
