// K12 placeholder — replaced by the KV-cached hipGraph decoder later in this round.
#include "common.h"
struct mrmt3_decoder { int dummy; };
extern "C" int mrmt3_decoder_create(mrmt3_decoder** out, int, int, int, int, int, int, int, int, int, float) {
  if (out) *out = nullptr;
  mrmt3_set_error("decoder: not built yet");
  return MRMT3_ERR_UNSUPPORTED;
}
extern "C" void mrmt3_decoder_destroy(mrmt3_decoder*) {}
extern "C" int mrmt3_decoder_begin(mrmt3_decoder*, const mrmt3_decoder_weights*, const void*, const void*, int, int,
                                   int64_t*, int, int, int, int, void*) { return MRMT3_ERR_UNSUPPORTED; }
extern "C" int mrmt3_decoder_run(mrmt3_decoder*, int, void*) { return MRMT3_ERR_UNSUPPORTED; }
extern "C" int mrmt3_decoder_poll(mrmt3_decoder*, int32_t*, void*) { return MRMT3_ERR_UNSUPPORTED; }
