// Small utility kernels: version, fp32 -> bf16 parameter shadow cast, column sums (bias gradients).
#include "e2k_device.h"
#include "../../include/e2k.h"

using namespace e2k;

extern "C" int e2k_version(void) { return 1; }
