// Library version (the ABI number bindings can check).
#include "e2k_device.h"
#include "../../include/e2k.h"

using namespace e2k;

extern "C" int e2k_version(void) { return 1; }
