// N = 4096: the product configuration (fsea_configs.h).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft4096, "", FSEA_CFG_4096)
FSEA_DEFINE_WINDOWED(fsea_fft4096, FSEA_WIN)
FSEA_REGISTER_BEGIN(4096)
FSEA_REGISTER_WIN(fsea_fft4096)
FSEA_REGISTER_END
