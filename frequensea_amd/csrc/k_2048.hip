// N = 2048: the product configuration (fsea_configs.h).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft2048, "", FSEA_CFG_2048)
FSEA_DEFINE_WINDOWED(fsea_fft2048, FSEA_WIN)
FSEA_REGISTER_BEGIN(2048)
FSEA_REGISTER_WIN(fsea_fft2048)
FSEA_REGISTER_END
