// N = 1024: the product configuration (fsea_configs.h).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft1024, "", FSEA_CFG_1024)
FSEA_DEFINE_WINDOWED(fsea_fft1024, FSEA_WIN)
FSEA_REGISTER_BEGIN(1024)
FSEA_REGISTER_WIN(fsea_fft1024)
FSEA_REGISTER_END
