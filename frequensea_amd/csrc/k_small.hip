// N = 32 ... 512: one frame per 4/4/8/16/16 lanes, no barriers (single-wave frames).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft32, "", FSEA_CFG_32)
FSEA_DEFINE_KERNEL(fsea_fft64, "", FSEA_CFG_64)
FSEA_DEFINE_KERNEL(fsea_fft128, "", FSEA_CFG_128)
FSEA_DEFINE_KERNEL(fsea_fft256, "", FSEA_CFG_256)
FSEA_DEFINE_KERNEL(fsea_fft512, "", FSEA_CFG_512)
FSEA_REGISTER_BEGIN(small)
FSEA_REGISTER(fsea_fft32)
FSEA_REGISTER(fsea_fft64)
FSEA_REGISTER(fsea_fft128)
FSEA_REGISTER(fsea_fft256)
FSEA_REGISTER(fsea_fft512)
FSEA_REGISTER_END
