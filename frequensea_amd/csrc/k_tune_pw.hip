// Tuning variants (round 3): the last butterfly level in power form (FftCfg OPT 8388608), every size class;
// 32 / 64 points with two lanes per frame.
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL_U8(fsea_fft512f8, "f8", FSEA_CFG_512_F8)
FSEA_DEFINE_KERNEL_U8(fsea_fft256f8, "f8", FSEA_CFG_256_F8)
FSEA_DEFINE_KERNEL_U8(fsea_fft32t2a, "t2a", FSEA_CFG_32_T2A)
FSEA_DEFINE_KERNEL_U8(fsea_fft32t2b, "t2b", FSEA_CFG_32_T2B)
FSEA_DEFINE_KERNEL_U8(fsea_fft32t2c, "t2c", FSEA_CFG_32_T2C)
FSEA_DEFINE_KERNEL_U8(fsea_fft64t2c, "t2c", FSEA_CFG_64_T2C)
FSEA_DEFINE_KERNEL_U8(fsea_fft64t2d, "t2d", FSEA_CFG_64_T2D)
FSEA_REGISTER_BEGIN(tune_pw)
FSEA_REGISTER(fsea_fft512f8)
FSEA_REGISTER(fsea_fft256f8)
FSEA_REGISTER(fsea_fft32t2a)
FSEA_REGISTER(fsea_fft32t2b)
FSEA_REGISTER(fsea_fft32t2c)
FSEA_REGISTER(fsea_fft64t2c)
FSEA_REGISTER(fsea_fft64t2d)
FSEA_REGISTER_END
