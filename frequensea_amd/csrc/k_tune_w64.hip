// Tuning variants (round 3): 4096 points as one wavefront per frame (64 lanes x 64 points, no s_barrier); the pixel
// epilogue on v_cvt_pk_u8_f32; 256 points with 64 points per lane.
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL_U8(fsea_fft4096w64, "w64", FSEA_CFG_4096_W64)
FSEA_DEFINE_KERNEL_U8(fsea_fft4096w64b, "w64b", FSEA_CFG_4096_W64B)
FSEA_DEFINE_KERNEL_U8(fsea_fft4096s2, "s2", FSEA_CFG_4096_S2)
FSEA_DEFINE_KERNEL_U8(fsea_fft4096pk, "pk", FSEA_CFG_4096_PK)
FSEA_DEFINE_KERNEL_U8(fsea_fft4096px0, "px0", FSEA_CFG_4096_PX0)
FSEA_DEFINE_KERNEL_U8(fsea_fft8192pk, "pk", FSEA_CFG_8192_PK)
FSEA_DEFINE_KERNEL_U8(fsea_fft8192px0, "px0", FSEA_CFG_8192_PX0)
FSEA_DEFINE_KERNEL_U8(fsea_fft256pk, "pk", FSEA_CFG_256_PK)
FSEA_DEFINE_KERNEL_U8(fsea_fft256px0, "px0", FSEA_CFG_256_PX0)
FSEA_DEFINE_KERNEL_U8(fsea_fft1024px0, "px0", FSEA_CFG_1024_PX0)
FSEA_DEFINE_KERNEL_U8(fsea_fft256p64, "p64", FSEA_CFG_256_P64)
FSEA_REGISTER_BEGIN(tune_w64)
FSEA_REGISTER(fsea_fft4096w64)
FSEA_REGISTER(fsea_fft4096w64b)
FSEA_REGISTER(fsea_fft4096s2)
FSEA_REGISTER(fsea_fft4096pk)
FSEA_REGISTER(fsea_fft4096px0)
FSEA_REGISTER(fsea_fft8192pk)
FSEA_REGISTER(fsea_fft8192px0)
FSEA_REGISTER(fsea_fft256pk)
FSEA_REGISTER(fsea_fft256px0)
FSEA_REGISTER(fsea_fft1024px0)
FSEA_REGISTER(fsea_fft256p64)
FSEA_REGISTER_END
