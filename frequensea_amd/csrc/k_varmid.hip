// Tuning variants of the 2048/4096/16384-point kernels (other radix orders).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft4096C, "C", FSEA_CFG_4096_C)
FSEA_DEFINE_KERNEL(fsea_fft4096D, "D", FSEA_CFG_4096_D)
FSEA_DEFINE_KERNEL(fsea_fft16384B, "B", FSEA_CFG_16384_B)
FSEA_DEFINE_KERNEL(fsea_fft2048B, "B", FSEA_CFG_2048_B)
FSEA_DEFINE_KERNEL(fsea_fft2048C, "C", FSEA_CFG_2048_C)
extern "C" int fsea_kernels_varmid(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft4096C_entry();
    if (n < cap) out[n++] = fsea_fft4096D_entry();
    if (n < cap) out[n++] = fsea_fft16384B_entry();
    if (n < cap) out[n++] = fsea_fft2048B_entry();
    if (n < cap) out[n++] = fsea_fft2048C_entry();
    return n;
}
