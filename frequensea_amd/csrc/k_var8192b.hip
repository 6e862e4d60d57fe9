// Tuning variants of the 8192-point kernel (twiddle sources).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192notwl, "notwl", FSEA_CFG_8192_NOTWL)
FSEA_DEFINE_KERNEL(fsea_fft8192notwr, "notwr", FSEA_CFG_8192_NOTWR)
FSEA_DEFINE_KERNEL(fsea_fft8192H, "H", FSEA_CFG_8192_H)
FSEA_DEFINE_KERNEL(fsea_fft8192HB, "HB", FSEA_CFG_8192_HB)
FSEA_DEFINE_KERNEL(fsea_fft4096H, "H", FSEA_CFG_4096_H)
extern "C" int fsea_kernels_var8192b(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft8192notwl_entry();
    if (n < cap) out[n++] = fsea_fft8192notwr_entry();
    if (n < cap) out[n++] = fsea_fft8192H_entry();
    if (n < cap) out[n++] = fsea_fft8192HB_entry();
    if (n < cap) out[n++] = fsea_fft4096H_entry();
    return n;
}
