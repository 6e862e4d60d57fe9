// Tuning variants of the 8192-point kernel (twiddle sources).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192notwl, "notwl", FSEA_CFG_8192_NOTWL)
FSEA_DEFINE_KERNEL(fsea_fft8192notwr, "notwr", FSEA_CFG_8192_NOTWR)
extern "C" int fsea_kernels_var8192b(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft8192notwl_entry();
    if (n < cap) out[n++] = fsea_fft8192notwr_entry();
    return n;
}
