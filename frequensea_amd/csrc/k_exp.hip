// Schedule-option variants (FftCfg::OPT): same results, different instruction order.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192x0, "x0", FSEA_CFG_8192_X0)
FSEA_DEFINE_KERNEL(fsea_fft8192x7, "x7", FSEA_CFG_8192_X7)
FSEA_DEFINE_KERNEL(fsea_fft4096x0, "x0", FSEA_CFG_4096_X0)
extern "C" int fsea_kernels_exp(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft8192x0_entry();
    if (n < cap) out[n++] = fsea_fft8192x7_entry();
    if (n < cap) out[n++] = fsea_fft4096x0_entry();
    return n;
}
