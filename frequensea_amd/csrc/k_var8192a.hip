// Tuning variants of the 8192-point kernel (pass orders).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192A, "A", FSEA_CFG_8192_A)
FSEA_DEFINE_KERNEL(fsea_fft8192B, "B", FSEA_CFG_8192_B)
FSEA_DEFINE_KERNEL(fsea_fft8192D, "D", FSEA_CFG_8192_D)
extern "C" int fsea_kernels_var8192a(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft8192A_entry();
    if (n < cap) out[n++] = fsea_fft8192B_entry();
    if (n < cap) out[n++] = fsea_fft8192D_entry();
    return n;
}
