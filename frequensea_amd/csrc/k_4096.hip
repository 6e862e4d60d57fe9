// N = 4096: 256 lanes x 16 points, 16 x 16 x 16, four workgroups per CU.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft4096, "", FSEA_CFG_4096)
extern "C" int fsea_kernels_4096(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft4096_entry();
    return n;
}
