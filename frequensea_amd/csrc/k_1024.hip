// N = 1024: 32 lanes x 32 points, radix 32 x 32, one LDS exchange, no barriers.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft1024, "", FSEA_CFG_1024)
extern "C" int fsea_kernels_1024(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft1024_entry();
    return n;
}
