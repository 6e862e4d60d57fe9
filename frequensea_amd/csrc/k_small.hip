// N = 128, 256, 512: one frame per 8/16/16 lanes, no barriers (single-wave frames).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft128, "", FSEA_CFG_128)
FSEA_DEFINE_KERNEL(fsea_fft256, "", FSEA_CFG_256)
FSEA_DEFINE_KERNEL(fsea_fft512, "", FSEA_CFG_512)
extern "C" int fsea_kernels_small(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft128_entry();
    if (n < cap) out[n++] = fsea_fft256_entry();
    if (n < cap) out[n++] = fsea_fft512_entry();
    return n;
}
