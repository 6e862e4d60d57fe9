// fsea_configs.h -- the kernel configurations compiled into libfsea_hip.so (and, for index-math
// tests on the CPU, into tests/emu): one per transform size.
// FftCfg arguments: N, T, FPW, WPE, NP, R0, R1, R2, R3, TWL, TWR, ABL, OPT (schedule options).
// OPT 4096: the output rows are streamed (nt stores: they are not read again by the kernel, and kept
// out of the caches they would only evict other lines) -- every size; OPT 32768: the input bytes as
// well (nt loads) -- sizes whose pass-0 loads are at least a dword per lane (profiles/r02_tune_nt_*).
#pragma once

// single-wave frames, no s_barrier; 32 points per lane from 128 points up (dword pass-0 loads)
#define FSEA_CFG_32 32, 4, 64, 2, 2, 8, 4, 1, 1, true, true, 0, 4096
#define FSEA_CFG_64 64, 4, 64, 2, 2, 16, 4, 1, 1, true, true, 0, 4096
#define FSEA_CFG_128 128, 4, 64, 2, 2, 16, 8, 1, 1, true, true, 0, 4096
#define FSEA_CFG_256 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, 4096
#define FSEA_CFG_512 512, 16, 16, 2, 2, 32, 16, 1, 1, true, true, 0, 4096
#define FSEA_CFG_1024 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, 4106
#define FSEA_CFG_2048 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, 36894
// multi-wave frames: 32 points per lane (4096: two waves per frame, two frames per workgroup), the
// middle pass's twiddles deferred and register-resident (OPT 128)
#define FSEA_CFG_4096 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, 37054
#define FSEA_CFG_8192 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, 37022
#define FSEA_CFG_16384 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, 37000
