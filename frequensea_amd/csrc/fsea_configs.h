// fsea_configs.h -- the kernel configurations that are compiled into
// libfsea_hip.so (and, for index-math tests on the CPU, into tests/emu).
// FftCfg arguments: N, T, FPW, WPE, NP, R0, R1, R2, R3, TWL, TWR, ABL, OPT (schedule options).
#pragma once

// single-wave frames, no s_barrier
#define FSEA_CFG_128 128, 8, 32, 2, 2, 16, 8, 1, 1, true, true
#define FSEA_CFG_256 256, 16, 16, 2, 2, 16, 16, 1, 1, true, true
#define FSEA_CFG_512 512, 16, 16, 2, 2, 32, 16, 1, 1, true, true
#define FSEA_CFG_1024 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, 10
#define FSEA_CFG_2048 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, 14
// multi-wave frames (4096: 16 points per lane, four workgroups per CU)
#define FSEA_CFG_4096 4096, 256, 1, 4, 3, 16, 16, 16, 1, true, true, 0, 10
#define FSEA_CFG_8192 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, 30
#define FSEA_CFG_16384 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, 8

// ---- tuning variants (selected with fsea_plan_create_variant; not the defaults) ----
#define FSEA_CFG_8192_A 8192, 256, 1, 2, 3, 32, 16, 16, 1, true, true
#define FSEA_CFG_8192_B 8192, 256, 1, 2, 3, 16, 32, 16, 1, true, true
#define FSEA_CFG_8192_D 8192, 256, 1, 2, 3, 8, 32, 32, 1, true, true
#define FSEA_CFG_8192_NOTWL 8192, 256, 1, 2, 3, 16, 32, 16, 1, false, true
#define FSEA_CFG_8192_NOTWR 8192, 256, 1, 2, 3, 16, 32, 16, 1, true, false
#define FSEA_CFG_1024_B 1024, 64, 4, 4, 3, 16, 16, 4, 1, true, true
#define FSEA_CFG_1024_C 1024, 64, 4, 4, 3, 4, 16, 16, 1, true, true
#define FSEA_CFG_1024_D 1024, 32, 4, 2, 2, 32, 32, 1, 1, true, true
#define FSEA_CFG_4096_B 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true
#define FSEA_CFG_4096_C 4096, 128, 2, 2, 3, 16, 8, 32, 1, true, true
#define FSEA_CFG_4096_D 4096, 128, 2, 2, 3, 8, 16, 32, 1, true, true
#define FSEA_CFG_16384_B 16384, 512, 1, 2, 3, 32, 32, 16, 1, true, true
#define FSEA_CFG_2048_B 2048, 64, 4, 2, 3, 8, 8, 32, 1, true, true
#define FSEA_CFG_2048_C 2048, 64, 4, 2, 3, 4, 16, 32, 1, true, true
// measurement-only ablations of 8192 "B" (results are wrong by design; never the default)
#define FSEA_CFG_8192_B_NOST 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 1
#define FSEA_CFG_8192_B_NOLDS 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 2
#define FSEA_CFG_8192_B_NOFLOP 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 4
#define FSEA_CFG_8192_B_IO 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 6
#define FSEA_CFG_8192_B_VALU 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 3
// the defaults' schedule options switched off / changed (FftCfg::OPT), for A/B timing in one process
#define FSEA_CFG_8192_X0 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, 0
#define FSEA_CFG_8192_X7 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, 7
#define FSEA_CFG_4096_X0 4096, 256, 1, 4, 3, 16, 16, 16, 1, true, true, 0, 0
#define FSEA_CFG_2048_X0 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, 0
#define FSEA_CFG_1024_X0 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, 0
