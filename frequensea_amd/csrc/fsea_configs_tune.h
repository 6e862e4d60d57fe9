// fsea_configs_tune.h -- tuning variants and measurement-only ablations, compiled into
// libfsea_hip_tune.so only (fsea_plan_create_variant, include/fsea_tune.h); never product defaults.
#pragma once

#include "fsea_configs.h"

// packed-add +-i butterflies without the deferred twiddles
#define FSEA_CFG_8192_ND 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_16384_ND 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, fo::TW_FUSE
// deferred twiddles on the sizes that did not gain from them
#define FSEA_CFG_4096_DF 4096, 256, 1, 4, 3, 16, 16, 16, 1, true, true, 0, fo::DEFER | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_2048_DF 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, fo::DEFER | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// cache policy of the input loads / output stores (fo::ST_NT nt stores, fo::LD_NT nt loads; the sc0 / sc1 store variants were removed in round 4):
// "cp0" = default policy for both (the round-2 kernel before the policy was chosen), then the alternatives
#define FSEA_CFG_8192_CP0 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_STNT 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_LDNT 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_16384_CP0 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, fo::DEFER | fo::TW_FUSE
#define FSEA_CFG_16384_STNT 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, fo::ST_NT | fo::DEFER | fo::TW_FUSE
#define FSEA_CFG_4096_CP0 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_STNT 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_2048_CP0 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_2048_STNT 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, fo::ST_NT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_1024_CP0 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_1024_LDSTNT 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_256_CP0 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, 0
#define FSEA_CFG_256_LDSTNT 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, fo::LD_NT | fo::ST_NT
// V2 schedule (fo::V2): first exchange inside each wavefront, two barriers per frame; with its
// measurement-only ablations (8: static units + early prefetch, 16: V1 load mapping, wrong results,
// 32: no first exchange, wrong results)
#define FSEA_CFG_8192_V2 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::V2 | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_8192_V2S 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::V2_STATIC, fo::V2 | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_8192_V2L 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::V2_V1_LOADS, fo::V2 | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_8192_V2SL 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::V2_V1_LOADS | fsea::abl::V2_STATIC, fo::V2 | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_8192_V2NA 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::V2_NO_EXCHANGE_A, fo::V2 | fo::TW_FUSE | fo::BATCH_READS
// other pass orders with round 2's options: 16 x 32 x 16 (8-byte row stores), 8 x 32 x 32 (8-byte loads)
#define FSEA_CFG_8192_B2 8192, 256, 1, 2, 3, 16, 32, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_D2 8192, 256, 1, 2, 3, 8, 32, 32, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// 32 x 32 x 8: four adjacent bins per lane in the last pass (16-byte row stores), 2-byte pass-0 loads; with and without nt loads
// issue priority between the two workgroups of a CU: alternating per frame (65536), catching up with the pool's average (131072)
// the younger workgroup of every CU (block >= grid / 2) at a constant higher priority (the older one wins the arbitration
// otherwise: 39 vs 53 us for the same 8 frames), levels 1, 2, 3
// the last pass's register twiddles each gathered directly from the two factor tables (the form up to mid round 2)
// static unit interleave (unit = blockIdx + k * grid) instead of the ticket pools
#define FSEA_CFG_8192_W 8192, 256, 1, 2, 3, 32, 32, 8, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_W2 8192, 256, 1, 2, 3, 32, 32, 8, 1, true, true, 0, fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// other pass orders / twiddle sources
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
// small sizes with 16 points per lane (2-byte pass-0 loads), as in round 1 and round 2a
#define FSEA_CFG_256_P16 256, 16, 16, 2, 2, 16, 16, 1, 1, true, true, 0, fo::ST_NT
#define FSEA_CFG_128_P16 128, 8, 32, 2, 2, 16, 8, 1, 1, true, true, 0, fo::ST_NT
// 4096 as in rounds 1 and 2a: 256 lanes x 16 points, four workgroups per CU (with and without the streaming policy);
// "B" is the 128-lane layout without round 2's options, "B3" the product layout without deferred twiddles
#define FSEA_CFG_4096_T256 4096, 256, 1, 4, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_4096_F1 4096, 128, 1, 2, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_B3 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// lane rotation against LDS read conflicts (fo::LANE_ROT = middle pass, fo::LANE_ROT_LAST = last pass) switched OFF where the product has it:
// 4096 without the last-pass rotation, 2048 without the middle-pass rotation (scripts/lds_conflicts.py predicts 2 cycles
// per ds_read_b128 group; measured SQ_LDS_BANK_CONFLICT 4.3 M / 8.5 M cycles per launch against 0.1 M / 4.3 M with it)
#define FSEA_CFG_4096_LR 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS   /* "nr": without the last-pass rotation */
#define FSEA_CFG_2048_LR 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS      /* "nr": without the middle-pass rotation */
#define FSEA_CFG_16384_B 16384, 512, 1, 2, 3, 32, 32, 16, 1, true, true
#define FSEA_CFG_2048_B 2048, 64, 4, 2, 3, 8, 8, 32, 1, true, true
#define FSEA_CFG_2048_C 2048, 64, 4, 2, 3, 4, 16, 32, 1, true, true
// measurement-only ablations of the 8192-point kernel (results are wrong by design):
// 1 = no output stores, 2 = no LDS exchange / barriers, 4 = no butterflies, 64 = no per-frame loads,
// 128 = no magnitude arithmetic
#define FSEA_CFG_8192_NOST 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_STORES, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOLDS 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_LDS, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOFLOP 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_FLOPS, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_IO 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_FLOPS | fsea::abl::NO_LDS, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_VALU 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_LDS | fsea::abl::NO_STORES, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_IONT 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_FLOPS | fsea::abl::NO_LDS, fo::LD_NT | fo::ST_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOLDSNT 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_LDS, fo::LD_NT | fo::ST_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOFLOPNT 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_FLOPS, fo::LD_NT | fo::ST_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOLOAD 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_LOADS, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// 256 = the row stored 16 bytes per lane (bins misplaced), 512 = the frame loaded 16 bytes per lane (samples
// misplaced): what layouts with 4 adjacent bins / 8 adjacent samples per lane would issue, without their other costs
// the small sizes without nt stores, as full kernel sets: the pixel kernels' 16- and 32-byte row pieces are merged in L2
// only when the stores are allowed to stay there
#define FSEA_CFG_128_ST0 128, 4, 64, 2, 2, 16, 8, 1, 1, true, true, 0, 0
#define FSEA_CFG_256_ST0 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, 0
#define FSEA_CFG_512_ST0 512, 16, 16, 2, 2, 32, 16, 1, 1, true, true, 0, 0
#define FSEA_CFG_1024_ST0 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, fo::TW_FUSE | fo::BATCH_READS
// pixel-mode ablations (full kernel sets, so that the compile-time DB5 / DB10 kernels exist): 128 = no logarithm,
// 1 = no pixel stores, 256 = four pixels per dword store (misplaced), 6 = loads + epilogue only
#define FSEA_CFG_4096_PXNOLOG 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, fsea::abl::NO_EPILOGUE_MATH, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_PXNOST 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, fsea::abl::NO_STORES, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_PXIO 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, fsea::abl::NO_FLOPS | fsea::abl::NO_LDS, fo::LD_NT | fo::ST_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_PXNOLOG 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_EPILOGUE_MATH, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_NOMAG 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::NO_EPILOGUE_MATH, fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// schedule options of the defaults switched off (FftCfg::OPT), for A/B timing in one process
#define FSEA_CFG_8192_X0 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, 0
#define FSEA_CFG_4096_X0 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, 0
#define FSEA_CFG_2048_X0 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, 0
#define FSEA_CFG_1024_X0 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, 0
// 4096 points as ONE wavefront per frame, 64 lanes x 64 points, no s_barrier (round 3):
// "w64": 64 x 64, one exchange (FftKernel::run_w64; fo::W64), last-pass twiddles deferred and register-resident;
// "s2": 16 x 16 x 16 in the V1 schedule, two exchanges, dwordx2 loads and four adjacent bins per lane in the last pass
#define FSEA_CFG_4096_W64 4096, 64, 1, 1, 2, 64, 64, 1, 1, false, false, 0, fo::W64 | fo::LD_NT | fo::ST_NT | fo::BATCH_READS
#define FSEA_CFG_4096_S2 4096, 64, 1, 1, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_W64B 4096, 64, 1, 1, 2, 64, 64, 1, 1, false, false, 0, fo::PX_BIAS | fo::W64 | fo::LD_NT | fo::ST_NT | fo::BATCH_READS   /* + biased rounding instead of v_trunc */
// pixel epilogue with v_cvt_pk_u8_f32 (fo::PX_PACK: v_trunc + convert-and-pack; + fo::PX_BIAS: biased rounding, no v_trunc)
// "pk": with the v_trunc (exact truncation); "px0": the round-2 form (cast, clamp, shift/or); the product has both bits
#define FSEA_CFG_4096_PK 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::PX_PACK | fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT_LAST | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_PX0 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT_LAST | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_PK 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::PX_PACK | fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_PX0 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::LD_NT | fo::ST_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_256_PK 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, fo::PX_PACK | fo::ST_NT
#define FSEA_CFG_256_PX0 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, fo::ST_NT
#define FSEA_CFG_1024_PX0 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, fo::ST_NT | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_1024_R2 FSEA_CFG_1024_RT   /* the round-2 layout (32 x 32) with round 3's pixel epilogue */
// 256 points with 64 points per lane (4 lanes per frame, 16 x 16 with four columns: dwordx2 loads, four adjacent bins per
// lane in the last pass), one wave per workgroup, one wave per SIMD
#define FSEA_CFG_256_P64 256, 4, 16, 1, 2, 16, 16, 1, 1, true, true, 0, fo::ST_NT
// pixel kernels at higher occupancy / finer workgroups (round 3; full u8 kernel sets with the product's pixel epilogue):
// "t256px": 256 lanes x 16 points, four workgroups per CU = 4 waves per SIMD (round 1's layout: 2-byte loads, 1-byte stores);
// "f1px": the product layout with one frame per workgroup (two waves, four workgroups per CU)
#define FSEA_CFG_4096_T256PX 4096, 256, 1, 4, 3, 16, 16, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_4096_F1PX 4096, 128, 1, 2, 3, 16, 16, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// 32 and 64 points with two adjacent samples per lane in pass 0 (dword loads instead of 2-byte loads): 4 x 8 and 8 x 8
#define FSEA_CFG_32_C2 32, 4, 64, 2, 2, 4, 8, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_64_C2 64, 4, 64, 2, 2, 8, 8, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_64_T2 64, 2, 128, 2, 2, 8, 8, 1, 1, true, true, 0, fo::STREAMING_PIXELS
// 1024 points in three passes with dword / dwordx2 pass-0 loads (the product's 32 x 32 loads 2 bytes per lane and row):
// "e" = 16 x 8 x 8, "f" = 8 x 16 x 8, "g" = 16 x 16 x 4; 32 lanes x 32 points, no barrier (single-wave frames), two exchanges
#define FSEA_CFG_1024_E 1024, 32, 8, 2, 3, 16, 8, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_1024_F 1024, 32, 8, 2, 3, 8, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_1024_G 1024, 32, 8, 2, 3, 16, 16, 4, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_1024_H 1024, 32, 8, 2, 3, 8, 8, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_1024_FD 1024, 32, 8, 2, 3, 8, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS   /* f + deferred middle-pass twiddles */
#define FSEA_CFG_1024_F0 1024, 32, 8, 2, 3, 8, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS   /* f without the middle-pass lane rotation (fo::LANE_ROT) */
#define FSEA_CFG_1024_FL 1024, 32, 8, 2, 3, 8, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS   /* f without nt loads */
// Round 3, after 1024 gained from wider pass-0 loads and four bins per lane: the same question at the other sizes
// (full u8 kernel sets with the product's options; names = the radix order)
#define FSEA_CFG_512_888 512, 16, 16, 2, 3, 8, 8, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_512_1632 FSEA_CFG_512_PX       /* product for the pixel modes since round 4 */
#define FSEA_CFG_256_488 FSEA_CFG_256_ROWS     /* product for the f32-row modes since round 4 */
#define FSEA_CFG_256_884 256, 8, 32, 2, 3, 8, 8, 4, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_128_448 128, 4, 64, 2, 3, 4, 4, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_2048_81616 2048, 64, 4, 2, 3, 8, 16, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_16328 4096, 128, 2, 2, 3, 16, 32, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT_LAST | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_4096_83216 4096, 128, 2, 2, 3, 8, 32, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT_LAST | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_163216 8192, 256, 1, 2, 3, 16, 32, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_83232 8192, 256, 1, 2, 3, 8, 32, 32, 1, true, true, 0, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// "f8" = the product configuration + fo::TW_FUSE at the small sizes (what round 3's power-form experiment, removed in round 4, built on)
#define FSEA_CFG_512_F8 512, 16, 16, 2, 2, 32, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::TW_FUSE
#define FSEA_CFG_256_F8 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::TW_FUSE
// 32 / 64 points with two lanes per frame (16 / 32 points per lane): radix orders by pass-0 load width and bins per lane
//   32: "t2a" 4 x 8 (8-byte loads, 2 bins per lane), "t2b" 8 x 4 (dword loads, 4 bins), "t2c" 2 x 16 (16-byte loads, 1 bin)
//   64: "t2c" 16 x 4 (dword loads, 8 bins per lane), "t2d" 4 x 16 (16-byte loads, 2 bins)
#define FSEA_CFG_32_T2A 32, 2, 128, 2, 2, 4, 8, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_32_T2B 32, 2, 128, 2, 2, 8, 4, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_32_T2C 32, 2, 128, 2, 2, 2, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_64_T2C 64, 2, 128, 2, 2, 16, 4, 1, 1, true, true, 0, fo::STREAMING_PIXELS
#define FSEA_CFG_64_T2D 64, 2, 128, 2, 2, 4, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS
// measurement only (wrong rows): the product configuration with 16 / 32 packed ops per lane-frame left out of the middle pass --
// what a radix-4 regrouping of the in-register DFTs could save at most, as a rate (profiles/r04_radix4_rejected.txt)
#define FSEA_CFG_8192_M16 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::DROP_16_OPS, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192_M32 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, fsea::abl::DROP_32_OPS, fo::STREAMING_PIXELS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
