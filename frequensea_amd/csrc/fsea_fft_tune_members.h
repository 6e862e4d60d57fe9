// fsea_fft_tune_members.h -- member functions of fsea::FftKernel that exist in the tuning library only (-DFSEA_TUNE;
// libfsea_hip_tune.so and the CPU emulation of tests/emu): two schedules that were built, are parity-green and lost to the
// product schedule (run_v1), kept buildable for the A/B measurements DESIGN.md quotes.  Included INSIDE struct FftKernel by
// fsea_fft_core.h; never part of a product build (static_assert there: no product configuration names opt::TUNE_ONLY).
//   run_w64 / pixels_w64   opt::W64: 4096 points as 64 x 64 in one wavefront (profiles/r03_w64_and_pixel_epilogue.txt)
//   run_v2 / wave_order    opt::V2: first exchange inside each wavefront, two barriers per frame (profiles/r02_tune_v2_schedule.txt)
    // -----------------------------------------------------------------------------------------
    // W64 schedule (opt::W64): N = 64 x 64, one wavefront per frame, 64 points per lane, ONE exchange
    // through LDS and no s_barrier at all.
    //
    // Sample n = 64 n1 + n2, bin k = k1 + 64 k2.  Pass 0: lane n2 transforms x[64 n1 + n2] over n1 (64 points, constant
    // twiddles) -> y[k1]; exchange: element (k1, n2) goes from lane n2 to lane k1; pass 1: lane k1 transforms over n2 with
    // the twiddles W_N^{n2 k1} deferred into the butterflies (dft_regs_def; 32 register pairs per lane, resident) -> k2.
    //   * (-1)^n = (-1)^{n2} is a shift of the spectrum by N/2 bins = of k2 by 32: nothing is negated, row r of the
    //     last pass is bin k1 + 64 (r ^ 32) -- register renaming.
    //   * Loads.  A lane holding one sample per row would load 2 bytes at a time.  Instead lane L loads the dword
    //     sigma(L) + 64 j (j < 32): 256 contiguous bytes per wave instruction, two adjacent samples (n2 = 2d, 2d + 1) of
    //     row n1 = 2j (+1 in lanes 32-63: sigma(L + 32) = sigma(L) + 32).  Lanes L and L + 32 therefore hold the same
    //     two n2 with complementary n1, and ONE v_permlane32_swap_b32 per pair of rows hands each the other's half:
    //     A = perm(keep / send), B = perm(send / keep), swap(A, B) -> A = rows (2j, 2j'), B = rows (2j + 1, 2j' + 1)
    //     of the lane's own n2, the same registers in every lane.  Three VALU ops per four samples.
    //   * Which n2 a lane ends up with is free (passes meet in LDS at logical addresses); sigma and the kept half are
    //     chosen so that the 16 lanes of every ds_write_b64 lane group hold n2 that differ mod 16 (conflict-free):
    //     n2(L) = 2 pi(L & 31) + ((L >> 5) ^ (L & 1)), pi(t) = (t & 16) + ((t & 15) >> 1) + 8 (t & 1).
    //   * LDS: element (k1, n2) at k1 * 66 + n2 (complex units; 16 bytes of pad per row): 64 ds_write_b64 whose lanes
    //     cover one 512-byte row each, 32 ds_read_b128 of the lane's own row (row pitch 33 x 16 bytes: conflict-free).
    //   * Pixel rows (u8 modes): a lane's 64 pixels are bins k1 + 64 r, one byte each.  Four rows are packed into a dword
    //     by v_cvt_pk_u8_f32 (conversion and packing in one op), transposed 4 x 4 inside each quad of lanes (two
    //     v_mov_b32_dpp quad_perm + two v_perm_b32) and stored as dwords: lane 4m + i writes bins 4m .. 4m+3 of row
    //     4q + i, the wave 256 contiguous bytes per instruction, 16 stores per frame.
    // -----------------------------------------------------------------------------------------
    static __device__ __forceinline__ void run_w64(const FftArgs &a, cf *lds) {
        static_assert(!W64 || (N == 4096 && T == 64 && FPW == 1 && NP == 2 && R0 == 64 && RL == 64), "W64 is the 64 x 64 layout");
        static_assert(!W64 || (IN == IN_U8 && !ROT), "W64 serves the u8 kernels");
        static_assert(!W64 || (!Cfg::TWL && !Cfg::TWR), "W64 keeps its (deferred) twiddles in registers: no table block in LDS");
        constexpr int ROW = 66;  // LDS row pitch in complex units
        const int L = threadIdx.x;
        const unsigned b = blockIdx.x;
        const size_t n_units = a.n_frames;
        const int mode = (MODE_T >= 0) ? MODE_T : a.mode;
        const uint32_t xormask = (MODE_T >= 0) ? 0u : a.xormask;
        const uint32_t esz = elem_bytes(mode);
        const size_t total_in = (size_t)IN_BPS * ((a.n_frames - 1) * a.hop + (size_t)N);
        const bool tiled = a.tile_rows != 0;
        const size_t total_out = (size_t)esz * (tiled ? a.out_span : a.n_frames * (size_t)N);
        auto row_elem = [&](size_t f) -> size_t {
            if (!tiled) return f * (size_t)N;
            const uint32_t k = (uint32_t)f / a.tile_rows, y = (uint32_t)f - k * a.tile_rows;
            return (size_t)y * a.pitch_row + (size_t)k * a.pitch_tile;
        };
        // pass-0 identity of this lane
        const int t5 = L & 31, odd = L & 1, hi = L >> 5;
        const int pi = (t5 & 16) + ((t5 & 15) >> 1) + 8 * odd;
        const int n2 = 2 * pi + (hi ^ odd);
        const uint32_t in_voff = 4u * (uint32_t)(pi + 32 * hi);
        const uint32_t sel_a = odd ? 0x07060302u : 0x05040100u;  // the half (sample) that ends up in A: c = L & 1
        const uint32_t sel_b = odd ? 0x05040100u : 0x07060302u;
        // this lane's deferred twiddles of the last pass: row k1 = L of the [64][32] table (build_deferred_table)
        cf twd[RL / 2];
        ld_c<RL / 2>(a.tw_def + (size_t)L * (RL / 2), twd);

        auto load_frame = [&](size_t u, uint32_t *raw) {
            const rsrc_t rs = buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0);
#pragma unroll
            for (int j = 0; j < 32; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_voff, (uint32_t)(256 * j), LD_AUX);
        };
        size_t u = b;
        uint32_t raw[32];
        load_frame(u, raw);
        // the grid size lives in a VGPR: with the 64-point DFT's constants the SGPR file is full, and a gridDim.x re-read
        // from the dispatch packet (s_load_dword) inside the loop is waited for with lgkmcnt(0) -- together with every
        // LDS write in flight
        unsigned grid_v = gridDim.x;
#if defined(__AMDGCN__)
        asm volatile("" : "+v"(grid_v));
#endif
        cf *const wr = lds + n2;               // + ROW k1
        const cf *const rd = lds + ROW * L;    // + n2

        while (u < n_units) {
            cf v[64];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t wa = byte_perm(raw[j + 16], raw[j], sel_a);
                uint32_t wb = byte_perm(raw[j + 16], raw[j], sel_b);
                lane_swap32(wa, wb);
                wa ^= xormask;
                wb ^= xormask;
                v[2 * j] = cf{s8f(wa, 0), s8f(wa, 1)};
                v[2 * j + 32] = cf{s8f(wa, 2), s8f(wa, 3)};
                v[2 * j + 1] = cf{s8f(wb, 0), s8f(wb, 1)};
                v[2 * j + 33] = cf{s8f(wb, 2), s8f(wb, 3)};
            }
            dft_regs<64, 1, (Cfg::ABL & abl::NO_FLOPS) != 0>(v);
            frame_sync();  // the previous frame's reads precede these writes
            if constexpr ((Cfg::ABL & abl::NO_LDS) == 0) {
#pragma unroll
                for (int k1 = 0; k1 < 64; ++k1) wr[ROW * k1] = v[k1];
            }
            frame_sync();
            const size_t un = u + __builtin_amdgcn_readfirstlane(grid_v);
            if constexpr ((Cfg::ABL & abl::NO_LOADS) == 0) load_frame(un, raw);
            if constexpr ((Cfg::ABL & abl::NO_LDS) == 0) {
                // the first butterflies of pass 1 pair elements j and j + 32: fetched in that order
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ld_c<2>(rd + 2 * q, v + 2 * q);
                    ld_c<2>(rd + 2 * q + 32, v + 2 * q + 32);
                }
            }
            after_reads();
            if constexpr ((Cfg::ABL & abl::NO_FLOPS) == 0) dft_regs_def<64, 1, 1>(v, twd);
            cf w[64];  // centred order: row r = bin L + 64 r
#pragma unroll
            for (int r = 0; r < 64; ++r) w[r] = v[r ^ 32];
            const rsrc_t out = buffer_window(a.out, (size_t)esz * row_elem(u), total_out);
            if (mode == MODE_DB10_U8 || mode == MODE_DB5_U8_DCFIX) pixels_w64(mode, out, w, L);
            else epilogue(mode, out, (uint32_t)L, w, L);
            u = un;
        }
    }

    static __device__ __forceinline__ void pixels_w64(int mode, rsrc_t out, cf *w, int L) {
        const bool patched = (mode == MODE_DB5_U8_DCFIX);
        if (!patched && L == 0) w[32] += cf{128.0f * (float)N, 128.0f * (float)N};  // DC of the offset-binary samples (see epilogue)
        const float kdb = (mode == MODE_DB10_U8 ? 100.0f : 50.0f) * 0.30102999566398120f;
        const float koff = -16.0f * kdb;  // log2 of the 1/256^2 the integer-unit power still carries
        const uint32_t sel1 = (L & 1) ? 0x03070105u : 0x06020400u;
        const uint32_t sel2 = (L & 2) ? 0x03020706u : 0x05040100u;
        const uint32_t voff = (uint32_t)((L & 3) * 64 + (L & ~3));
        uint32_t left = 0;  // DB5: the pixel of bin N/2 - 1 (row 31, lane 63), copied over bin N/2 (row 32, lane 0)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint32_t px = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const cf z = w[4 * q + i];
                const float p = __builtin_fmaf(z[0], z[0], z[1] * z[1]);
                float d = __builtin_fmaf(kdb, __builtin_amdgcn_logf(p), PX_BIAS ? koff - 0.49999997f : koff);
                if constexpr (Cfg::ABL & abl::NO_EPILOGUE_MATH) d = kdb * p;
                px = cvt_pk_u8(PX_BIAS ? d : trunc_f32(d), (uint32_t)i, px);  // truncation toward zero, then saturation: the reference's cast + clamp
            }
            const uint32_t u1 = byte_perm(quad_xor1(px), px, sel1);
            uint32_t rows = byte_perm(quad_xor2(u1), u1, sel2);  // lane 4m + i: bins 4m .. 4m + 3 of row 4q + i
            if (q == 7) left = read_lane(rows, 63) >> 24;
            if (q == 8 && patched && L == 0) rows = (rows & 0xffffff00u) | left;
            if constexpr (Cfg::ABL & abl::NO_STORES) {
                if (rows == 0x01020304u && w[0][0] == -1.0f) __builtin_amdgcn_raw_buffer_store_b32(rows, out, voff, (uint32_t)(256 * q), ST_AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(rows, out, voff, (uint32_t)(256 * q), ST_AUX);
            }
        }
    }

    // -----------------------------------------------------------------------------------------
    // V2 schedule for frames spread over several wavefronts (three passes RA x RB x RC).
    //
    // Sample index n = a N/RA + b RC + c, bin k = ka + RA kb + RA RB kc:
    //   pass 0 sums over a (-> ka), twiddle W_{RA RB}^{b ka}, pass 1 over b (-> kb), twiddle
    //   W_N^{c (ka + RA kb)}, pass 2 over c (-> kc).
    // Only the bits a lane holds in registers are forced (a, b, c in turn); which of the other
    // bits are lane bits and which are wave bits is free.  V1 takes them in Stockham order, which
    // makes both exchanges cross-wave: two s_barriers each.  Here the wave bits of passes 0 and 1
    // are the top bits of c, so the first exchange (ka <-> b) stays inside a wavefront -- LDS
    // operations of one wave execute in order, no barrier -- and only the second one crosses
    // waves.  Every element of that second exchange is read by exactly one wave, so the buffer
    // is a partition S_w by reading wave; once wave w has read its S_w nobody else touches it
    // until the next cross-wave write, and w runs its own first exchange of the next frame in
    // it.  Per frame: [pass 0] A-write A-read [pass 1] BARRIER B-write BARRIER B-read [pass 2]:
    // two barriers instead of four, one rendezvous per frame.
    //   * Pass-0 loads: a wave reads 16-byte pieces at 64-byte stride (its c bits), the four
    //     waves of the workgroup cover the lines between them (L1 hits); stores keep 256-byte runs.
    //   * Middle-pass twiddles W_{RA RB}^{b ka} depend on the lane (ka) only: deferred into the
    //     butterflies (dft_regs_def) they are RB/2 register pairs per lane, resident for the
    //     workgroup's lifetime; no twiddle is read from LDS per frame.
    // LDS slots are 16 bytes (two complex, the c0 pair):
    //   A (inside S_w): slot = 65 ka + 16 g + b      writer lane (b, g), reader lane (ka, g)
    //   B:              slot = 17 m + j              m = ka + RA kb, j = c / 2; S_w = rows 64 w ..
    // both conflict-free for ds_write_b128 (8 consecutive lanes -> 8 consecutive slots mod 8) and
    // ds_read_b128 (a 16-lane group -> 16 distinct slots mod 16; 65 and 17 are odd).
    // -----------------------------------------------------------------------------------------
    static __device__ __forceinline__ void wave_order() {
        // LDS operations of one wavefront execute in program order; this only stops the compiler
        // from moving them across (and keeps the CPU emulation's lanes together)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    static __device__ __forceinline__ void run_v2(const FftArgs &a, cf *lds_all) {
        constexpr int RA = Cfg::R(0), RB = Cfg::R(1), RC = Cfg::R(2);
        static_assert(NP == 3 && FPW == 1 && (T % 64) == 0 && !ONE_WAVE, "V2 is for multi-wave frames in three passes");
        static_assert(RA == 16 && RB == 16 && RC == 32 && P == 32, "V2 layout constants are written for 16 x 16 x 32");
        static_assert(Cfg::TWR, "V2 keeps the last pass's twiddles in registers");
        constexpr int G = 64 / RB;            // lane groups per wave (the c bits below the wave bits, above c0)
        constexpr int ROW_B = RC / 2 + 1;     // 16-byte slots per B row (m), odd
        constexpr int SW = 64 * ROW_B;        // slots of one wave's partition S_w
        constexpr int ROW_A = 4 * RB + 1;     // slots between consecutive ka in the A layout, odd
        static_assert((RA - 1) * ROW_A + 16 * (G - 1) + RB <= SW, "the A layout must fit the wave's partition");
        static_assert(2 * SW * (T / 64) <= Cfg::LDS_FRAME, "LDS frame too small for the B layout");

        const int tid = threadIdx.x;
        const int w = tid >> 6, l = tid & 63;
        unsigned *tk = reinterpret_cast<unsigned *>(lds_all + Cfg::LDS_TOTAL);
        cf *lds = lds_all;

        const unsigned bidx = blockIdx.x;
        const bool issuer = (tid == 0);
        const size_t n_units = a.n_frames;
        Pools pools;
        pools.n_units = (unsigned)n_units;
        pools.grid = gridDim.x;
        unsigned cur = bidx % POOLS;

        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            const unsigned hw_id = read_hw_id(), xcc_id = read_xcc_id();
            a.trace[32 * bidx + 0] = wall_clock64();
            a.trace[32 * bidx + 2] = __builtin_readcyclecounter();
            a.trace[32 * bidx + 4] = hw_id;
            a.trace[32 * bidx + 5] = xcc_id;
        }

        const int mode = (MODE_T >= 0) ? MODE_T : a.mode;
        const uint32_t xormask = (MODE_T >= 0) ? 0u : a.xormask;
        const uint32_t esz = elem_bytes(mode);
        const size_t total_in = (size_t)IN_BPS * ((a.n_frames - 1) * a.hop + (size_t)N);
        const bool tiled = a.tile_rows != 0;
        const size_t total_out = (size_t)esz * (tiled ? a.out_span : a.n_frames * (size_t)N);
        auto row_elem = [&](size_t f) -> size_t {  // element offset of frame f's row (FPW == 1 here)
            if (!tiled) return f * (size_t)N;
            const uint32_t k = (uint32_t)f / a.tile_rows, y = (uint32_t)f - k * a.tile_rows;
            return (size_t)y * a.pitch_row + (size_t)k * a.pitch_tile;
        };
        // pass 0: lane (b, g) of wave w holds samples n = a N/RA + b RC + (w G + g) C0 + c0
        const int b0 = l % RB, g = l / RB;
        const int n_base = b0 * RC + (w * G + g) * C0;
        // ABL 16 (measurement only, wrong results): the V1 load mapping, 256-byte runs per wave
        const uint32_t in_voff = (uint32_t)IN_BPS * (uint32_t)((Cfg::ABL & abl::V2_V1_LOADS) ? C0 * tid : n_base);
        // ABL 8 (measurement only): static unit interleave, next unit's bytes requested right after the
        // conversion of this one's
        constexpr bool STATIC = (Cfg::ABL & abl::V2_STATIC) != 0;
        // pass 1: lane (ka, g); pass 2: thread m = ka + RA kb = tid
        const int ka = l % RA;
        const int m = tid;
        const uint32_t out_elem = (uint32_t)m;

        size_t u = (size_t)pools.start(cur) + bidx / POOLS;
        if (u >= pools.start(cur + 1)) u = n_units;
        if constexpr (STATIC) u = bidx;
        unsigned tick_next = 0;

        constexpr int TAB_COPY = Cfg::TAB_SMALL;
        constexpr int TAB_REGS = (TAB_COPY + Cfg::WG - 1) / Cfg::WG;
        cf tabv[TAB_REGS];
#pragma unroll
        for (int i = 0; i < TAB_REGS; ++i) {
            const int e = tid + i * Cfg::WG;
            tabv[i] = a.tw_small[e < TAB_COPY ? e : TAB_COPY - 1];
        }
        // this lane's deferred middle-pass twiddles: RB/2 pairs, resident
        cf tw1[RB / 2];
        ld_c<RB / 2>(a.tw_def + ka * (RB / 2), tw1);
        Raw raw[R0];
        load_raw(buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        if (!STATIC && issuer) tick_next = atomicAdd(a.ctr + 32 * cur, 1u);

#pragma unroll
        for (int i = 0; i < TAB_REGS; ++i) {
            const int e = tid + i * Cfg::WG;
            lds_all[Cfg::LDS_FRAME + (e < TAB_COPY ? e : TAB_COPY - 1)] = tabv[i];
        }
        __syncthreads();
        cf twl[(RL - 1) * CL];
        {
            const cf *hi = lds_all + Cfg::LDS_HI, *lo = lds_all + Cfg::LDS_LO;
#pragma unroll
            for (int r = 1; r < RL; ++r) {
                const unsigned e = (unsigned)r * (unsigned)m;
                cf tw = pk_cmul(hi[e >> 6], lo[e & 63u]);
                if constexpr (PRESCALED) tw = tw * cf{SC, SC};
                twl[r - 1] = tw;
            }
        }

        cf ebase[ROT ? C0 : 1];
        if constexpr (ROT) {
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                const unsigned n0 = (unsigned)(n_base + c);
                const cf e = turn_phasor_f64(a.rot_delta * (double)n0);
                ebase[c] = (n0 & 1u) ? -e : e;
            }
        }

        unsigned par = 0;
        if (!STATIC && u >= n_units) {
            if (issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[0] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }
            __syncthreads();
            const unsigned nu = __builtin_amdgcn_readfirstlane(tk[0]);
            __syncthreads();
            u = (nu == NO_UNIT) ? n_units : (size_t)nu;
            load_raw(buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        }

        // LDS addresses (complex units; a slot is two complex)
        cf *const sw = lds + 2 * SW * w;                                   // this wave's partition
        cf *const a_wr = sw + 2 * (16 * g + b0);                           // + 2 ROW_A ka
        const cf *const a_rd = sw + 2 * (ROW_A * ka + 16 * g);             // + 2 b
        cf *const b_wr = lds + 2 * (ROW_B * ka + (w * G + g));             // + 2 ROW_B RA kb
        const cf *const b_rd = lds + 2 * ROW_B * m;                        // + 2 j

        unsigned iter = 0;
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) a.trace[32 * bidx + 6] = wall_clock64();
        while (u < n_units) {
            if (!STATIC && issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[par] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }

            cf v[P];
            if constexpr (ROT) {
                const size_t first = u * a.hop;
                const cf ef = turn_phasor_f32(a.rot_phase0 + a.rot_delta * (double)first);
                cf fr[C0];
#pragma unroll
                for (int c = 0; c < C0; ++c) fr[c] = pk_cmul(ebase[c], ef);
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    convert_row<IN, C0, false>(raw[r], a.xormask, n_base, v + r * C0);
                    const cf wr = a.rot_row[r];
#pragma unroll
                    for (int c = 0; c < C0; ++c) {
                        v[r * C0 + c] = pk_cmul(v[r * C0 + c] + cf{128.0f, 128.0f}, pk_cmul_uniform(fr[c], wr));
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R0; ++r) convert_row<IN, C0>(raw[r], xormask, n_base, v + r * C0);
            }
            size_t un = u + gridDim.x;
            if constexpr (STATIC) {
                load_raw(buffer_window(a.in, (size_t)IN_BPS * un * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
#pragma unroll
            for (int c = 0; c < C0; ++c) dft_regs<R0, C0>(v + c);

            // exchange A, inside this wave's partition: ka <-> b
            if constexpr ((Cfg::ABL & abl::V2_NO_EXCHANGE_A) == 0) {  // ABL 32 (measurement only): no exchange A
                wave_order();  // this wave's reads of the previous frame (B) precede these writes
#pragma unroll
                for (int r = 0; r < RA; ++r) st_c<2>(a_wr + 2 * ROW_A * r, v + 2 * r);
                wave_order();
#pragma unroll
                for (int r = 0; r < RB; ++r) ld_c<2>(a_rd + 2 * r, v + 2 * r);
                after_reads();
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) dft_regs_def<RB, 2, 1>(v + c, tw1);

            __syncthreads();  // every wave has read its partition: the cross-wave writes may land
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter == 0) a.trace[32 * bidx + 7] = wall_clock64();
            // exchange B, across waves: row m = ka + RA kb gets this lane's c pair at j = w G + g
#pragma unroll
            for (int r = 0; r < RB; ++r) st_c<2>(b_wr + 2 * ROW_B * RA * r, v + 2 * r);
            __syncthreads();
            // nothing but the writes sits between the two barriers; the ticket word is read in
            // front of the data (LDS returns in order), and the next unit's bytes are requested
            // while the rows come in
            const unsigned tkv = STATIC ? 0u : tk[par];
#pragma unroll
            for (int j = 0; j < RC / 2; ++j) ld_c<2>(b_rd + 2 * j, v + 2 * j);
            if constexpr (!STATIC) {
                __builtin_amdgcn_sched_barrier(0);  // the reads are issued before the ticket is waited for
                const unsigned nu = __builtin_amdgcn_readfirstlane(tkv);
                par ^= 1u;
                un = (nu == NO_UNIT) ? n_units : (size_t)nu;
                load_raw(buffer_window(a.in, (size_t)IN_BPS * un * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
            after_reads();

            if constexpr (TW_FUSE) {
                dft_regs_tw<RL, 1, 1>(v, twl, PRESCALED ? SC : 1.0f);
            } else {
                if constexpr (PRESCALED) v[0] = v[0] * cf{SC, SC};
#pragma unroll
                for (int r = 1; r < RL; ++r) v[r] = pk_cmul(v[r], twl[r - 1]);
                dft_regs<RL, 1>(v);
            }
            epilogue(mode, buffer_window(a.out, (size_t)esz * row_elem(u), total_out), out_elem, v, m);
            u = un;
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter < 24) a.trace[32 * bidx + 8 + iter] = wall_clock64();
            ++iter;
        }

        if (!STATIC && issuer) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned finished = atomicAdd(a.ctr + 32 * POOLS, 1u);
            if (finished == gridDim.x - 1) {
#pragma unroll
                for (unsigned q = 0; q <= POOLS; ++q) a.ctr[32 * q] = 0;
            }
        }
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            a.trace[32 * bidx + 1] = wall_clock64();
            a.trace[32 * bidx + 3] = __builtin_readcyclecounter();
        }
    }

