// 1 x 1 convolutions (plain GEMMs) on the round-6 main loop: persistent workgroups, ONE wave per SIMD on a 128-pixel x 128-channel
// accumulator block (256 AGPRs), the K loop laid out instruction by instruction exactly as in conv_lstm_w128.h (which explains the
// method and carries the measurements).  Included by conv_fwd.hip inside its anonymous namespace, after conv_lstm_w128.h (shares its
// helper macros and types).  models/_resnet.py:96-114 (the frozen teacher's conv1 / conv3 / downsample layers: bias-free conv
// followed by train-mode BatchNorm).
//
// What differs from the ConvLSTM kernel:
//  * operands: a K-slab is 256 pixel rows x 64 channels (32 KB, streamed once from HBM) + 256 weight rows x 64 (32 KB, L2 hits).
//    The pixel operand runs in a THREE-deep ring (its pieces are issued 2.5 slabs = ~6 k cycles ahead: HBM latency), the weights in
//    a two-deep one: 3 x 32 + 2 x 32 = 160 KB = all of the CU's LDS.  Slab s reads pixel stage s % 3 and weight stage s % 2; behind
//    the slab's barrier, group 3 issues weight slab s + 2 and group 0 of the next slab pixel slab s + 3; the barrier's wait is
//    vmcnt(8): the eight youngest pieces (pixel slab s + 3) stay in flight.  The K loop is unrolled six-fold (lcm of the rings) so
//    that every stage is an immediate offset.
//  * epilogue = what a conv in front of a BatchNorm needs: the raw result rounded to bf16, stored as 16-byte pieces (two
//    v_permlane32_swap per pair of channel quadruples give a lane eight consecutive channels of one pixel), and the per-128-row
//    tile statistics (sum, sum of squares of the STORED values) that engine._ConvBNTrainFn / norm_ops consume: lane-local sums
//    over the wave's four pixel blocks, then a 32-lane DPP reduction per channel (fixed order: bit-repeatable).
//  * tiles are dealt statically: workgroup b (XCD b % 8) walks every 32nd tile of its XCD's contiguous chunk, n-tiles of one
//    m-tile adjacent (the pixel rows are then read once from HBM and again from that XCD's L2).
// Takes: R = S = 1, stride 1, Cin % 64 == 0, Cout % 256 == 0, no bias / activation / residual, bf16 output.
// one DPP reduction step on four independent registers (operands %0..%3 of the asm statement)
#ifndef G128_ABL
#define G128_ABL 0          // measurement builds: 1 = no output stores, 2 = no statistics, 4 = no DPP reduction, 8 = no epilogue at all
#endif
#ifndef G128_AUX
#define G128_AUX 0          // cache policy of the result stores (gfx950 buffer aux: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#define G128_DPP4(CTRL) "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
constexpr int G128_STAGE = 256 * 128;                     // 32 768 bytes: one operand stage
constexpr int G128_LDS = 5 * G128_STAGE;                  // 163 840: [pixels 0][pixels 1][weights 0][weights 1][pixels 2] (ds_read immediates are 16 bits)
constexpr int g128_pstage_byte(int st) { return st == 2 ? 4 * G128_STAGE : st * G128_STAGE; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv1x1_w128_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    constexpr int MT = 4, NT = 4, P_INSTR = 8, B_INSTR = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int p31 = lane & 31, hi = lane >> 5;
    const int lrow = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const unsigned oob = 0x80000000u;

    int Cin_s = a.Cin, M_s = a.M, ips_s = (int)a.in_pix_stride, ops_s = (int)a.out_pix_stride, Cout_s = a.Cout, tiles_n = a.tiles_n;
    asm volatile("" : "+s"(Cin_s), "+s"(M_s), "+s"(ips_s), "+s"(ops_s), "+s"(Cout_s), "+s"(tiles_n));
    const int nslab = Cin_s >> 6;
    const long long in_bytes = (((long long)M_s - 1) * ips_s + Cin_s) * 2;
    const long long out_bytes = (((long long)M_s - 1) * ops_s + Cout_s) * 2;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)out_bytes, 0x00020000);
    const int stat_rows = (M_s + 127) >> 7;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.stats, 0, a.stats ? stat_rows * 2 * Cout_s * 4 : 0, 0x00020000);
    const bool with_stats = a.stats != nullptr && !(G128_ABL & 2);
    const bool nt_out = a.ksplit != 0;                    // host rule: Cout >= 4 Cin
    const bool with_bias = a.bias != nullptr, with_relu = a.relu == 1;

    // fragment addresses: k-step ks of row r reads 16-byte chunk (2 ks + half) ^ ((r >> 1) & 7) of its 128-byte LDS row
    uint32_t pa[MT], pa2[MT], wa[NT][4];                  // pa: pixel stages 0 / 1 (+ immediate), pa2: pixel stage 2
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 128 + i * 32 + p31;
        pa[i] = lds0 + (uint32_t)(r * 128) + ((((uint32_t)hi) ^ (uint32_t)((r >> 1) & 7)) << 4);
        pa2[i] = pa[i] + 4 * G128_STAGE;
        asm volatile("" : "+v"(pa[i]), "+v"(pa2[i]));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * 128 + j * 32 + p31;
        const uint32_t sw = (uint32_t)((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wa[j][ks] = lds0 + (uint32_t)(2 * G128_STAGE + r * 128) + ((((uint32_t)(ks * 2) + (uint32_t)hi) ^ sw) << 4);
            asm volatile("" : "+v"(wa[j][ks]));
        }
    }

    f32x16_t acc[16];
    bf16x8_t fp[2][MT], fw[2][NT];
    unsigned aoff[P_INSTR];                               // pixel pieces of this tile (byte offset of the lane's 16 bytes at k = 0; OOB: row >= M)
    int boff[B_INSTR];
    int ovoff[MT];                                        // output byte offset of pixel block i (OOB: row >= M)
    int n0 = 0, trow = 0;

    auto setup = [&](int bid) __attribute__((always_inline)) {
        int lane_o = lane, wave_o = wave;
        asm volatile("" : "+v"(lane_o), "+s"(wave_o));    // per-tile values are recomputed, not hoisted and spilled (conv_lstm_w128.h)
        const int lrow_ = lane_o >> 3, slot_ = lane_o & 7, p31_ = lane_o & 31, hi_ = lane_o >> 5;
        const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
        const int m0 = tile_m * 256;
        n0 = tile_n * 256; trow = tile_m * 2 + (wave_o >> 1);
#pragma unroll
        for (int i = 0; i < P_INSTR; ++i) {
            const int r = (wave_o * P_INSTR + i) * 8 + lrow_;
            const int m = m0 + r;
            aoff[i] = m < M_s ? (unsigned)(m * ips_s * 2 + (slot_ ^ ((r >> 1) & 7)) * 16) : oob;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int r = (wave_o * B_INSTR + i) * 8 + lrow_;
            boff[i] = ((n0 + r) * a.Kpad + (slot_ ^ ((r >> 1) & 7)) * 8) * 2;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + (wave_o >> 1) * 128 + i * 32 + p31_;
            ovoff[i] = m < M_s ? (m * ops_s + n0 + (wave_o & 1) * 128 + 8 * hi_) * 2 : (int)oob;
        }
    };
    // `live` = the slab exists (slabs past the tile's end: every lane out of range, no traffic)
    auto p_piece = [&](auto st_c, auto i_c, int koff, bool live) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value, i = decltype(i_c)::value;
        const unsigned av = live ? aoff[i] : oob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + g128_pstage_byte(st) + (wave * P_INSTR + i) * 1024),
                                                 16, av, koff, 0, 0);
    };
    auto w_piece = [&](auto st_c, auto i_c, int koff, bool live) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value, i = decltype(i_c)::value;
        const unsigned bv = live ? (unsigned)boff[i] : oob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + (2 + st) * G128_STAGE + (wave * B_INSTR + i) * 1024),
                                                 16, bv, koff, 0, 0);
    };

#define G128_RD_P(BUF, I, KS, PA, OFF) { uint32_t t_; asm volatile("v_xor_b32 %1, %4, %2\n\tds_read_b128 %0, %1 offset:%3" : "=v"(fp[BUF][I]), "=&v"(t_) : "v"(PA[I]), "n"(OFF), "n"((KS) << 5) : "memory"); }
#define G128_RD_P0(BUF, I, PA, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fp[BUF][I]) : "v"(PA[I]), "n"(OFF) : "memory")
#define G128_RD_W(BUF, J, KS, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[BUF][J]) : "v"(wa[J][KS]), "n"(OFF) : "memory")
    // PST = pixel stage (0..2), WOFF = byte offset of the weight stage
    auto frag_read = [&fp, &fw, &pa, &pa2, &wa](auto buf_c, auto q_c, auto ks_c, auto pst_c, auto woff_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, q = decltype(q_c)::value, KS = decltype(ks_c)::value;
        constexpr int PST = decltype(pst_c)::value, WOFF = decltype(woff_c)::value, POFF = PST == 2 ? 0 : PST * G128_STAGE;
        if constexpr (q == 0) G128_RD_W(BUF, 0, KS, WOFF);
        else if constexpr (q <= 4) {
            if constexpr (PST == 2) { if constexpr (KS == 0) G128_RD_P0(BUF, q - 1, pa2, POFF); else G128_RD_P(BUF, q - 1, KS, pa2, POFF); }
            else { if constexpr (KS == 0) G128_RD_P0(BUF, q - 1, pa, POFF); else G128_RD_P(BUF, q - 1, KS, pa, POFF); }
        }
        else G128_RD_W(BUF, q - 4, KS, WOFF);
    };
    auto mma = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };
    auto mma_first = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };

    // slab s = S6 (mod 6): pixel stage S6 % 3, weight stage S6 % 2
    auto slab = [&](auto s6_c, int s) __attribute__((always_inline)) {
        constexpr int S6 = decltype(s6_c)::value;
        constexpr int PST = S6 % 3, WST = S6 % 2, nPST = (S6 + 1) % 3, nWST = (S6 + 1) % 2;
        constexpr int WOFF = WST * G128_STAGE, nWOFF = nWST * G128_STAGE;
        using cP = w128_c<PST>; using cW = w128_c<WOFF>;
        const int k2 = (s + 2) * 128;                      // byte offset along K of slab s + 2
        const bool live2 = s + 2 < nslab;
        // G0: buffer 0; reads of k-step 1; pixel slab s + 2 into stage (s + 2) % 3 (free since the barrier of slab s - 1)
        auto g0_fill = [&](auto m) __attribute__((always_inline)) {
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, w128_c<1>{}, cP{}, cW{});
            else p_piece(w128_c<(S6 + 2) % 3>{}, w128_c<m - 8>{}, k2, live2);
        };
        if (S6 == 0 && s == 0) { W128_FOR(16, m, { mma_first(w128_c<0>{}, m); g0_fill(m); }); }
        else { W128_FOR(16, m, { mma(w128_c<0>{}, m); g0_fill(m); }); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, w128_c<2>{}, cP{}, cW{});
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_FOR(16, m, {
            mma(w128_c<0>{}, m);
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, w128_c<3>{}, cP{}, cW{});
        });
        // slab s + 1 has landed (pixels: issued in G0 of slab s - 1; weights: in G3 of slab s - 1), every wave is done reading slab s;
        // the eight youngest pieces (pixel slab s + 2, issued in this slab's G0) may stay in flight
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // G3: buffer 1; reads of k-step 0 of slab s + 1; weight slab s + 2 into the stage this slab read
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, w128_c<0>{}, w128_c<nPST>{}, w128_c<nWOFF>{});
            else w_piece(w128_c<WST>{}, w128_c<m - 8>{}, k2, live2);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // first operands of the tile just set up: pixel slabs 0, 1 and weight slabs 0, 1 (pixel slab 2 follows in G0 of slab 0)
    auto fill = [&]() __attribute__((always_inline)) {
        W128_FOR(P_INSTR, i, { p_piece(w128_c<0>{}, i, 0, true); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<0>{}, i, 0, true); });
        W128_FOR(P_INSTR, i, { p_piece(w128_c<1>{}, i, 128, nslab > 1); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<1>{}, i, 128, nslab > 1); });
    };

    // ---- static tile walk: XCD x owns a contiguous chunk of the tile list, its workgroups take every (grid / 8)-th tile of it
    const int nwg = a.tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0), base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    int li = wslot;
    if (li < cnt) { setup(base + li); fill(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int stores_in_flight = 0;
    while (li < cnt) {
        // the tile's first operands (32 pieces, issued before the previous tile's result stores) have landed
        // (issued before the previous tile's 32 output + 32 statistics stores, which may stay in flight: vmcnt is 6 bits)
        if (stores_in_flight) { if (with_stats) asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
        asm volatile("s_barrier" ::: "memory");
        W128_FOR(8, q, { frag_read(w128_c<0>{}, q, w128_c<0>{}, w128_c<0>{}, w128_c<0>{}); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int s = 0; s < nslab; s += 6) {
            slab(w128_c<0>{}, s);
            if (s + 1 < nslab) slab(w128_c<1>{}, s + 1);
            if (s + 2 < nslab) slab(w128_c<2>{}, s + 2);
            if (s + 3 < nslab) slab(w128_c<3>{}, s + 3);
            if (s + 4 < nslab) slab(w128_c<4>{}, s + 4);
            if (s + 5 < nslab) slab(w128_c<5>{}, s + 5);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("s_barrier" ::: "memory");        // every wave is done with the operand stages

        int ov_t[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) ov_t[i] = ovoff[i];
        const int n0_t = n0, trow_t = trow;
        li += per;
        if (li < cnt) { setup(base + li); fill(); }

        if (G128_ABL & 8) { _Pragma("unroll") for (int t = 0; t < 16; ++t) asm volatile("" : "+a"(acc[t])); stores_in_flight = 0; continue; }
        // ---- epilogue: lane (p31, hi) holds, of tile (i, j), register e = 4 q + g <-> channel j*32 + 8 q + 4 hi + g of pixel i*32 + p31
        W128_FOR(NT, jc, {
            constexpr int j = decltype(jc)::value;
            float s1[4][4], s2[4][4];                     // [q][g] sums over the wave's four pixel blocks (values as stored)
            f32x4_t bq[4];                                // bias of the lane's channels j*32 + 8 q + 4 hi + g (layers with a bias: no BatchNorm behind them)
            if (with_bias) {
                const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, Cout_s * 4, 0x00020000);
                _Pragma("unroll") for (int q = 0; q < 4; ++q)
                    bq[q] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsBias, (n0_t + wn * 128 + j * 32 + 8 * q + 4 * hi) * 4, 0, 0));
            }
            _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int g = 0; g < 4; ++g) { s1[q][g] = 0.f; s2[q][g] = 0.f; }
            W128_FOR(MT, ic, {
                constexpr int i = decltype(ic)::value;
                asm volatile("" : "+a"(acc[i * 4 + j]));   // the tile stays in its AGPRs up to here (conv_lstm_w128.h)
                f32x16_t tv = acc[i * 4 + j];
                if (with_bias) { _Pragma("unroll") for (int e = 0; e < 16; ++e) tv[e] += bq[e >> 2][e & 3]; }
                if (with_relu) { _Pragma("unroll") for (int e = 0; e < 16; ++e) tv[e] = __builtin_fmaxf(tv[e], 0.f); }
                unsigned pk[4][2];
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {
                    pk[q][0] = pack_bf16x2(tv[q * 4 + 0], tv[q * 4 + 1]);
                    pk[q][1] = pack_bf16x2(tv[q * 4 + 2], tv[q * 4 + 3]);
                    if (with_stats) {
                        const float v0 = __uint_as_float(pk[q][0] << 16), v1 = __uint_as_float(pk[q][0] & 0xffff0000u);
                        const float v2 = __uint_as_float(pk[q][1] << 16), v3 = __uint_as_float(pk[q][1] & 0xffff0000u);
                        s1[q][0] += v0; s1[q][1] += v1; s1[q][2] += v2; s1[q][3] += v3;
                        s2[q][0] = __builtin_fmaf(v0, v0, s2[q][0]); s2[q][1] = __builtin_fmaf(v1, v1, s2[q][1]);
                        s2[q][2] = __builtin_fmaf(v2, v2, s2[q][2]); s2[q][3] = __builtin_fmaf(v3, v3, s2[q][3]);
                    }
                }
                // channel quadruples (q, q + 1) of the lane pair (l, l ^ 32) -> eight consecutive channels per lane: 16-byte stores
                // (one asm block: only the first swap can follow the VALU write of its operands closely enough to need wait states)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7"
                             : "+v"(pk[0][0]), "+v"(pk[0][1]), "+v"(pk[1][0]), "+v"(pk[1][1]), "+v"(pk[2][0]), "+v"(pk[2][1]), "+v"(pk[3][0]), "+v"(pk[3][1]));
                // lane < 32: (pk[qq][0..1], pk[qq+1][0..1]) = its own channels 8 qq .. + 3 and the partner's 8 qq + 4 .. + 7; lane >= 32: quadruple qq + 1
                if (G128_ABL & 16) {     // timing only: the same bytes as fully coalesced 1 KB pieces (wrong addresses)
                    const int cb = ((((trow_t >> 1) * tiles_n + (n0_t >> 8)) * 4 + wave) << 15) + lane * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[0][0], pk[0][1], pk[1][0], pk[1][1]}, rsO, cb + (i * 8 + j * 2) * 1024, 0, G128_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[2][0], pk[2][1], pk[3][0], pk[3][1]}, rsO, cb + (i * 8 + j * 2 + 1) * 1024, 0, G128_AUX);
                } else
                if (G128_ABL & 1) asm volatile("" :: "v"(pk[0][0]), "v"(pk[0][1]), "v"(pk[1][0]), "v"(pk[1][1]), "v"(pk[2][0]), "v"(pk[2][1]), "v"(pk[3][0]), "v"(pk[3][1]));
                else if (nt_out) {       // 4 x-expanding layers: the result (4 x the input) streams past the L2s (R6-8: -6 ... -10 %)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[0][0], pk[0][1], pk[1][0], pk[1][1]}, rsO, ov_t[i] + (j * 32) * 2, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[2][0], pk[2][1], pk[3][0], pk[3][1]}, rsO, ov_t[i] + (j * 32 + 16) * 2, 0, 2);
                } else {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[0][0], pk[0][1], pk[1][0], pk[1][1]}, rsO, ov_t[i] + (j * 32) * 2, 0, G128_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[2][0], pk[2][1], pk[3][0], pk[3][1]}, rsO, ov_t[i] + (j * 32 + 16) * 2, 0, G128_AUX);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (with_stats) {
                // 32-lane sums (lanes of one hi) on the DPP network: row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15; lane
                // 31 / 63 end up with the totals of the channels of hi = 0 / 1
                // (four independent chains per asm block: a register's next DPP read is three instructions behind its write, which covers
                //  the two wait states a VALU write -> DPP read needs without s_nops)
                if (!(G128_ABL & 4)) _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int g = 0; g < 4; g += 2) {
                    asm volatile("s_nop 1\n\t"
                                 G128_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                                 G128_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                                 G128_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                                 G128_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                                 G128_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                                 : "+v"(s1[q][g]), "+v"(s1[q][g + 1]), "+v"(s2[q][g]), "+v"(s2[q][g + 1]));
                }
                // lanes 31 and 63 write their 16 channels x {sum, sumsq} of gate block j: four 16-byte stores each per plane
                const int chan = n0_t + wn * 128 + j * 32 + 4 * hi;
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {
                    const int off0 = ((trow_t * 2 + 0) * Cout_s + chan + 8 * q) * 4, off1 = ((trow_t * 2 + 1) * Cout_s + chan + 8 * q) * 4;     // lanes other than 31 / 63: dropped
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(s1[q][0]), __float_as_uint(s1[q][1]), __float_as_uint(s1[q][2]), __float_as_uint(s1[q][3])},
                                                           rsS, (lane & 31) == 31 ? off0 : (int)oob, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(s2[q][0]), __float_as_uint(s2[q][1]), __float_as_uint(s2[q][2]), __float_as_uint(s2[q][3])},
                                                           rsS, (lane & 31) == 31 ? off1 : (int)oob, 0, 0);
                }
            }
        });
        stores_in_flight = 1;
    }
#undef G128_RD_P
#undef G128_RD_P0
#undef G128_RD_W
}
