// 3 x 3 / stride-1 / "same" convolutions (any dilation) in front of a BatchNorm on the round-6 main loop: the K loop of
// conv_lstm_w128.h (row-halo reuse of the pixel operand, one wave per SIMD on a 128 x 128 accumulator block, every instruction placed
// by hand; the tap rows move by the dilation) with the epilogue of conv_w128_gemm.h (raw bf16 result as 16-byte pieces + per-128-row
// BatchNorm statistics) and its static tile walk.  Included by conv_fwd.hip after those two files.  models/_resnet.py:96-114 (conv2 of
// the frozen teacher's dilated bottlenecks, 256 and 512 channels).
// Takes: R = S = 3, stride 1, pad = dil, Cin % 64 == 0, Cout % 256 == 0, no bias / activation / residual, bf16 output, a 256-pixel
// tile's halo within 320 rows and within the map's height.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_w128_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    constexpr int NWAVES = 4, MT = 4, NT = 4;
    constexpr int H_INSTR = W128_HROWS / 8 / NWAVES, B_INSTR = 8;
    constexpr int HALO_BYTES = W128_HALO_BYTES, WST = W128_WST_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int p31 = lane & 31, hi = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const unsigned oob = 0x80000000u;

    int Cin_s = a.Cin, H_s = a.H, W_s = a.W, M_s = a.M, ips_s = (int)a.in_pix_stride, ops_s = (int)a.out_pix_stride, Cout_s = a.Cout, tiles_n = a.tiles_n, dil_s = a.dil;
    asm volatile("" : "+s"(Cin_s), "+s"(H_s), "+s"(W_s), "+s"(M_s), "+s"(ips_s), "+s"(ops_s), "+s"(Cout_s), "+s"(tiles_n), "+s"(dil_s));
    const int nch = Cin_s >> 6, NJ = 3 * nch;
    const long long in_bytes = (((long long)M_s - 1) * ips_s + Cin_s) * 2;
    const long long out_bytes = (((long long)M_s - 1) * ops_s + Cout_s) * 2;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)out_bytes, 0x00020000);
    const int stat_rows = (M_s + 127) >> 7;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.stats, 0, a.stats ? stat_rows * 2 * Cout_s * 4 : 0, 0x00020000);
    const bool with_stats = a.stats != nullptr;
    const unsigned mg_w = a.mg_w, mg_wd = a.mg_wd;

    uint32_t wa[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * 128 + j * 32 + p31;
        const uint32_t sw = (uint32_t)((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wa[j][ks] = lds0 + (uint32_t)(2 * HALO_BYTES + r * 128) + ((((uint32_t)(ks * 2) + (uint32_t)hi) ^ sw) << 4);
            asm volatile("" : "+v"(wa[j][ks]));
        }
    }

    f32x16_t acc[16];
    bf16x8_t fp[2][MT], fw[2][NT];
    int hy[H_INSTR], hoff[H_INSTR], boff[B_INSTR];
    uint32_t pa[MT][3];
    int ovoff[MT];
    int n0 = 0, trow = 0;

    auto setup = [&](int bid) __attribute__((always_inline)) {
        int lane_o = lane, wave_o = wave;
        asm volatile("" : "+v"(lane_o), "+s"(wave_o));    // per-tile values are recomputed, not hoisted and spilled (conv_lstm_w128.h)
        const int lrow = lane_o >> 3, slot = lane_o & 7, p31 = lane_o & 31, hi = lane_o >> 5, wave = wave_o, wm = wave_o >> 1, wn = wave_o & 1;
        const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
        const int m0 = tile_m * 256;
        n0 = tile_n * 256; trow = tile_m * 2 + wm;
        const int W = W_s, dil = dil_s, wd = W + dil;
        const int hw = H_s * W;
        const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
        const int oy0 = (int)__umulhi((unsigned)rem0, mg_w), ox0 = rem0 - oy0 * W;
        const int L0 = (W - ox0 < 256) ? W - ox0 : 256;
        // halo rows = [dil lead pixels][segment 0][dil gap][segment 1][dil gap] ... [last segment + dil trail pixels]
#pragma unroll
        for (int i = 0; i < H_INSTR; ++i) {
            const int h = (wave * H_INSTR + i) * 8 + lrow;
            const int hp = h - dil;
            const int h2 = hp - (L0 + dil);
            const int q = (int)__umulhi((unsigned)(h2 < 0 ? 0 : h2), mg_wd), r = h2 - q * wd;
            const bool first = h2 < 0;
            const int m_seg = first ? m0 : m0 + L0 + q * W, px = first ? ox0 + hp : r, drow = first ? 0 : q + 1;
            const bool valid = m_seg < M_s && (m_seg == m0 || m_seg - m0 < 256) && (unsigned)px < (unsigned)W;
            int oy = oy0 + drow;
            const int grow = b0 * H_s + oy;
            if (oy >= H_s) oy -= H_s;
            hy[i] = valid ? oy : -0x4000;
            hoff[i] = valid ? ((grow * W + px) * ips_s * 2) + (slot ^ ((h >> 1) & 7)) * 16 : 0;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int r = (wave * B_INSTR + i) * 8 + lrow;
            boff[i] = ((n0 + r) * a.Kpad + (slot ^ ((r >> 1) & 7)) * 8) * 2;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = wm * 128 + i * 32 + p31;
            const int t = r - L0, q = (int)__umulhi((unsigned)(t < 0 ? 0 : t), mg_w), rr = t - q * W;
            const int hr = t < 0 ? r : L0 + dil + q * wd + rr;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int h = hr + dx * dil;
                pa[i][dx] = lds0 + (uint32_t)h * 128 + ((((uint32_t)hi) ^ (uint32_t)((h >> 1) & 7)) << 4);
            }
            const int m = m0 + r;
            ovoff[i] = m < M_s ? (m * ops_s + n0 + wn * 128 + 8 * hi) * 2 : (int)oob;
        }
    };

    // ---- LDS-DMA pieces (the VALU of a halo piece is volatile asm too: left to hipcc it is hoisted in front of the slab's first MFMA)
    auto halo_piece = [&](auto par_c, auto i_c, int ddy, int tapoff) __attribute__((always_inline)) {
        constexpr int par = decltype(par_c)::value, i = decltype(i_c)::value;
        unsigned voff;
        asm volatile("v_add_u32 %0, %1, %2\n\tv_cmp_gt_u32 vcc, %3, %0\n\tv_add_u32 %0, %4, %5\n\tv_cndmask_b32 %0, %6, %0, vcc"
                     : "=&v"(voff) : "v"(hy[i]), "s"(ddy), "s"(H_s), "v"(hoff[i]), "s"(tapoff), "v"(oob) : "vcc");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + par * HALO_BYTES + (wave * H_INSTR + i) * 1024),
                                                 16, voff, 0, 0, 0);
    };
    auto w_piece = [&](auto st_c, auto i_c, int koff) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value, i = decltype(i_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + 2 * HALO_BYTES + st * WST + (wave * B_INSTR + i) * 1024),
                                                 16, (unsigned)boff[i], koff, 0, 0);
    };
    // byte offset along K of slab (dy, cc, dx); slabs past the end re-fetch the last one (never read)
    int koff_last = 0;
    auto slab_koff = [&](int dy, int cc, int dx) {
        const int k = ((dy * 3 + dx) * Cin_s + cc * 64) * 2;
        return k < koff_last ? k : koff_last;
    };

    // the pixel fragment address of k-step KS is built by a v_xor in front of the read (12 address registers instead of 48)
#define W128_RD_P(BUF, I, DX, KS, OFF) { uint32_t t_; asm volatile("v_xor_b32 %1, %4, %2\n\tds_read_b128 %0, %1 offset:%3" : "=v"(fp[BUF][I]), "=&v"(t_) : "v"(pa[I][DX]), "n"(OFF), "n"((KS) << 5) : "memory"); }
#define W128_RD_P0(BUF, I, DX, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fp[BUF][I]) : "v"(pa[I][DX]), "n"(OFF) : "memory")
#define W128_RD_W(BUF, J, KS, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[BUF][J]) : "v"(wa[J][KS]), "n"(OFF) : "memory")
    // read piece q (0..7) of k-step KS of the slab (halo offset HOFF, tap DX, weight stage offset WOFF) into fragment buffer BUF
    auto frag_read = [&fp, &fw, &pa, &wa](auto buf_c, auto q_c, auto dx_c, auto ks_c, auto hoff_c, auto woff_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, q = decltype(q_c)::value, DX = decltype(dx_c)::value, KS = decltype(ks_c)::value;
        constexpr int HOFF = decltype(hoff_c)::value, WOFF = decltype(woff_c)::value;
        if constexpr (q == 0) W128_RD_W(BUF, 0, KS, WOFF);
        else if constexpr (q <= 4) {
            if constexpr (KS == 0) W128_RD_P0(BUF, q - 1, DX, HOFF);
            else W128_RD_P(BUF, q - 1, DX, KS, HOFF);
        }
        else W128_RD_W(BUF, q - 4, KS, WOFF);
    };
    // MFMA m of a group on fragment buffer BUF: m = j*4 + i (weights are the A operand: a lane holds gate rows of ONE pixel)
    auto mma = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };
    auto mma_first = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {   // first k-step of a tile: C = 0
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[i * 4 + j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };

    int dy_c = 0, cc_c = 0, dy_n = 0, cc_n = 0;          // (dy, chunk) of macro steps j and j + 1
    bool first_slab = false;
    // one K-slab: macro step j of parity PAR, tap DX
    auto slab = [&](auto par_c, auto dx_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, DX = decltype(dx_c)::value;
        constexpr int HOFF = PAR * HALO_BYTES, WSTAGE = (PAR + DX) & 1, WOFF = WSTAGE * WST;
        constexpr int nPAR = (DX == 2) ? (PAR ^ 1) : PAR, nDX = (DX + 1) % 3;
        constexpr int nHOFF = nPAR * HALO_BYTES, nWOFF = (WSTAGE ^ 1) * WST;
        using cDX = w128_c<DX>; using cH = w128_c<HOFF>; using cW = w128_c<WOFF>;
        // halo of macro step j + 1 (buffer PAR ^ 1): all ten pieces in slab (j, 0), G0 / G1 / G2 = 4 / 3 / 3; past the tile's last macro
        // step every row is out of range (no traffic)
        const int ddy_n = dy_n < 3 ? (dy_n - 1) * dil_s : 0x2000;
        const int tap_n = ((dy_n - 1) * dil_s * W_s * ips_s + cc_n * 64) * 2;
        // weight slab s + 2 -> the stage this slab reads (free behind the barrier): (j, DX + 2) or (j + 1, DX - 1)
        const int koff2 = (DX == 0) ? slab_koff(dy_c, cc_c, 2) : slab_koff(dy_n, cc_n, DX - 1);
        // G0: MFMAs on buffer 0, reads of k-step 1 into buffer 1
        auto g0_fill = [&](auto m) __attribute__((always_inline)) {
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, cDX{}, w128_c<1>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0) halo_piece(w128_c<PAR ^ 1>{}, w128_c<(m - 8) / 2>{}, ddy_n, tap_n);
        };
        if (PAR == 0 && DX == 0 && first_slab) { W128_FOR(16, m, { mma_first(w128_c<0>{}, m); g0_fill(m); }); }
        else { W128_FOR(16, m, { mma(w128_c<0>{}, m); g0_fill(m); }); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // G1: buffer 1, reads of k-step 2 into buffer 0
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, cDX{}, w128_c<2>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0 && m < 14) halo_piece(w128_c<PAR ^ 1>{}, w128_c<4 + (m - 8) / 2>{}, ddy_n, tap_n);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // G2: buffer 0, reads of k-step 3 into buffer 1
        W128_FOR(16, m, {
            mma(w128_c<0>{}, m);
            if constexpr (m < 8) frag_read(w128_c<1>{}, m, cDX{}, w128_c<3>{}, cH{}, cW{});
            else if constexpr (DX == 0 && (m & 1) == 0 && m < 14) halo_piece(w128_c<PAR ^ 1>{}, w128_c<7 + (m - 8) / 2>{}, ddy_n, tap_n);
        });
        // slab s + 1 landed (this wave's pieces), every wave is done reading slab s
        if constexpr (DX == 0) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // G3: buffer 1, reads of k-step 0 of slab s + 1 into buffer 0, weight slab s + 2
        W128_FOR(16, m, {
            mma(w128_c<1>{}, m);
            if constexpr (m < 8) frag_read(w128_c<0>{}, m, w128_c<nDX>{}, w128_c<0>{}, w128_c<nHOFF>{}, w128_c<nWOFF>{});
            else w_piece(w128_c<WSTAGE>{}, w128_c<m - 8>{}, koff2);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PAR == 0 && DX == 0) first_slab = false;
    };
    auto macro_step = [&](auto par_c) __attribute__((always_inline)) {
        dy_n = dy_c; cc_n = cc_c + 1;
        if (cc_n == nch) { cc_n = 0; ++dy_n; }
        slab(par_c, w128_c<0>{});
        slab(par_c, w128_c<1>{});
        slab(par_c, w128_c<2>{});
        dy_c = dy_n; cc_c = cc_n;
    };
    // first halo + weight slabs 0, 1 of the tile just set up (all operand buffers are free: behind a barrier every wave has passed)
    auto fill = [&]() __attribute__((always_inline)) {
        koff_last = ((8 * Cin_s) + (nch - 1) * 64) * 2;
        W128_FOR(H_INSTR, i, { halo_piece(w128_c<0>{}, i, -dil_s, (-dil_s * W_s * ips_s) * 2); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<0>{}, i, slab_koff(0, 0, 0)); });
        W128_FOR(B_INSTR, i, { w_piece(w128_c<1>{}, i, slab_koff(0, 0, 1)); });
    };

    // ---- static tile walk (as conv1x1_w128_kernel)
    const int nwg = a.tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0), base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    int li = wslot;
    if (li < cnt) { setup(base + li); fill(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int stores_in_flight = 0;
    while (li < cnt) {
        // the tile's first operands (26 pieces, issued before the previous tile's 32 output + 32 statistics stores) have landed
        if (stores_in_flight) { if (with_stats) asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
        asm volatile("s_barrier" ::: "memory");
        W128_FOR(8, q, { frag_read(w128_c<0>{}, q, w128_c<0>{}, w128_c<0>{}, w128_c<0>{}, w128_c<0>{}); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dy_c = 0; cc_c = 0; first_slab = true;
        for (int j = 0; j < NJ; j += 2) {
            macro_step(w128_c<0>{});
            if (j + 1 < NJ) macro_step(w128_c<1>{});
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("s_barrier" ::: "memory");

        int ov_t[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) ov_t[i] = ovoff[i];
        const int n0_t = n0, trow_t = trow;
        li += per;
        if (li < cnt) { setup(base + li); fill(); }

        // ---- epilogue: lane (p31, hi) holds, of tile (i, j), register e = 4 q + g <-> channel j*32 + 8 q + 4 hi + g of pixel i*32 + p31
        W128_FOR(NT, jc, {
            constexpr int j = decltype(jc)::value;
            float s1[4][4], s2[4][4];                     // [q][g] sums over the wave's four pixel blocks (values as stored)
            _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int g = 0; g < 4; ++g) { s1[q][g] = 0.f; s2[q][g] = 0.f; }
            W128_FOR(MT, ic, {
                constexpr int i = decltype(ic)::value;
                asm volatile("" : "+a"(acc[i * 4 + j]));   // the tile stays in its AGPRs up to here (conv_lstm_w128.h)
                const f32x16_t tv = acc[i * 4 + j];
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
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[0][0], pk[0][1], pk[1][0], pk[1][1]}, rsO, ov_t[i] + (j * 32) * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{pk[2][0], pk[2][1], pk[3][0], pk[3][1]}, rsO, ov_t[i] + (j * 32 + 16) * 2, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (with_stats) {
                // 32-lane sums (lanes of one hi) on the DPP network: row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15; lane
                // 31 / 63 end up with the totals of the channels of hi = 0 / 1
                // (four independent chains per asm block: a register's next DPP read is three instructions behind its write, which covers
                //  the two wait states a VALU write -> DPP read needs without s_nops)
                _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int g = 0; g < 4; g += 2) {
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
#undef W128_RD_P
#undef W128_RD_P0
#undef W128_RD_W
}
