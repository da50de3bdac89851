// Fused ConvLSTM gate convolution on 128 x 128 WAVE tiles (round 6).  Included by conv_fwd.hip inside its anonymous namespace
// (uses ConvArgs / ConvGroup / lstm_bias_init / lstm_epilogue of that file).  e2vid/model/submodules.py:175-214.
//
// Why: the 64 x 64 wave tile of conv3x3_halo_tile issues one ds_read_b128 per MFMA (round-5 PMC: 1.1 LDS instructions per MFMA,
// matrix pipe 46 % busy, waves 40 % in issue stalls).  Here ONE wave per SIMD owns a 128-pixel x 128-gate-column accumulator
// block (16 x 32x32 tiles = 256 AGPRs of the unified 512-entry file): 8 fragment reads per 16 MFMAs = 0.5 per MFMA, and the
// workgroup tile is 256 pixels x 256 gate columns (64 hidden channels), so a K-slab moves 32 KB of weights + a third of a 40 KB
// halo for 256 MFMAs (0.71 KB of L2 -> LDS traffic per MFMA; the 256 x 128 tile moves 0.92).
//
// Nothing hides behind a second wave on the SIMD, so the instruction stream is laid out by hand: every instruction of the K loop
// is a volatile asm statement (MFMAs with "+a" accumulators, ds_read_b128, s_waitcnt, s_barrier) or an LDS-DMA builtin -- hipcc
// keeps their relative order and only allocates registers.  A K-slab (64 k) is four groups G0..G3 of 16 MFMAs; gap m of a group
// (the issue slots behind MFMA m) carries
//     m = 0..7   one fragment read of the NEXT k-step into the other fragment buffer (W0 P0 P1 P2 P3 W1 W2 W3),
//     m = 8..15  at most one LDS-DMA piece (1 KB),
// and the group starts with lgkmcnt(0), which the last read precedes by eight MFMAs (256 cycles).  The barrier that opens slab
// s + 1 sits between G2 and G3 of slab s: behind it G3's gaps read k-step 0 of slab s + 1 (across the slab boundary) and issue
// the weight slab s + 2 into the stage slab s has just finished reading; the halo of macro step j + 1 is issued in G0..G2 of
// slab (j, dx = 0).  vmcnt is counted: 10 halo pieces stay in flight across the barrier of (j, 0).
// LDS: [halo 0][halo 1] 2 x 40 KB, [weights 0][weights 1] 2 x 32 KB = 144 KB, one workgroup per CU.
// Requires: 3 x 3, dil 1, stride 1, Cin % 128 == 0 (an even number of macro steps), Cout % 256 == 0, lds base 128-byte aligned.
#ifndef W128_ABL
#define W128_ABL 0     // debug ablations (tools/bench_lstm_group.py): 1 no epilogue, 2 no LDS-DMA in the K loop, 4 no barrier, 8 no fragment reads,
                       // 16 no vmcnt waits, 32 no halo DMA, 64 no weight DMA, 128 no previous-cell loads, 256 no gate math, 512 no stores
#endif
constexpr int W128_HROWS = 320;
constexpr int W128_HALO_BYTES = W128_HROWS * 128;         // 40 960
constexpr int W128_WST_BYTES = 256 * 128;                 // 32 768
constexpr int W128_LDS = 2 * W128_HALO_BYTES + 2 * W128_WST_BYTES;   // 147 456

// halo piece issued in gap m of group G (0..2) of slab dx, or -1.  Default: all ten pieces in slab dx = 0 (4 / 3 / 3 over G0..G2, every
// other gap from m = 8).  W128_ABL & 4096: spread over the three slabs (4 / 3 / 3 pieces; G0 gaps 8, 12 and G1 gap 8 (+ gap 12 on dx = 0)).
constexpr bool W128_SPREAD = (W128_ABL & 4096) != 0;
constexpr int w128_halo_piece_at(int dx, int G, int m) {
    if (W128_SPREAD) {
        const int base = dx == 0 ? 0 : (dx == 1 ? 4 : 7);
        if (G == 0 && m == 8) return base;
        if (G == 0 && m == 12) return base + 1;
        if (G == 1 && m == 8) return base + 2;
        if (G == 1 && m == 12 && dx == 0) return 3;
        return -1;
    }
    if (dx != 0 || m < 8 || (m & 1)) return -1;
    const int k = (m - 8) / 2;
    if (G == 0) return k;
    if (k >= 3) return -1;
    return G == 1 ? 4 + k : 7 + k;
}
constexpr int w128_halo_vmcnt(int dx) { return W128_SPREAD ? (dx == 0 ? 4 : (dx == 1 ? 3 : 0)) : (dx == 0 ? 10 : 0); }

template <typename F, int... Is>
__device__ __forceinline__ void w128_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
#define W128_FOR(N, VAR, ...) w128_for([&](auto VAR) __attribute__((always_inline)) __VA_ARGS__, std::make_integer_sequence<int, N>{})

// Cell update of a 256-pixel x 64-hidden-channel tile straight from the accumulators (layout: lstm_epilogue of conv_fwd.hip, four
// waves as 2 x 2, 64 cells per lane).  smem: fp32 cell image [256][65] + bf16 hidden image [256][66].
__device__ __forceinline__ void w128_lstm_epilogue(const ConvArgs& a, f32x16_t (&acc)[4][4], unsigned char* smem, int m0, int n0,
                                                   int wm, int wn, int lane, int tid) {
    constexpr int MT = 4, NT = 4, ROWS = 256, HC = 64, CP = HC + 1, HP = HC + 2;
    float* lc = reinterpret_cast<float*>(smem);
    uint16_t* lh = reinterpret_cast<uint16_t*>(smem + ROWS * CP * 4);
    const int C = a.lstm_C;
    const int hc0 = n0 >> 2;
    const int p = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = wm * 128 + i * 32 + p;
        const int m = m0 + ml;
        const bool valid = m < a.M;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hcl = wn * 32 + j * 8 + 2 * q + hi;
                const int hc = hc0 + hcl;
                const float gi = acc[i][j][q * 4 + 0], gr = acc[i][j][q * 4 + 1], go = acc[i][j][q * 4 + 2], gc = acc[i][j][q * 4 + 3];
                float pc = 0.0f;
                if constexpr ((W128_ABL & 128) == 0) pc = (a.lstm_prev && valid) ? a.lstm_prev[(long long)m * C + hc] : 0.0f;
                float nc, hv;
                if constexpr ((W128_ABL & 256) != 0) { nc = gr * pc + gi * gc; hv = go * nc; }
                else {
                    nc = fast_sigmoid(gr) * pc + fast_sigmoid(gi) * fast_tanh(gc);     // submodules.py:211
                    hv = fast_sigmoid(go) * fast_tanh(nc);                              // submodules.py:212
                }
                lc[ml * CP + hcl] = nc;
                lh[ml * HP + hcl] = (uint16_t)pack_bf16x2(hv, 0.0f);
            }
    }
    __syncthreads();
    if constexpr ((W128_ABL & 512) != 0) { if (lc[tid] == 12345.678f) a.lstm_cell[tid] = 1.f; return; }
    const bool vec_ok = (C & 3) == 0 && (a.lstm_h_stride & 7) == 0 && (((uintptr_t)a.lstm_h | (uintptr_t)a.lstm_cell) & 15) == 0;
    if (vec_ok) {
#pragma unroll
        for (int idx = tid; idx < ROWS * (HC / 4); idx += 256) {
            const int row = idx / (HC / 4), c4 = idx - row * (HC / 4);
            const int m = m0 + row;
            const float* sp = lc + row * CP + c4 * 4;
            if (m < a.M) out_store16(a.lstm_cell + (long long)m * C + hc0 + c4 * 4, make_uint4(__float_as_uint(sp[0]), __float_as_uint(sp[1]), __float_as_uint(sp[2]), __float_as_uint(sp[3])));
        }
        const uint32_t* lhv = reinterpret_cast<const uint32_t*>(lh);
#pragma unroll
        for (int idx = tid; idx < ROWS * (HC / 8); idx += 256) {
            const int row = idx / (HC / 8), c8 = idx - row * (HC / 8);
            const int m = m0 + row;
            const uint32_t* sp = lhv + row * (HP / 2) + c8 * 4;
            if (m < a.M) out_store16(a.lstm_h + (long long)m * a.lstm_h_stride + hc0 + c8 * 8, make_uint4(sp[0], sp[1], sp[2], sp[3]));
        }
        return;
    }
    for (int idx = tid; idx < ROWS * HC; idx += 256) {
        const int row = idx / HC, col = idx - row * HC;
        const int m = m0 + row;
        if (m < a.M) a.lstm_cell[(long long)m * C + hc0 + col] = lc[row * CP + col];
    }
    const uint32_t* lh32 = reinterpret_cast<const uint32_t*>(lh);
    for (int idx = tid; idx < ROWS * (HC / 2); idx += 256) {
        const int row = idx / (HC / 2), col = idx - row * (HC / 2);
        const int m = m0 + row;
        if (m < a.M) *reinterpret_cast<uint32_t*>(a.lstm_h + (long long)m * a.lstm_h_stride + hc0 + col * 2) = lh32[row * (HP / 2) + col];
    }
}

// W128_ABL & 8192: s_memtime stamps (debug).  Each wave sums the cycles of group G of slab dx into tacc[dx * 4 + G] (a stamp is taken
// just before the s_waitcnt that closes a group, so a group's figure = the wait in front of it + its 16 MFMAs), plus the phases
// prologue / fill / K loop / epilogue in tacc[12..15]; wave w of every problem's first tile overwrites lstm_cell[w * 16 + k].
constexpr bool W128_STAMP = (W128_ABL & 8192) != 0;
#define W128_STAMP_TAKE() do { if constexpr (W128_STAMP) asm volatile("s_memtime %0" : "=s"(tnow)); } while (0)
#define W128_STAMP_ADD(K) do { if constexpr (W128_STAMP) { tacc[K] += (unsigned)tnow - tlast; tlast = (unsigned)tnow; } } while (0)

__device__ __forceinline__ void conv3x3_lstm_w128_tile(const ConvArgs& a, const int bid, unsigned char* smem) {
    unsigned long long tnow = 0; unsigned tlast = 0; unsigned tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    W128_STAMP_TAKE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tlast = (unsigned)tnow;
    constexpr int BMX = 256, BN = 256, NWAVES = 4, MT = 4, NT = 4;
    constexpr int H_INSTR = W128_HROWS / 8 / NWAVES;     // 10 halo pieces per wave and macro step
    constexpr int B_INSTR = BN * 8 / 64 / NWAVES;        // 8 weight pieces per wave and slab
    constexpr int HALO_BYTES = W128_HALO_BYTES, WST = W128_WST_BYTES;

    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BMX, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int Cin_s = a.Cin, H_s = a.H, ips_s = (int)a.in_pix_stride;
    asm volatile("" : "+s"(Cin_s), "+s"(H_s), "+s"(ips_s));
    const int nch = Cin_s >> 6;
    const int NJ = 3 * nch;                              // macro steps (dy, chunk); even by the dispatch rule
    const int W = a.W, dil = a.dil, wd = W + dil;

    const long long in_bytes = (((long long)a.B * a.H * a.W - 1) * a.in_pix_stride + a.Cin) * 2;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7ffffff0, 0x00020000);

    const int hw = a.H * W;
    const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
    const int oy0 = rem0 / W, ox0 = rem0 - oy0 * W;
    const int L0 = (W - ox0 < BMX) ? W - ox0 : BMX;

    // ---- halo DMA geometry (as conv3x3_halo_tile): lane (lrow, slot) of piece q writes halo row q*8 + lrow, 16-byte slot `slot`
    const int lrow = lane >> 3, slot = lane & 7;
    int hy[H_INSTR], hoff[H_INSTR];
#pragma unroll
    for (int i = 0; i < H_INSTR; ++i) {
        const int h = (wave * H_INSTR + i) * 8 + lrow;
        const int hp = h - dil;
        int m_seg, px, drow;
        if (hp < L0 + dil) { m_seg = m0; px = ox0 + hp; drow = 0; }
        else {
            const int h2 = hp - (L0 + dil);
            const int q = a.mg_wd ? (int)__umulhi((unsigned)h2, a.mg_wd) : h2 / wd, r = h2 - q * wd;
            m_seg = m0 + L0 + q * W; px = r; drow = q + 1;
        }
        const bool valid = m_seg < a.M && (m_seg == m0 || m_seg - m0 < BMX) && (unsigned)px < (unsigned)W;
        int oy = oy0 + drow;
        const long long grow = (long long)b0 * a.H + oy;
        while (oy >= a.H) oy -= a.H;
        hy[i] = valid ? oy : -0x4000;
        hoff[i] = valid ? (int)((grow * W + px) * a.in_pix_stride * 2) + (slot ^ ((h >> 1) & 7)) * 16 : 0;
    }
    int boff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        boff[i] = ((n0 + r) * a.Kpad + (slot ^ ((r >> 1) & 7)) * 8) * 2;
    }

    // ---- accumulators (AGPRs): acc[i][j] = pixels i*32.. x gate rows j*32.., start value = gate bias
    f32x16_t acc[MT][NT];
    lstm_bias_init<MT, NT>(a, acc, n0, wn, lane);

    // ---- fragment addresses, complete: one VGPR per (pixel block, dx, k-step) and per (gate block, k-step); the halo buffer and
    // the weight stage enter as the ds_read's immediate offset.  k-step ks of a row reads 16-byte chunk (2 ks + half) ^ sw.
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t half = (uint32_t)(lane >> 5);
    uint32_t pa[MT][3][4], wa[NT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 128 + i * 32 + (lane & 31);
        int hr;
        if (r < L0) hr = r;
        else {
            const int t = r - L0, q = a.mg_w ? (int)__umulhi((unsigned)t, a.mg_w) : t / W, rr = t - q * W;
            hr = L0 + dil + q * wd + rr;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int h = hr + dx * dil;
            const uint32_t sw = (uint32_t)((h >> 1) & 7);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                pa[i][dx][ks] = lds0 + (uint32_t)h * 128 + ((((uint32_t)(ks * 2) + half) ^ sw) << 4);
                asm volatile("" : "+v"(pa[i][dx][ks]));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * 128 + j * 32 + (lane & 31);
        const uint32_t sw = (uint32_t)((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wa[j][ks] = lds0 + (uint32_t)(2 * HALO_BYTES + r * 128) + ((((uint32_t)(ks * 2) + half) ^ sw) << 4);
            asm volatile("" : "+v"(wa[j][ks]));
        }
    }

    // ---- LDS-DMA pieces
    // halo piece i of the macro step (dy, cc) into halo buffer `par`
    // (the four VALU instructions of a piece are volatile asm as well: left to hipcc they are hoisted in front of the slab's first
    //  MFMA, ~40 instructions during which the matrix pipe idles)
    const unsigned oob = 0x80000000u;
    auto halo_piece0 = [&](auto par_c, auto i_c, int ddy, int tapoff) __attribute__((always_inline)) {
        constexpr int par = decltype(par_c)::value, i = decltype(i_c)::value;
        if constexpr ((W128_ABL & 2048) != 0) {                     // no in-loop VALU: unchecked rows (timing only)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + par * HALO_BYTES + (wave * H_INSTR + i) * 1024),
                                                     16, (unsigned)hoff[i], (W128_ABL & 1024) ? 0 : tapoff + 2 * dil * W * ips_s, 0, 0);
            return;
        }
        if constexpr ((W128_ABL & 1024) != 0) { ddy = 0; tapoff = 0; }     // always the tile's own rows (cache-hot; timing only)
        unsigned voff;
        asm volatile("v_add_u32 %0, %1, %2\n\tv_cmp_gt_u32 vcc, %3, %0\n\tv_add_u32 %0, %4, %5\n\tv_cndmask_b32 %0, %6, %0, vcc"
                     : "=&v"(voff) : "v"(hy[i]), "s"(ddy), "s"(H_s), "v"(hoff[i]), "s"(tapoff), "v"(oob) : "vcc");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + par * HALO_BYTES + (wave * H_INSTR + i) * 1024),
                                                 16, voff, 0, 0, 0);
    };
    // weight piece i of the slab at byte offset koff along K (scalar offset of the instruction) into weight stage `st`
    auto w_piece0 = [&](auto st_c, auto i_c, int koff) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value, i = decltype(i_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + 2 * HALO_BYTES + st * WST + (wave * B_INSTR + i) * 1024),
                                                 16, (unsigned)boff[i], koff, 0, 0);
    };
    auto halo_piece = [&](auto par_c, auto i_c, int ddy, int tapoff) __attribute__((always_inline)) {
        if constexpr ((W128_ABL & (2 | 32)) == 0) halo_piece0(par_c, i_c, ddy, tapoff);
    };
    auto w_piece = [&](auto st_c, auto i_c, int koff) __attribute__((always_inline)) {
        if constexpr ((W128_ABL & (2 | 64)) == 0) w_piece0(st_c, i_c, koff);
    };
    // byte offset along K of slab (dy, cc, dx); slabs past the end re-fetch the last one (never read)
    const int koff_last = ((8 * Cin_s) + (nch - 1) * 64) * 2;
    auto slab_koff = [&](int dy, int cc, int dx) {
        const int k = ((dy * 3 + dx) * Cin_s + cc * 64) * 2;
        return k < koff_last ? k : koff_last;
    };

    bf16x8_t fp[2][MT], fw[2][NT];                       // fragment double buffer: pixels / weights

#define W128_RD_P(BUF, I, DX, KS, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fp[BUF][I]) : "v"(pa[I][DX][KS]), "n"(OFF) : "memory")
#define W128_RD_W(BUF, J, KS, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[BUF][J]) : "v"(wa[J][KS]), "n"(OFF) : "memory")
    // read piece q (0..7) of k-step KS of the slab (halo offset HOFF, tap DX, weight stage offset WOFF) into fragment buffer BUF
    auto frag_read = [&fp, &fw, &pa, &wa](auto buf_c, auto q_c, auto dx_c, auto ks_c, auto hoff_c, auto woff_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, q = decltype(q_c)::value, DX = decltype(dx_c)::value, KS = decltype(ks_c)::value;
        constexpr int HOFF = decltype(hoff_c)::value, WOFF = decltype(woff_c)::value;
        if constexpr ((W128_ABL & 8) != 0) return;
        if constexpr (q == 0) W128_RD_W(BUF, 0, KS, WOFF);
        else if constexpr (q <= 4) W128_RD_P(BUF, q - 1, DX, KS, HOFF);
        else W128_RD_W(BUF, q - 4, KS, WOFF);
    };
    // MFMA m of a group on fragment buffer BUF: m = j*4 + i (weights are the A operand: the result is transposed, see lstm_epilogue)
    auto mma = [&acc, &fp, &fw](auto buf_c, auto m_c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, m = decltype(m_c)::value, j = m >> 2, i = m & 3;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[BUF][j]), "v"(fp[BUF][i]));
    };

    // ---- pipeline fill: halo 0, weight slabs 0 and 1, fragments of k-step 0
    int dy_c = 0, cc_c = 0, dy_n = 0, cc_n = 0;          // (dy, chunk) of macro steps j and j + 1
    W128_STAMP_TAKE(); if constexpr (W128_STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W128_STAMP_ADD(12);
    W128_FOR(H_INSTR, i, { halo_piece0(std::integral_constant<int, 0>{}, i, -dil, (-dil * W * ips_s) * 2); });
    W128_FOR(B_INSTR, i, { w_piece0(std::integral_constant<int, 0>{}, i, slab_koff(0, 0, 0)); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    W128_FOR(8, q, { frag_read(std::integral_constant<int, 0>{}, q, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{},
                               std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); });
    W128_FOR(B_INSTR, i, { w_piece0(std::integral_constant<int, 1>{}, i, slab_koff(0, 0, 1)); });
    W128_STAMP_TAKE();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W128_STAMP_ADD(13);

    // one K-slab: macro step j of parity PAR, tap DX
    auto slab = [&](auto par_c, auto dx_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, DX = decltype(dx_c)::value;
        constexpr int HOFF = PAR * HALO_BYTES, WSTAGE = (PAR + DX) & 1, WOFF = WSTAGE * WST;
        constexpr int nPAR = (DX == 2) ? (PAR ^ 1) : PAR, nDX = (DX + 1) % 3;
        constexpr int nHOFF = nPAR * HALO_BYTES, nWOFF = (WSTAGE ^ 1) * WST;
        using cDX = std::integral_constant<int, DX>;
        using cH = std::integral_constant<int, HOFF>;
        using cW = std::integral_constant<int, WOFF>;
        // halo of macro step j + 1 (buffer PAR ^ 1): all ten pieces in slab (j, 0), G0 / G1 / G2 = 4 / 3 / 3
        const int ddy_n = (dy_n - 1) * dil;
        const int tap_n = (ddy_n * W * ips_s + cc_n * 64) * 2;
        // weight slab s + 2 -> the stage this slab reads (free behind the barrier): (j, DX + 2) or (j + 1, DX - 1)
        const int koff2 = (DX == 0) ? slab_koff(dy_c, cc_c, 2) : slab_koff(dy_n, cc_n, DX - 1);
        // G0: MFMAs on buffer 0, reads of k-step 1 into buffer 1
        W128_FOR(16, m, {
            mma(std::integral_constant<int, 0>{}, m);
            if constexpr (m < 8) frag_read(std::integral_constant<int, 1>{}, m, cDX{}, std::integral_constant<int, 1>{}, cH{}, cW{});
            else if constexpr (w128_halo_piece_at(DX, 0, m) >= 0) halo_piece(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, w128_halo_piece_at(DX, 0, m)>{}, ddy_n, tap_n);
        });
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 0);
        // G1: buffer 1, reads of k-step 2 into buffer 0
        W128_FOR(16, m, {
            mma(std::integral_constant<int, 1>{}, m);
            if constexpr (m < 8) frag_read(std::integral_constant<int, 0>{}, m, cDX{}, std::integral_constant<int, 2>{}, cH{}, cW{});
            else if constexpr (w128_halo_piece_at(DX, 1, m) >= 0) halo_piece(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, w128_halo_piece_at(DX, 1, m)>{}, ddy_n, tap_n);
        });
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 1);
        // G2: buffer 0, reads of k-step 3 into buffer 1
        W128_FOR(16, m, {
            mma(std::integral_constant<int, 0>{}, m);
            if constexpr (m < 8) frag_read(std::integral_constant<int, 1>{}, m, cDX{}, std::integral_constant<int, 3>{}, cH{}, cW{});
            else if constexpr (w128_halo_piece_at(DX, 2, m) >= 0) halo_piece(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, w128_halo_piece_at(DX, 2, m)>{}, ddy_n, tap_n);
        });
        // slab s + 1 landed (this wave's pieces), every wave is done reading slab s
        W128_STAMP_TAKE();
        if constexpr ((W128_ABL & 16) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(w128_halo_vmcnt(DX)) : "memory");
        if constexpr ((W128_ABL & 4) == 0) asm volatile("s_barrier" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 2);
        // G3: buffer 1, reads of k-step 0 of slab s + 1 into buffer 0, weight slab s + 2
        W128_FOR(16, m, {
            mma(std::integral_constant<int, 1>{}, m);
            if constexpr (m < 8) frag_read(std::integral_constant<int, 0>{}, m, std::integral_constant<int, nDX>{}, std::integral_constant<int, 0>{},
                                           std::integral_constant<int, nHOFF>{}, std::integral_constant<int, nWOFF>{});
            else w_piece(std::integral_constant<int, WSTAGE>{}, std::integral_constant<int, m - 8>{}, koff2);
        });
        W128_STAMP_TAKE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W128_STAMP_ADD(DX * 4 + 3);
    };
    auto macro_step = [&](auto par_c) __attribute__((always_inline)) {
        dy_n = dy_c; cc_n = cc_c + 1;
        if (cc_n == nch) { cc_n = 0; ++dy_n; }
        slab(par_c, std::integral_constant<int, 0>{});
        slab(par_c, std::integral_constant<int, 1>{});
        slab(par_c, std::integral_constant<int, 2>{});
        dy_c = dy_n; cc_c = cc_n;
    };
    for (int j = 0; j < NJ; j += 2) {
        macro_step(std::integral_constant<int, 0>{});
        macro_step(std::integral_constant<int, 1>{});
    }
#undef W128_RD_P
#undef W128_RD_W
    // the MFMAs are opaque to hipcc's hazard recognizer: let the last ones retire before the accumulators are read
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    if constexpr ((W128_ABL & 1) != 0) {
        float sum = 0.f;
        for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 12345.678f) a.lstm_cell[tid] = sum;
        return;
    }
    w128_lstm_epilogue(a, acc, smem, m0, n0, wm, wn, lane, tid);
    if constexpr (W128_STAMP) {
        W128_STAMP_TAKE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W128_STAMP_ADD(14);
        __syncthreads();
        if (bid == 0 && lane == 0) {
            tacc[15] = (unsigned)NJ;
#pragma unroll
            for (int k = 0; k < 16; ++k) a.lstm_cell[wave * 16 + k] = (float)tacc[k];
        }
    }
}

// up to three problems in one launch (ConvGroup as conv3x3_halo_group_kernel); tiles_m counts 256-pixel tiles, tiles_n 256-column tiles
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_lstm_w128_group_kernel(ConvGroup g) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int p = (idx >= g.start8[1] ? 1 : 0) + (idx >= g.start8[2] ? 1 : 0);
    const int li = idx - g.start8[p];
    const ConvArgs& a = g.a[p];
    const int nwg = a.tiles_m * a.tiles_n;
    const int q = nwg >> 3, r = nwg & 7;
    if (li >= q + (xcd < r ? 1 : 0)) return;
    conv3x3_lstm_w128_tile(a, (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + li, smem);
}
